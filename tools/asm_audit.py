"""Audit of hand-issued LDS loads in a gfx950 assembly listing (hipcc -S --cuda-device-only).

The exact bf16 K scan (kvc_score.hip: ld_step / wait_step) issues its MFMA operands with inline-asm `ds_read_*`
instructions whose completion the compiler does not track (cdna_hip_programming.md §5.7 item 1): the destination VGPRs
count as written at `;;#ASMEND`, the data lands later, and the asm `s_waitcnt lgkmcnt(N)` that retires them names the
same variables "+v".  That pins ORDER, not register allocation: under register pressure the compiler may copy, spill or
reuse a destination between the load and its wait, and the copy races with the LDS return — wrong values on some waves
of some launches.  This tool makes the absence of that a checked property of every build.

Per kernel, walking the instruction stream in program order:
  * a `;;#ASMSTART` block containing `ds_read*` opens pending registers: its destination VGPRs;
  * a block containing `s_waitcnt lgkmcnt(N) ; retire vA vB ...` retires the named registers and every load issued
    before them (the LDS returns a wave's reads in order).  The wait statement prints its "+v" operands in that comment:
    had the register allocator inserted a copy between load and wait, the names would not be the load's destinations —
    reported.  The count N must equal the number of asm loads issued after the youngest retired one;
  * between a load and its retirement every compiler instruction must be `v_mfma*` or scalar (`s_*`, no branch) and must
    not name a pending register; a label (basic-block boundary) while loads are pending is reported too;
  * at the end of the kernel nothing may be pending.
Returned with the findings: NumVgprs, ScratchSize, Occupancy and the line numbers of scratch accesses, so a test can
require that spills stay out of the tile loop.

    python tools/asm_audit.py file.s [kernel-name-substring ...]
"""
import re
import sys

_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_in(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_kernels(lines):
    """{symbol: (line number of the label, body lines up to .Lfunc_end, metadata comment lines after it)}."""
    out, name, start, body, meta, in_body = {}, None, 0, [], [], False
    for no, ln in enumerate(lines, 1):
        m = re.match(r"^(_Z\w+):\s*; @", ln)
        if m:
            if name is not None:
                out[name] = (start, body, meta)
            name, start, body, meta, in_body = m.group(1), no, [], [], True
            continue
        if name is None:
            continue
        if in_body:
            body.append(ln)
            if ln.startswith(".Lfunc_end"):
                in_body = False
        else:
            meta.append(ln)
    if name is not None:
        out[name] = (start, body, meta)
    return out


def asm_blocks(body):
    """Yield ('asm', index, [instructions]) for inline-asm blocks and ('ins', index, text) for compiler instructions / labels."""
    i, n = 0, len(body)
    while i < n:
        s = body[i].strip()
        if s.startswith(";;#ASMSTART"):
            j, block = i + 1, []
            while j < n and not body[j].strip().startswith(";;#ASMEND"):
                block.append(body[j].strip())
                j += 1
            yield "asm", i, block
            i = j + 1
            continue
        if s and not s.startswith(";") and not s.startswith("."):
            yield "ins", i, s
        elif re.match(r"^\.LBB\w+:", s):
            yield "ins", i, s
        i += 1


def audit_kernel(body, first_line=0, meta=()):
    problems, pending, seq = [], {}, 0       # pending: vgpr -> sequence number of the asm load writing it
    stats = {"asm_load_blocks": 0, "asm_wait_blocks": 0, "asm_loads": 0, "scratch_lines": []}
    for kind, i, item in asm_blocks(body):
        no = first_line + 1 + i
        if kind == "asm":
            for b in item:
                if b.startswith("ds_read"):
                    stats["asm_loads"] += 1
                    dst = regs_in(b.split(",", 1)[0])
                    addr = regs_in(b.split(",", 1)[1].split(";")[0])
                    if addr & set(pending):
                        problems.append(f"line {no}: asm load addresses through pending v{sorted(addr & set(pending))}")
                    if dst & addr:
                        problems.append(f"line {no}: asm load destination overlaps its address register")
                    seq += 1
                    for r in dst:
                        pending[r] = seq
                elif b.startswith("s_waitcnt") and "lgkmcnt" in b:
                    stats["asm_wait_blocks"] += 1
                    cnt = int(re.search(r"lgkmcnt\((\d+)\)", b).group(1))
                    named = regs_in(b.split(";", 1)[1]) if ";" in b else set()
                    if not named:
                        if cnt == 0:
                            pending.clear()
                        elif pending:
                            problems.append(f"line {no}: counted asm wait without a '; retire' list while loads are pending")
                        continue
                    ghost = named - set(pending)
                    if ghost:
                        problems.append(f"line {no}: wait retires v{sorted(ghost)} which no pending asm load writes "
                                        f"(a compiler copy between load and wait?)")
                    horizon = max((pending[r] for r in named if r in pending), default=0)
                    younger = seq - horizon
                    if cnt > younger:
                        problems.append(f"line {no}: lgkmcnt({cnt}) but only {younger} asm loads are younger than the "
                                        f"registers it retires: they may not have landed")
                    for r in [r for r, q in pending.items() if q <= horizon]:
                        del pending[r]
            if any(b.startswith("ds_read") for b in item):
                stats["asm_load_blocks"] += 1
            continue
        s = item
        if s.startswith("scratch_"):
            stats["scratch_lines"].append(no)
        if not pending:
            continue
        if re.match(r"^[.\w$]+:", s):
            problems.append(f"line {no}: label {s.split(':')[0]} while asm loads are pending")
            continue
        op, text = s.split()[0], s.split(";", 1)[0]
        touched = regs_in(text) & set(pending)
        if touched:
            problems.append(f"line {no}: `{text.strip()}` names pending v{sorted(touched)} before their wait")
        if not (op.startswith("v_mfma") or op.startswith("s_")) or op.startswith("s_cbranch") or op in (
                "s_branch", "s_endpgm", "s_setpc_b64", "s_barrier"):
            problems.append(f"line {no}: `{op}` between an asm load and its wait (only v_mfma* / plain s_* expected)")
    if pending:
        problems.append(f"end of kernel with pending asm loads v{sorted(pending)}")
    for ln in meta:
        m = re.match(r"^;\s*(ScratchSize|NumVgprs|NumAgprs|Occupancy):\s*(\d+)", ln)
        if m:
            stats[m.group(1)] = int(m.group(2))
    return problems, stats


def loop_spans(body, first_line=0):
    """(first, last) listing lines of every backward branch's span: the loops of the kernel."""
    labels, spans = {}, []
    for i, ln in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", ln.strip())
        if m:
            labels[m.group(1)] = i
    for i, ln in enumerate(body):
        m = re.match(r"^\s*s_c?branch\w*\s+(\.LBB\w+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            spans.append((first_line + 1 + labels[m.group(1)], first_line + 1 + i))
    return spans


def audit_file(path, wanted=()):
    """{kernel: (problems, stats)} for every kernel of the listing that issues asm LDS loads (or is named in `wanted`)."""
    lines = open(path).read().split("\n")
    res = {}
    for name, (start, body, meta) in split_kernels(lines).items():
        if wanted and not any(w in name for w in wanted):
            continue
        has_asm_load = any(k == "asm" and any(b.startswith("ds_read") for b in it) for k, _, it in asm_blocks(body))
        if not has_asm_load and not wanted:
            continue
        problems, stats = audit_kernel(body, start, meta)
        # a spill inside the innermost loop that holds the asm loads would be paid per tile: report where scratch sits
        spans = loop_spans(body, start)
        load_lines = [start + 1 + i for k, i, it in asm_blocks(body) if k == "asm" and any(b.startswith("ds_read") for b in it)]
        inner = [sp for sp in spans if load_lines and sp[0] <= load_lines[0] and load_lines[-1] <= sp[1]]
        inner = min(inner, key=lambda sp: sp[1] - sp[0]) if inner else None
        stats["tile_loop"] = inner
        stats["scratch_in_tile_loop"] = [n for n in stats["scratch_lines"] if inner and inner[0] <= n <= inner[1]]
        res[name] = (problems, stats)
    return res


if __name__ == "__main__":
    bad = 0
    for name, (problems, stats) in audit_file(sys.argv[1], sys.argv[2:]).items():
        short = {k: v for k, v in stats.items() if k != "scratch_lines"}
        short["scratch_accesses"] = len(stats["scratch_lines"])
        print(f"{name}: {short}")
        for p in problems[:40]:
            print("   !!", p)
        bad += len(problems)
    print("problems:", bad)
    sys.exit(1 if bad else 0)
