"""Build audit of the hand-issued LDS operand loads (kvc_ldsasm.h: ld_step / wait_step, ld_b_step / wait_b_step).

hipcc does not track inline-asm loads: between such a load and the wait that retires it the compiler must not read,
copy, spill or reuse a destination register, and a destination must never share the address register of its own
statement (round 1's "rare, run-to-run different" H2O logit errors — DESIGN.md §4).  Constraints in the source pin most
of it; register allocation can only be checked in the listing, so this test disassembles the kernels of every build
(hipcc -S cross-compiles without a GPU) and applies tools/asm_audit.py.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_audit  # noqa: E402

CSRC = os.path.join(ROOT, "kvcache_factory_amd", "csrc")


def _listing(name, diag=""):
    out = os.path.join("isa", name + (".diag" if diag else "") + ".s")
    if diag:                                           # a differently named target: built from the same rule by hand
        flags = subprocess.check_output(["make", "-C", CSRC, "-s", "--eval", "pf:\n\t@echo $(HIPCC) $(HIPFLAGS)", "pf"], text=True).split()
        os.makedirs(os.path.join(CSRC, "isa"), exist_ok=True)
        subprocess.check_call(flags + [diag, "-S", "--cuda-device-only", name + ".hip", "-o", out], cwd=CSRC,
                              stderr=subprocess.DEVNULL)
    else:
        subprocess.check_call(["make", "-C", CSRC, "-s", os.path.join("isa", name + ".s")], stderr=subprocess.DEVNULL)
    return os.path.join(CSRC, out)


@pytest.fixture(scope="module")
def score_audit():
    return asm_audit.audit_file(_listing("kvc_score"))


@pytest.fixture(scope="module")
def h2o_audit():
    return asm_audit.audit_file(_listing("kvc_h2o"))


def test_logits_kernel_asm_loads_are_clean(score_audit):
    """Every exact bf16 K-scan instantiation (D in {64,128} x W in {8,16,32,64,runtime}): between each asm load block and
    its wait only v_mfma / scalar instructions, no mention of a pending destination, the wait names the very registers
    the loads wrote, and the wait count leaves exactly the younger loads outstanding."""
    ks = {k: v for k, v in score_audit.items() if "logits_kernel" in k}
    assert len(ks) == 10, sorted(ks)
    for name, (problems, stats) in ks.items():
        assert not problems, (name, problems[:5])
        d = 128 if "Li128E" in name else 64
        assert stats["asm_load_blocks"] == stats["asm_wait_blocks"] == d // 8 and stats["asm_loads"] == 5 * (d // 8)


def test_logits_mt4_kernel_asm_loads_are_clean(score_audit):
    """The four-M-tile scan (G * W == 128: four waves share one staged K tile) uses the same hand-issued loads."""
    ks = {k: v for k, v in score_audit.items() if "logits_mt4_kernel" in k and "ILi0E" in k and "Lb0E" in k}   # bf16 exact
    assert len(ks) == 6, sorted(ks)                     # D in {64, 128} x W in {16, 32, 64}
    for name, (problems, stats) in ks.items():
        assert not problems, (name, problems[:5])
        d = 128 if "Li128E" in name else 64
        assert stats["asm_load_blocks"] == stats["asm_wait_blocks"] == d // 8 and stats["asm_loads"] == 5 * (d // 8)
        assert stats["ScratchSize"] == 0 and stats["Occupancy"] >= 2


def test_headline_kernel_registers_and_scratch(score_audit):
    """logits_kernel<bf16, 128, W=8, exact> — the headline's K scan: 3 waves per SIMD (<= 168 VGPRs); the few bytes of
    scratch it has (a prologue spill of the Q-image loop) stay outside the tile loop."""
    (name, (problems, stats)), = [(k, v) for k, v in score_audit.items() if "logits_kernelILi0ELi128ELi8ELb0E" in k]
    assert stats["NumVgprs"] <= 168 and stats["Occupancy"] == 3
    assert stats["ScratchSize"] <= 64
    assert stats["tile_loop"] is not None and stats["scratch_in_tile_loop"] == []
    for w in (16, 32, 64):                              # the other compile-time windows likewise
        (_, (_, st)), = [(k, v) for k, v in score_audit.items() if f"logits_kernelILi0ELi128ELi{w}ELb0E" in k]
        assert st["scratch_in_tile_loop"] == [] and st["Occupancy"] == 3


def test_h2o_kernel_asm_loads_are_clean(h2o_audit):
    ks = {k: v for k, v in h2o_audit.items() if "h2o_logits_kernel" in k}
    assert len(ks) == 2, sorted(ks)                     # bf16, D = 128 and 64
    for name, (problems, stats) in ks.items():
        # compiler VALU work between a load and its wait is harmless as long as it names no pending register; the
        # H2O kernel's epilogue of the previous tile may be scheduled there
        hard = [p for p in problems if "between an asm load and its wait" not in p]
        assert not hard, (name, hard[:5])
        assert stats["ScratchSize"] == 0


def test_h2o_fused_kernel_asm_loads_are_clean(h2o_audit):
    """h2o_fused_kernel<bf16, D> (round 3): its MFMA operands come from hand-issued ds_read_b128 / ds_read_u16_d16_hi out of an
    image that hand-issued global_load_lds statements fill.  Between a load block and its wait: MFMAs and scalar
    instructions only; 124-128 VGPRs (16 waves per CU hold 16 query rows of the logit matrix in registers) and no scratch
    at all — one spilled vector of logits inside the tile loop cost 8 scratch loads and stores per tile in the first build."""
    ks = {k: v for k, v in h2o_audit.items() if "h2o_fused_kernelILi0E" in k}
    assert len(ks) == 2, sorted(ks)                     # bf16, D = 128 and 64
    for name, (problems, stats) in ks.items():
        assert not problems, (name, problems[:5])
        d = 128 if "Li128E" in name else 64
        assert stats["asm_load_blocks"] == stats["asm_wait_blocks"] == d // 16 and stats["asm_loads"] == 5 * (d // 16)
        assert stats["NumVgprs"] <= 128 and stats["ScratchSize"] == 0 and stats["Occupancy"] == 4


def test_audit_sees_round1_hazard():
    """The same H2O kernel built with plain "=v" outputs (-DKVC_DIAG_NO_EARLYCLOBBER): hipcc gives the last step's first
    destination the address register (`ds_read_u16_d16_hi vN, vN` followed by loads addressed through vN) — the cause
    of round 1's non-deterministic logits.  The audit must flag it; the shipped build must not have it."""
    res = asm_audit.audit_file(_listing("kvc_h2o", diag="-DKVC_DIAG_NO_EARLYCLOBBER"))
    found = [p for _, (problems, _) in res.items() for p in problems if "overlaps its address register" in p]
    assert found, "the diagnostic build no longer shows the hazard (compiler changed?): the audit is unproven"


def test_audit_rules_on_synthetic_listings():
    """Each rule fires on a minimal listing: a copy of a pending register, a scratch spill, a wrong count, a renamed
    destination at the wait, a branch target while loads are pending."""
    def run(mid, wait="s_waitcnt lgkmcnt(0) ; retire v5 v6"):
        body = ["\t;;#ASMSTART", "\tds_read_u16_d16_hi v5, v9 offset:0", "\tds_read_u16_d16_hi v6, v9 offset:4", "\t;;#ASMEND",
                *mid, "\t;;#ASMSTART", "\t" + wait, "\t;;#ASMEND", "\tv_mfma_f32_32x32x2_f32 v[10:25], v1, v5, v[10:25]", ".Lfunc_end0:"]
        return asm_audit.audit_kernel(body)[0]
    assert run(["\tv_mfma_f32_32x32x2_f32 v[10:25], v1, v2, v[10:25]", "\ts_nop 0"]) == []
    assert any("names pending" in p for p in run(["\tv_mov_b32_e32 v30, v5"]))
    assert any("between an asm load and its wait" in p for p in run(["\tscratch_store_dword off, v40, off"]))
    assert any("names pending" in p for p in run(["\tv_mfma_f32_32x32x2_f32 v[10:25], v1, v6, v[10:25]"]))
    assert any("no pending asm load writes" in p for p in run([], "s_waitcnt lgkmcnt(0) ; retire v5 v7"))
    assert any("may not have landed" in p for p in run([], "s_waitcnt lgkmcnt(1) ; retire v5 v6"))
    assert any("label" in p for p in run([".LBB0_3:"]))
    assert any("end of kernel with pending" in p for p in run([], "s_nop 0"))
    body = ["\t;;#ASMSTART", "\tds_read_u16_d16_hi v9, v9 offset:0", "\tds_read_u16_d16_hi v6, v9 offset:4", "\t;;#ASMEND",
            "\t;;#ASMSTART", "\ts_waitcnt lgkmcnt(0) ; retire v9 v6", "\t;;#ASMEND", ".Lfunc_end0:"]
    probs = asm_audit.audit_kernel(body)[0]
    assert any("overlaps its address register" in p for p in probs) and any("addresses through pending" in p for p in probs)
