"""Determinism of the H2O logits with hand-issued d16 operand loads, shipped build vs the round-1 hazard.

    python tools/h2o_d16_stress.py [L] [reps]

Runs the H2O scoring (kvc_scores, method H2O) `reps` times with the shipped library and with
tools/diag/libkvc_hip_noearlyclobber.so (tools/build_diag.sh noearlyclobber: asm load outputs "=v" instead of "=&v", so
hipcc allocates a destination onto the address register — kvc_ldsasm.h (1)), and compares the materialised logit
matrices S [Hq][L][L] bit for bit with the first run of the shipped build.  Expectation: shipped = 0 differing logits in
every run; the diagnostic build shows rare, run-to-run different errors (timing dependent: it may also show none).
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kvcache_factory_amd import _kvc as kvc, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
HQ, HKV, W = 8, 2, 8
dev = torch.device("cuda:0")
q, k, v = synth.make_qkv(HQ, HKV, L, 128, torch.bfloat16, 4242, device=dev)


def logits_once():
    p = kvc.make_params(kvc.H2O, q, k, None, W, 0, 7, None)
    nbytes = kvc.lib().kvc_workspace_bytes(ctypes.byref(p))
    ws = kvc.workspace(dev, nbytes)
    sc = torch.empty(1, HQ, L - W, dtype=q.dtype, device=dev)
    rc = kvc.lib().kvc_scores(ctypes.byref(p), kvc._ptr(q), kvc._ptr(k), kvc._ptr(sc), kvc._ptr(ws), nbytes, kvc._stream(dev))
    assert rc == 0, kvc.lib().kvc_last_error()
    offs = (ctypes.c_size_t * 3)()
    assert kvc.lib().kvc_workspace_layout(ctypes.byref(p), ctypes.byref(offs)) == 0
    torch.cuda.synchronize()
    return ws[offs[0]:offs[0] + HQ * L * L * 2].view(torch.int16).clone(), sc.view(torch.int16).clone()


def use(path):
    kvc._lib = None
    kvc.LIB_PATH = path
    kvc._workspaces.clear()
    kvc.lib()


total = {}
ref = None
for tag, path in (("shipped", os.path.join(ROOT, "kvcache_factory_amd", "libkvc_hip.so")),
                  ("noearlyclobber", os.path.join(ROOT, "tools", "diag", "libkvc_hip_noearlyclobber.so"))):
    if not os.path.exists(path):
        print(tag, "library missing:", path)
        continue
    use(path)
    total[tag] = 0
    for r in range(reps):
        S, sc = logits_once()
        if ref is None:
            ref = (S, sc)
        bad = int((S != ref[0]).sum())
        bads = int((sc != ref[1]).sum())
        total[tag] += bad
        where = ""
        if bad:
            i = (S != ref[0]).nonzero()[:3].flatten().tolist()
            where = " first at flat index %s: %s vs %s" % (i, [int(S[j]) for j in i], [int(ref[0][j]) for j in i])
        print(f"{tag} run {r}: {bad} of {S.numel()} logits differ from the shipped build's first run; {bads} scores differ{where}", flush=True)
print("TOTAL differing logits:", total)
sys.exit(1 if total.get("shipped", 0) else 0)
