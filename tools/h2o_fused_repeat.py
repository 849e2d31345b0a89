"""Run-to-run check of the fused H2O kernel against round 2's materialising kernels (debug_stage_mask bit 11): twelve repeats of four
shapes, fp16 and bf16.  This is the script that showed the fp16 path's write-after-read on the K ring (one head off by a unit in the
last place in one run of five) before tests/test_gpu_parity.py::test_h2o_fused_same_bits_every_run pinned it."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kvcache_factory_amd import _kvc as kvc, synth
dev = torch.device("cuda:0")
for rep in range(12):
  for dtype, hq, hkv, L, W, D, seed in [(torch.float16, 8, 2, 512, 32, 128, 900 + 512), (torch.float16, 8, 2, 512, 8, 128, 1412), (torch.float16, 2, 2, 512, 32, 128, 1412), (torch.bfloat16, 8, 2, 512, 32, 128, 1412)]:
    q, k, v = synth.make_qkv(hq, hkv, L, D, dtype, seed, device=dev)
    f = kvc.scores(kvc.H2O, q, k, W, 7, None)[0].float().cpu()
    l = kvc.scores(kvc.H2O, q, k, W, 7, None, debug_mask=2048)[0].float().cpu()
    d = (f != l)
    print(rep, dtype, hq, L, W, D, "mismatch", int(d.sum()), "of", d.numel(), "heads", sorted(set(d.nonzero()[:, -2].tolist())) if d.any() else [], "cols", d.nonzero()[:12, -1].tolist())
    if d.any():
        ix = d.nonzero()[0]; print("   first", ix.tolist(), float(f[tuple(ix)]), float(l[tuple(ix)]))
