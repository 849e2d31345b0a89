import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from kvcache_factory_amd import _kvc as kvc, synth
dev = torch.device("cuda:0")
for dtype, hq, hkv, L, W, D in [(torch.bfloat16, 1, 1, 40, 8, 128), (torch.float16, 1, 1, 40, 8, 128), (torch.bfloat16, 1, 1, 64, 8, 128), (torch.bfloat16, 1, 1, 512, 8, 128), (torch.bfloat16, 1, 1, 40, 8, 64)]:
    q, k, v = synth.make_qkv(hq, hkv, L, D, dtype, 5100 + L, device=dev)
    f = kvc.scores(kvc.H2O, q, k, W, 7, None)[0].float().cpu()
    l = kvc.scores(kvc.H2O, q, k, W, 7, None, debug_mask=2048)[0].float().cpu()
    d = (f != l)
    print(dtype, L, D, "mismatch", int(d.sum()), "of", d.numel(), "cols", d.nonzero()[:20, -1].tolist(), "max rel", float(((f - l).abs() / l.abs().clamp_min(1e-9)).max()))
    if d.any():
        j = int(d.nonzero()[0, -1]); print("   first", j, float(f.flatten()[j]), float(l.flatten()[j]))
