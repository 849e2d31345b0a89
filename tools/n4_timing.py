"""Times the N4 entry points under rocprofv3 (--kernel-trace --stats): kvc_l2norm_compress at a few shapes and kvc_merge_pivot
at 8k -> 128 (32 query heads over 8 KV heads, bf16)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kvcache_factory_amd import _kvc, synth
dev = torch.device("cuda:0")


def timed(name, fn, reps=3):
    for it in range(reps):
        torch.cuda.synchronize(); t0 = time.time()
        fn()
        torch.cuda.synchronize(); dt = (time.time() - t0) * 1e3
    print(f"{name:60s} {dt:10.3f} ms (last of {reps})", flush=True)


q, k, v = synth.make_qkv(32, 8, 8000, 128, torch.bfloat16, 0, device=dev)
timed("l2norm 8k -> 128, 32 heads", lambda: _kvc.l2norm_compress(k, v, 128, n_q_heads=32))
timed("l2norm 8k -> 4096, 32 heads", lambda: _kvc.l2norm_compress(k, v, 4096, n_q_heads=32))
sc = _kvc.scores(_kvc.SNAPKV, q, k, 8, 7, "maxpool")
idx = _kvc.select(sc, 120, "torch_cpu")
timed("merge_pivot 8k -> 128, 32 heads", lambda: _kvc.merge_pivot(_kvc.SNAPKV, k, v, idx, 8))
q2, k2, v2 = synth.make_qkv(4, 2, 20000, 128, torch.bfloat16, 1, device=dev)
timed("l2norm 20000 -> 313, 4 heads (array in the workspace)", lambda: _kvc.l2norm_compress(k2, v2, 313, n_q_heads=4))
k2c = (k2.float() * 4).round().div(4).to(torch.bfloat16)
timed("l2norm 20000 -> 313, coarse keys (many equal norms)", lambda: _kvc.l2norm_compress(k2c, v2, 313, n_q_heads=4))
q3, k3, v3 = synth.make_qkv(8, 2, 32000, 128, torch.bfloat16, 2, device=dev)
timed("l2norm 32000 -> 4096, 8 heads", lambda: _kvc.l2norm_compress(k3, v3, 4096, n_q_heads=8))
