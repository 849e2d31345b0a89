"""Times kvc_merge_pivot at the 8k -> 128 configuration (32 query heads over 8 KV heads, bf16): whole call and per kernel
(rocprofv3 --kernel-trace --stats -- python3 tools/merge_timing.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kvcache_factory_amd import _kvc, synth
dev = torch.device("cuda:0")
q, k, v = synth.make_qkv(32, 8, 8000, 128, torch.bfloat16, 0, device=dev)
t0 = time.time()
sc = _kvc.scores(_kvc.SNAPKV, q, k, 8, 7, "maxpool")
idx = _kvc.select(sc, 120, "torch_cpu")
torch.cuda.synchronize(); print("scores + select (first call): %.3f s" % (time.time() - t0))
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    ko, vo = _kvc.merge_pivot(_kvc.SNAPKV, k, v, idx, 8)
    torch.cuda.synchronize(); print("merge_pivot call %d: %.3f ms" % (it, (time.time() - t0) * 1e3))
