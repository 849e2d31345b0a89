"""Latency of the exact-tie top-k against the number of heads in flight (n = 31 992): the nth_element heads keep their node arrays in the
workspace (256 KB per head), so 1 024 concurrent heads no longer fit the L2s and every random access pays memory latency — k = 3 978:
772 us with 32 heads, 1 204 us with 1 024.  That, not the 9-level heap, sets the C5 launch after round 3."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from kvcache_factory_amd import _kvc as kvc
dev = torch.device("cuda:0")
n = 31992
g = torch.Generator().manual_seed(0)
for heads in (32, 256, 1024):
    sc = (torch.rand(heads, n, generator=g) * 0.01).to(torch.bfloat16).to(dev)[None].contiguous()
    for k in (478, 2000, 3978):
        for _ in range(2):
            kvc.select(sc, k, "torch_cpu")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            kvc.select(sc, k, "torch_cpu")
        torch.cuda.synchronize()
        print(f"heads {heads:5d} k {k:5d} {(time.perf_counter() - t0) / 3 * 1e6:9.1f} us")
