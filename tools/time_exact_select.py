"""Latency of the exact-tie select (one wave per head) by budget k at n = 31992 and 7992: which libstdc++ regime costs what."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kvcache_factory_amd import _kvc as kvc
dev = torch.device("cuda:0")
for n in (7992, 31992):
    g = torch.Generator().manual_seed(0)
    sc = (torch.rand(32, n, generator=g) * 0.01).to(torch.bfloat16).to(dev)[None].contiguous()
    for k in (64, 120, 128, 129, 250, 499, 500, 1000, 2040, 4000):
        if k > n:
            continue
        for _ in range(2):
            kvc.select(sc, k, "torch_cpu")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            kvc.select(sc, k, "torch_cpu")
        torch.cuda.synchronize()
        regime = "partial_sort" if k * 64 <= n else "nth_element+sort"
        print(f"n {n:6d} k {k:5d} {regime:18s} {(time.perf_counter() - t0) / 5 * 1e6:9.1f} us")
