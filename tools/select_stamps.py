"""Phase breakdown of select_exact_kernel (nth_element + sort regime) from the -DKVC_STAMPS diagnostic build
(tools/build_diag.sh stamps): one batched launch with debug_stage_mask = 32 (scores of the previous run, select only — the
diag build skips the gather, the index output holds stamps).  Usage: python tools/select_stamps.py [c2|c4|c5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KVC_LIB_PATH"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag", "libkvc_hip_stamps.so")
import torch, bench
from kvcache_factory_amd import _kvc
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = bench.CONFIGS[name]; ks = bench.layer_budgets(cfg)
p = bench.Prompt(cfg, dev, "torch_cpu", "batch", bench.make_inputs(cfg, dev, 0), ks)
pp = _kvc.Params.from_buffer_copy(p.bp.p); pp.debug_stage_mask = 1 | 2 | 4
# scores of a full scoring run (mask 7 = every scoring stage, no select), then the stamped select
assert p.bp.call(params=pp, stream=p.handle) == 0
torch.cuda.synchronize()
pp.debug_stage_mask = 32
for _ in range(2):
    assert p.bp.call(params=pp, stream=p.handle) == 0
    torch.cuda.synchronize()
H = bench.HQ
for li in sorted(set([0, len(ks) // 2, len(ks) - 1])):
    k = ks[li]
    if k < 20 or k * 64 <= cfg["L"] - cfg["W"]:
        print(f"layer {li}: k = {k}: partial_sort regime, no stamps"); continue
    t = p.bp.idx[li].view(-1)[: H * k].view(H, k)[:, :20].cpu().double()
    us = lambda a, b: float(((t[:, b] - t[:, a]) / 2.1e3).mean())
    print(f"layer {li}: k = {k}, n = {cfg['L'] - cfg['W']}: nth_element {us(0, 1):7.1f} us ({float(t[:, 11].mean()):.1f} large + "
          f"{float(t[:, 10].mean()):.1f} small partitions) | copy {us(1, 2):6.1f} | sort loop {us(2, 3):7.1f} us "
          f"({float((t[:, 9] - t[:, 11]).mean()):.1f} large + {float((t[:, 8] - t[:, 10]).mean()):.1f} small partitions) | "
          f"final placement {us(3, 4):6.1f} us | total {us(0, 4):7.1f} us (max {float(((t[:, 4] - t[:, 0]) / 2.1e3).max()):.1f})")
    acc = lambda i: float((t[:, i] / 2.1e3).mean())
    print(f"          large partitions of nth_element: median-of-three {acc(19):6.1f} us, scan {acc(16):6.1f}, bisection {acc(17):6.1f}, swaps {acc(18):6.1f}"
          f" | of the sort: median {acc(15) - acc(19):6.1f}, scan {acc(12) - acc(16):6.1f}, bisection {acc(13) - acc(17):6.1f}, swaps {acc(14) - acc(18):6.1f}")
