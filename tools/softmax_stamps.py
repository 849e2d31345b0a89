"""Phase breakdown of softmax_pool_kernel from the -DKVC_STAMPS diagnostic build (tools/build_diag.sh stamps): one 32-layer
batched launch with debug_stage_mask = 6; wave 0 of the 32 workgroups of item 0 stamps s_memtime (100 MHz constant clock)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KVC_LIB_PATH"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag", "libkvc_hip_stamps.so")
import torch, bench
from kvcache_factory_amd import _kvc
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["c2"]; ks = bench.layer_budgets(cfg)
p = bench.Prompt(cfg, dev, "torch_cpu", "batch", bench.make_inputs(cfg, dev, 0), ks)
p.run(); torch.cuda.synchronize()
pp = _kvc.Params.from_buffer_copy(p.bp.p); pp.debug_stage_mask = 6
for _ in range(3):
    p.bp.idx[0].zero_()
    assert p.bp.call(params=pp, stream=p.handle) == 0
    torch.cuda.synchronize()
t = p.bp.idx[0].view(-1)[:32 * 8].view(32, 8).cpu().double()
names = ["issue logit loads + tile maxima -> row max (2 barriers)", "wait logits + exponentials", "stage + chain adds (8 stages) + rinv",
         "window sums -> LDS + barrier", "pooling + stores"]
for i, nm in enumerate(names):
    d = (t[:, i + 1] - t[:, i]) / 2.1           # s_memtime counts core cycles (~2.1 GHz): ns
    print(f"  {nm:55s} mean {d.mean() / 1e3:7.2f} us   max {d.max() / 1e3:7.2f} us")
print("  second stage: barrier -> chain add done: %.0f cycles; stamp2 -> second stage barrier: %.0f cycles" % (
    float((t[:, 7] - t[:, 6]).mean()), float((t[:, 6] - t[:, 2]).mean())))
print("  workgroup total mean %.2f us; start spread %.2f us" % (float(((t[:, 5] - t[:, 0]) / 2.1).mean() / 1e3), float((t[:, 0].max() - t[:, 0].min()) / 2.1 / 1e3)))
