"""softmax_pool_kernel, the variant that keeps the exponentials in registers (KEEP = 8 at 8k, 94 VGPRs, one workgroup per CU)
against the one that reloads the logits and recomputes them in the second pass (KEEP = 0, 57 VGPRs, two per CU; debug_stage_mask
bit8): measured equal (97 vs 94 us per 32-layer launch) — the stage is bound by its ~20 barrier-separated phases per head, not by
VALU work, registers or the logit reload; de-phasing co-resident workgroups by s_sleep did not move it either."""
import sys, ctypes
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from kvcache_factory_amd import _kvc
dev=torch.device('cuda:0')
cfg=bench.CONFIGS['c2']; ks=bench.layer_budgets(cfg)
p=bench.Prompt(cfg,dev,'torch_cpu','batch',bench.make_inputs(cfg,dev,0),ks)
p.run(); torch.cuda.synchronize()
def timed(mask,reps=20):
    pp=_kvc.Params.from_buffer_copy(p.bp.p); pp.debug_stage_mask=mask
    p.bp.call(params=pp,stream=p.handle); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): p.bp.call(params=pp,stream=p.handle)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/reps
print("softmax_pool KEEP=8 (default): %.1f us"%timed(6))
print("softmax_pool KEEP=0 (reload)  : %.1f us"%timed(6|256))
print("logits: %.1f us"%timed(1))
