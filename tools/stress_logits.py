"""Stress the exact bf16 K-scan (hand-issued ds_read_u16_d16_hi operands): pooled scores of a 32-layer batch launch
(21 tiles per wave) against the per-call path (one tile per wave) and, for layer 0, against the oracle, repeated.
Any rare operand hazard would show as run-to-run different score bits."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kvcache_factory_amd import _kvc as kvc, synth
dev = torch.device("cuda:0")
L, W, layers, reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8000, 8, 32, int(sys.argv[2]) if len(sys.argv) > 2 else 10
qkv = [synth.make_qkv(32, 8, L, 128, torch.bfloat16, 10 + l, device=dev) for l in range(layers)]
ref = [kvc.scores(kvc.SNAPKV, q, k, W, 7, "maxpool")[0].clone() for q, k, v in qkv]          # per-call path
bp = kvc.BatchPlan(kvc.SNAPKV, qkv, W, 120, 7, "maxpool", "canonical", want_indices=True)
sc = [torch.empty(32, L - W, dtype=torch.bfloat16, device=dev) for _ in range(layers)]
arr = (ctypes.c_void_p * layers)(*[t.data_ptr() for t in sc])
bad_total = 0
for rep in range(reps):
    for t in sc:
        t.zero_()
    rc = bp.call(scores_out=arr)
    assert rc == 0, kvc.lib().kvc_last_error()
    torch.cuda.synchronize()
    bad = sum(int((a.view(torch.int16) != b.view(torch.int16)).sum()) for a, b in zip(sc, ref))
    again = [kvc.scores(kvc.SNAPKV, q, k, W, 7, "maxpool")[0] for q, k, v in qkv[:4]]
    bad2 = sum(int((a.view(torch.int16) != b.view(torch.int16)).sum()) for a, b in zip(again, ref[:4]))
    bad_total += bad + bad2
    print(f"rep {rep}: batch-vs-call differing scores {bad} of {layers * 32 * (L - W)}; call-vs-call {bad2}", flush=True)
try:
    from oracle import kvc_oracle as O
    q, k, v = [t.cpu() for t in qkv[0]]
    so = O.scores(q, k, W, 7, "maxpool", dot_mode=O.DOT_CHAIN, sum_mode=O.SUM_TORCH16)
    print("layer 0 vs oracle:", int((ref[0].cpu().view(torch.int16) != so.view(torch.int16)).sum()), "differing")
except Exception as e:
    print("oracle check skipped:", e)
print("TOTAL differing:", bad_total)
sys.exit(1 if bad_total else 0)
