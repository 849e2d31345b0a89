"""SURVEY 8f N2 as a measured prototype: scoring fused into the prefill attention, in the one variant that does not need a new
flash-attention forward for the whole prompt — the library's SDPA for the query rows < L - 128 and an OWN path
for the last 128 rows that reads K once and V once, produces their attention output AND exports the last W rows' logits to the
compression (so the separate K scan of kvc_scores disappears).  The own path here is built from library GEMMs (hipBLASLt through
torch.matmul: S = Q_last K^T per KV head with its four query heads stacked, fp32 softmax, O = P V) — what a hand-written kernel
would have to beat, and already at bf16-MFMA arithmetic (the `mfma16` tolerance class, not the exact fmaf chain).
Per prefill layer, Llama-3-8B head geometry, bf16:
  without N2:  SDPA(all L rows)                      + K scan of the scoring stage (logits_kernel, amortised over a 32-layer batch)
  with N2:     SDPA(rows < L - 128) + own tail path  (no K scan)
and the agreement of the pooled scores computed from the exported logits with the exact scores (kvc_scores).
Usage: python tools/n2_probe.py [8000 32000]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from kvcache_factory_amd import _kvc, synth
dev = torch.device("cuda:0")
HQ, HKV, D, G, W, TAIL = 32, 8, 128, 4, 8, 128


def us(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


out = []
for L in [int(x) for x in sys.argv[1:]] or [8000, 32000]:
    q, k, v = synth.make_qkv(HQ, HKV, L, D, torch.bfloat16, 0, device=dev)           # q [1,32,L,128] (model layout), k / v [1,8,L,128]
    qc = q.contiguous()
    scale = D ** -0.5
    full = lambda: F.scaled_dot_product_attention(qc, k, v, is_causal=True, enable_gqa=True)
    head = lambda: F.scaled_dot_product_attention(qc[:, :, :L - TAIL], k[:, :, :L - TAIL], v[:, :, :L - TAIL], is_causal=True, enable_gqa=True)
    causal = torch.full((TAIL, TAIL), float("-inf"), device=dev).triu(1)

    def tail():
        qt = qc[:, :, L - TAIL:].reshape(1, HKV, G * TAIL, D)                         # the four query heads of a KV head stacked: K read once
        S = torch.matmul(qt, k.transpose(2, 3)) * scale                               # [1, 8, 512, L] bf16 logits
        S4 = S.view(1, HKV, G, TAIL, L)
        S4[..., L - TAIL:] += causal.to(S.dtype)
        P = torch.softmax(S4.float(), dim=-1).to(S.dtype).view(1, HKV, G * TAIL, L)
        O = torch.matmul(P, v)                                                        # V read once
        return O.view(1, HQ, TAIL, D), S4[..., TAIL - W:, :]                          # + the exported last-W-row logits [1,8,4,W,L]
    o_full = full()
    o_head = head()
    o_tail, lg = tail()
    err = float((torch.cat([o_head, o_tail], 2).float() - o_full.float()).abs().max())
    t_full, t_head, t_tail = us(full), us(head), us(tail)
    # the K scan it would replace: the logits stage of a 32-layer batched scoring call, per layer (debug_stage_mask bit 0)
    qkv = [(q, k, v)] * 32 if L <= 8000 else [(q, k, v)] * 8
    bp = _kvc.BatchPlan(_kvc.SNAPKV, qkv, W, [120] * len(qkv), 7, "maxpool", "torch_cpu")
    pp = _kvc.Params.from_buffer_copy(bp.p); pp.debug_stage_mask = 1
    t_scan = us(lambda: bp.call(params=pp)) / len(qkv)
    # agreement of the scores: the reference's op sequence (:326-333) on the exported logits vs the exact scores
    lgx = lg.reshape(1, HQ, W, L)
    p = torch.softmax(lgx.float(), dim=-1).to(torch.bfloat16)
    sc_n2 = F.max_pool1d(p[..., :L - W].sum(dim=-2), kernel_size=7, padding=3, stride=1)
    sc_ex = _kvc.scores(_kvc.SNAPKV, q, k, W, 7, "maxpool")
    d = (sc_n2.view(torch.int16).int() - sc_ex.view(torch.int16).int()).abs()
    i_n2, i_ex = _kvc.select(sc_n2.contiguous(), 120, "torch_cpu")[0].cpu(), _kvc.select(sc_ex, 120, "torch_cpu")[0].cpu()
    overlap = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(i_n2, i_ex)) / i_ex.numel()
    r = {"L": L, "sdpa_all_rows_us": round(t_full, 1), "k_scan_per_layer_us": round(t_scan, 1), "without_n2_us": round(t_full + t_scan, 1),
         "sdpa_rows_below_L-128_us": round(t_head, 1), "own_tail_path_us": round(t_tail, 1), "with_n2_us": round(t_head + t_tail, 1),
         "n2_gain_us": round(t_full + t_scan - t_head - t_tail, 1), "attention_output_max_abs_diff": err,
         "scores_differing_frac": float((d > 0).float().mean()), "scores_max_ulp": int(d.max()), "topk_index_overlap": round(overlap, 4)}
    print(r, flush=True)
    out.append(r)
    del q, k, v, qc, o_full, o_head, o_tail, lg, bp
    torch.cuda.empty_cache()
print(json.dumps(out))
