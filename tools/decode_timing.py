"""us per decoded token per layer, Llama-3-8B head geometry (32 query heads over 8 KV heads, D 128, bf16):
kvc_decode_step (in-place tail append + one attention kernel over prefix + tail) beside the reference-shaped torch path on the
same tensors (repeat_kv of the new token, torch.cat onto the [1, 32, P + T, 128] cache, SDPA: llama_model.py:287-289, 306-313),
and kvc_ragged_decode_step on an AdaKV 8k -> 128 cache beside a torch stand-in for update_flatten_view (csrc/csrc/cuda_api.cu:
12-85: a copy of the whole flattened cache with one row inserted per head) + padded attention (flash_attn_varlen is absent).
HIP events over `reps` calls on torch's current stream (the library enqueues there).  Prints one JSON object."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from kvcache_factory_amd import _kvc, synth
dev = torch.device("cuda:0")
HQ, HKV, D, G = 32, 8, 128, 4
dt = torch.bfloat16
scaling = D ** -0.5


def per_call_us(fn, reps=200, warm=10):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


out = {"geometry": "H_q 32, H_kv 8, D 128, bf16, bsz 1, one new token", "unit": "us per token per layer", "decode_step": [], "ragged": []}
g = torch.Generator(device="cpu").manual_seed(0)
for P in (128, 2056):
    for T in (0, 256, 1024):
        kp = torch.randn(1, HQ, P + 256, D, generator=g).to(dt).to(dev); vp = torch.randn(1, HQ, P + 256, D, generator=g).to(dt).to(dev)
        kt = torch.randn(1, HKV, T + 64, D, generator=g).to(dt).to(dev); vt = torch.randn(1, HKV, T + 64, D, generator=g).to(dt).to(dev)
        q = torch.randn(1, HQ, 1, D, generator=g).to(dt).to(dev)
        kn = torch.randn(1, HKV, 1, D, generator=g).to(dt).to(dev); vn = torch.randn(1, HKV, 1, D, generator=g).to(dt).to(dev)
        fused = lambda: _kvc.decode_step(q, kn, vn, kp, vp, P, kt, vt, T, scaling)
        # the reference's cache: [1, 32, P + T, 128], expanded heads
        kc = torch.cat([kp[:, :, :P], kt[:, :, :T].repeat_interleave(G, 1)], 2).contiguous(); vc = torch.cat([vp[:, :, :P], vt[:, :, :T].repeat_interleave(G, 1)], 2).contiguous()

        def ref_shaped():
            kx = kn[:, :, None].expand(1, HKV, G, 1, D).reshape(1, HQ, 1, D); vx = vn[:, :, None].expand(1, HKV, G, 1, D).reshape(1, HQ, 1, D)
            K = torch.cat([kc, kx], 2); V = torch.cat([vc, vx], 2)                 # DynamicCache.update: a copy of the whole cache
            return F.scaled_dot_product_attention(q, K, V, is_causal=False)
        o1 = fused(); o2 = ref_shaped().transpose(1, 2)
        err = float((o1.float() - o2.float()).abs().max())
        tf, tr = per_call_us(fused), per_call_us(ref_shaped)
        out["decode_step"].append({"prefix_rows": P, "tail_rows": T, "kvc_decode_step_us": round(tf, 2), "repeat_kv_cat_sdpa_us": round(tr, 2),
                                   "speedup": round(tr / tf, 2), "max_abs_diff": err})
        print(out["decode_step"][-1], flush=True)
# AdaKV 8k -> 128 (ragged per-head budgets)
q8, k8, v8 = synth.make_qkv(HQ, HKV, 8000, D, dt, 0, device=dev)
r = _kvc.ragged_compress(_kvc.ADAKV, q8, k8, v8, 8, 120, 7, "maxpool", 0.2, True, slack=256)
lens = r["lens"]
q = torch.randn(1, HQ, 1, D, generator=g).to(dt).to(dev)
kn = torch.randn(1, HKV, 1, D, generator=g).to(dt).to(dev); vn = torch.randn(1, HKV, 1, D, generator=g).to(dt).to(dev)
fused = lambda: _kvc.ragged_decode_step(q, kn, vn, r["k_flat"], r["v_flat"], r["seg_off"], r["seg_len"], 0, scaling)
# stand-in for the reference: dense flattened cache (no slack), rebuilt per token with one row inserted per head, then attention
offs = r["seg_off"].tolist()
kd = torch.cat([r["k_flat"][o:o + n] for o, n in zip(offs, lens)]); vd = torch.cat([r["v_flat"][o:o + n] for o, n in zip(offs, lens)])
tot = sum(lens); mx = max(lens) + 1
cu = torch.tensor([0] + lens).cumsum(0)
old_pos = torch.cat([torch.arange(lens[h]) + cu[h] + h for h in range(HQ)]).to(dev)          # where every old row goes
new_pos = (cu[1:] + torch.arange(HQ)).to(dev)                                                  # where the new rows go
pad_idx = torch.stack([torch.cat([torch.arange(lens[h] + 1) + cu[h] + h, torch.full((mx - lens[h] - 1,), 0)]) for h in range(HQ)]).to(dev)
pad_mask = torch.stack([torch.arange(mx) < lens[h] + 1 for h in range(HQ)]).to(dev)[None, :, None, :]


def ref_ragged():
    kx = kn[0, :, 0].repeat_interleave(G, 0); vx = vn[0, :, 0].repeat_interleave(G, 0)
    K = torch.empty(tot + HQ, D, dtype=dt, device=dev); V = torch.empty_like(K)
    K[old_pos] = kd; K[new_pos] = kx; V[old_pos] = vd; V[new_pos] = vx                          # update_flatten_view: O(cache) per token
    return F.scaled_dot_product_attention(q, K[pad_idx][None], V[pad_idx][None], attn_mask=pad_mask)


o1 = fused(); o2 = ref_ragged().transpose(1, 2)
out["ragged"].append({"cache": "AdaKV 8k -> 128 (base), 32 heads", "rows_total": tot, "longest_head": max(lens), "kvc_ragged_decode_step_us": round(per_call_us(fused), 2),
                      "flatten_copy_plus_padded_sdpa_us": round(per_call_us(ref_ragged), 2), "max_abs_diff": float((o1.float() - o2.float()).abs().max())})
print(json.dumps(out))
