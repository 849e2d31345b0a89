"""ctypes binding of oracle/libkvc_oracle.so — the CPU restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package (kvcache_factory_amd/).  See kvc_oracle.cpp for the reference
file:line each function restates (pyramidkv/pyramidkv_utils.py:197-283, :306-347, :533-575, :595-620).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkvc_oracle.so")

BF16, FP16, FP32 = 0, 1, 2
POOL_NONE, POOL_AVG, POOL_MAX = 0, 1, 2
DOT_CHAIN, DOT_F64 = 0, 1
SUM_TORCH16 = 0
TIES_TORCH, TIES_CANON = 0, 1

_DTYPE = {torch.bfloat16: BF16, torch.float16: FP16, torch.float32: FP32}
_POOL = {None: POOL_NONE, "none": POOL_NONE, "avgpool": POOL_AVG, "maxpool": POOL_MAX}


class Params(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "dtype", "n_q_heads", "n_kv_heads", "q_len", "head_dim", "window", "k", "kernel_size",
        "pooling", "full_rows", "dot_mode", "sum_mode", "tie_mode", "n_threads")] + [
        (n, ctypes.c_int64) for n in (
            "q_stride_h", "q_stride_l", "k_stride_h", "k_stride_l", "v_stride_h", "v_stride_l")]


def build(force=False):
    """Compile the oracle with g++ (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "kvc_oracle.cpp"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libkvc_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i64p = ctypes.c_void_p, ctypes.c_void_p
        pp = ctypes.POINTER(Params)
        L.kvco_version.restype = ctypes.c_int
        L.kvco_scores.argtypes = [pp, vp, vp, vp, vp, vp, vp]
        L.kvco_topk.argtypes = [pp, vp, i64p, vp]
        L.kvco_gather.argtypes = [pp, vp, ctypes.c_int64, ctypes.c_int64, i64p, vp]
        L.kvco_compress.argtypes = [pp, vp, vp, vp, vp, vp, i64p, vp]
        L.kvco_streaming.argtypes = [pp, vp, vp, vp, vp, i64p]
        L.kvco_sort_desc.argtypes = [pp, vp, i64p, vp]
        L.kvco_adakv_caps.argtypes = [pp, vp, ctypes.c_int64, ctypes.c_double, ctypes.c_int, vp]
        L.kvco_ragged_gather.argtypes = [pp, vp, ctypes.c_int64, ctypes.c_int64, i64p, vp, vp]
        L.kvco_l2norm.argtypes = [pp, vp, vp, vp, i64p, vp, vp]
        L.kvco_merge_pivot.argtypes = [pp, vp, vp, i64p, vp, vp, vp, vp]
        L.kvco_think_prune.argtypes = [pp, vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, vp, vp, vp]
        L.kvco_pyramid_k.argtypes = [ctypes.c_int64] * 6
        L.kvco_pyramid_k.restype = ctypes.c_int64
        L.kvco_exp_u20.argtypes = [ctypes.c_float]
        L.kvco_exp_u20.restype = ctypes.c_float
        L.kvco_sum.argtypes = [vp, ctypes.c_int64, ctypes.c_int]
        L.kvco_sum.restype = ctypes.c_float
        L.kvco_f32_to_f16.argtypes = [ctypes.c_float]
        L.kvco_f32_to_f16.restype = ctypes.c_uint16
        L.kvco_f16_to_f32.argtypes = [ctypes.c_uint16]
        L.kvco_f16_to_f32.restype = ctypes.c_float
        L.kvco_f32_to_bf16.argtypes = [ctypes.c_float]
        L.kvco_f32_to_bf16.restype = ctypes.c_uint16
        _lib = L
    return _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _check_inner(t):
    assert t.device.type == "cpu" and t.stride(-1) == 1, "oracle wants CPU tensors, last dim contiguous"


def make_params(q, k, v, window, n_keep, kernel_size=5, pooling="avgpool", full_rows=False,
                dot_mode=DOT_CHAIN, sum_mode=SUM_TORCH16, tie_mode=TIES_TORCH, n_threads=0):
    """q: [1,Hq,L,D] (any h/l strides); k,v: [1,Hq or Hkv,L,D]."""
    assert q.dim() == 4 and q.shape[0] == 1, "bsz must be 1 (reference README.md:29)"
    _check_inner(q), _check_inner(k)
    p = Params()
    p.dtype = _DTYPE[q.dtype]
    p.n_q_heads, p.n_kv_heads = q.shape[1], k.shape[1]
    p.q_len, p.head_dim = q.shape[2], q.shape[3]
    p.window, p.k = window, n_keep
    p.kernel_size, p.pooling, p.full_rows = kernel_size, _POOL[pooling], int(full_rows)
    p.dot_mode, p.sum_mode, p.tie_mode, p.n_threads = dot_mode, sum_mode, tie_mode, n_threads
    p.q_stride_h, p.q_stride_l = q.stride(1), q.stride(2)
    p.k_stride_h, p.k_stride_l = k.stride(1), k.stride(2)
    if v is not None:
        _check_inner(v)
        p.v_stride_h, p.v_stride_l = v.stride(1), v.stride(2)
    return p


def _rc(code, what):
    if code != 0:
        raise RuntimeError(f"oracle {what} failed with code {code}")


def scores(q, k, window, kernel_size=5, pooling="avgpool", full_rows=False, want_intermediates=False, **modes):
    """A1-A5 (A10 with full_rows = True / 1; full_rows = 2: window MEAN, AdaKV / HeadKV).  Returns pooled scores [Hq, n]
    (+ logits, probs, wsum)."""
    p = make_params(q, k, None, window, 0, kernel_size, pooling, full_rows, **modes)
    H, L = q.shape[1], q.shape[2]
    n, R = L - window, (L if int(full_rows) == 1 else window)
    out = torch.empty(H, n, dtype=q.dtype)
    lg = pr = ws = None
    if want_intermediates:
        lg, pr = torch.empty(H, R, L, dtype=q.dtype), torch.empty(H, R, L, dtype=q.dtype)
        ws = torch.empty(H, n, dtype=q.dtype)
    _rc(lib().kvco_scores(ctypes.byref(p), _ptr(q), _ptr(k), _ptr(lg), _ptr(pr), _ptr(ws), _ptr(out)), "scores")
    return (out, lg, pr, ws) if want_intermediates else out


def topk(scores_t, n_keep, tie_mode=TIES_TORCH, n_threads=0):
    """A7 on pooled scores [H, n] -> (indices int64 [H,k], values [H,k])."""
    assert scores_t.dim() == 2 and scores_t.is_contiguous()
    p = Params()
    p.dtype = _DTYPE[scores_t.dtype]
    p.n_q_heads = p.n_kv_heads = scores_t.shape[0]
    p.window, p.head_dim = 1, 1
    p.q_len = scores_t.shape[1] + 1
    p.k, p.pooling, p.tie_mode, p.n_threads = n_keep, POOL_NONE, tie_mode, n_threads
    idx = torch.empty(scores_t.shape[0], n_keep, dtype=torch.int64)
    val = torch.empty(scores_t.shape[0], n_keep, dtype=scores_t.dtype)
    _rc(lib().kvco_topk(ctypes.byref(p), _ptr(scores_t), _ptr(idx), _ptr(val)), "topk")
    return idx, val


def gather(src, idx, window, n_q_heads):
    """A8 for one tensor: src [1,Hs,L,D], idx [Hq,k] -> [1,Hq,k+W,D]."""
    _check_inner(src)
    k = idx.shape[1]
    p = Params()
    p.dtype = _DTYPE[src.dtype]
    p.n_q_heads, p.n_kv_heads = n_q_heads, src.shape[1]
    p.q_len, p.head_dim, p.window, p.k, p.pooling = src.shape[2], src.shape[3], window, k, POOL_NONE
    out = torch.empty(1, n_q_heads, k + window, src.shape[3], dtype=src.dtype)
    _rc(lib().kvco_gather(ctypes.byref(p), _ptr(src), src.stride(1), src.stride(2), _ptr(idx.contiguous()),
                          _ptr(out)), "gather")
    return out


def compress(q, k, v, window, n_keep, kernel_size=5, pooling="avgpool", full_rows=False, **modes):
    """A0/A9/A10 end to end.  Returns (k_out, v_out, idx, pooled_scores)."""
    p = make_params(q, k, v, window, n_keep, kernel_size, pooling, full_rows, **modes)
    H, L, D = q.shape[1], q.shape[2], q.shape[3]
    k_out = torch.empty(1, H, n_keep + window, D, dtype=k.dtype)
    v_out = torch.empty_like(k_out)
    idx = torch.empty(H, n_keep, dtype=torch.int64)
    sc = torch.empty(H, L - window, dtype=q.dtype)
    _rc(lib().kvco_compress(ctypes.byref(p), _ptr(q), _ptr(k), _ptr(v), _ptr(k_out), _ptr(v_out), _ptr(idx),
                            _ptr(sc)), "compress")
    return k_out, v_out, idx, sc


def streaming(k, v, window, n_keep, n_q_heads):
    """A11 StreamingLLM: first n_keep + last W tokens."""
    _check_inner(k), _check_inner(v)
    p = Params()
    p.dtype = _DTYPE[k.dtype]
    p.n_q_heads, p.n_kv_heads = n_q_heads, k.shape[1]
    p.q_len, p.head_dim, p.window, p.k, p.pooling = k.shape[2], k.shape[3], window, n_keep, POOL_NONE
    p.k_stride_h, p.k_stride_l, p.v_stride_h, p.v_stride_l = k.stride(1), k.stride(2), v.stride(1), v.stride(2)
    k_out = torch.empty(1, n_q_heads, n_keep + window, k.shape[3], dtype=k.dtype)
    v_out = torch.empty_like(k_out)
    idx = torch.empty(n_q_heads, n_keep, dtype=torch.int64)
    _rc(lib().kvco_streaming(ctypes.byref(p), _ptr(k), _ptr(v), _ptr(k_out), _ptr(v_out), _ptr(idx)), "streaming")
    return k_out, v_out, idx


def ragged(q, k, v, window, max_capacity_prompt, kernel_size=7, pooling="maxpool", floor=0.2, normalize=True, head_capacity=None,
           n_threads=0):
    """AdaKV (head_capacity None: budgets from the scores, pyramidkv_utils.py:674-757) or HeadKV (head_capacity: per-head list,
    :813-878).  Returns (k_flat [sum_h (cap_h + W), D], v_flat, head_lens int32 [H], caps int32 [H], sorted idx int64 [H, n],
    pooled scores [H, n])."""
    sc = scores(q, k, window, kernel_size, pooling, full_rows=2, n_threads=n_threads)
    H, n = sc.shape
    p = make_params(q, k, v, window, 0, kernel_size, pooling, n_threads=n_threads)
    idx = torch.empty(H, n, dtype=torch.int64)
    val = torch.empty(H, n, dtype=sc.dtype)
    _rc(lib().kvco_sort_desc(ctypes.byref(p), _ptr(sc), _ptr(idx), _ptr(val)), "sort_desc")
    base = max_capacity_prompt - window
    caps = torch.empty(H, dtype=torch.int32)
    if head_capacity is None:
        _rc(lib().kvco_adakv_caps(ctypes.byref(p), _ptr(val), base, float(floor), int(bool(normalize)), _ptr(caps)), "adakv_caps")
    else:
        caps.copy_(torch.as_tensor(head_capacity, dtype=torch.int32))
    lens = caps + window
    rows = int(lens.sum())
    kf = torch.empty(rows, k.shape[3], dtype=k.dtype)
    vf = torch.empty_like(kf)
    _rc(lib().kvco_ragged_gather(ctypes.byref(p), _ptr(k), k.stride(1), k.stride(2), _ptr(idx), _ptr(caps), _ptr(kf)), "ragged_gather")
    _rc(lib().kvco_ragged_gather(ctypes.byref(p), _ptr(v), v.stride(1), v.stride(2), _ptr(idx), _ptr(caps), _ptr(vf)), "ragged_gather")
    return kf, vf, lens.to(torch.int32), caps, idx, sc


def merge_pivot(k, v, idx, window, n_threads=0):
    """merge_kv(key_states, value_states, indices, window_size, "pivot") (pyramidkv_utils.py:119-170) for idx [Hq, n_keep] int64
    (the method's topk indices).  Returns (k_out [1, Hq, n_keep + W, D] in the order [window, selected], v_out in the order
    [selected, window], pivot int32 [Hq, drop_len])."""
    _check_inner(k), _check_inner(v)
    assert idx.dim() == 2 and idx.is_contiguous() and idx.dtype == torch.int64
    p = Params()
    p.dtype = _DTYPE[k.dtype]
    p.n_q_heads, p.n_kv_heads = idx.shape[0], k.shape[1]
    p.q_len, p.head_dim = k.shape[2], k.shape[3]
    p.window, p.k, p.n_threads = window, idx.shape[1], n_threads
    p.kernel_size, p.pooling = 1, _POOL["maxpool"]
    p.k_stride_h, p.k_stride_l = k.stride(1), k.stride(2)
    p.v_stride_h, p.v_stride_l = v.stride(1), v.stride(2)
    H, R, D = idx.shape[0], idx.shape[1] + window, k.shape[3]
    ko = torch.empty(1, H, R, D, dtype=k.dtype)
    vo = torch.empty_like(ko)
    pivot = torch.empty(H, k.shape[2], dtype=torch.int32)
    dl = ctypes.c_int64(0)
    _rc(lib().kvco_merge_pivot(ctypes.byref(p), _ptr(k), _ptr(v), _ptr(idx), _ptr(ko), _ptr(vo), _ptr(pivot), ctypes.byref(dl)), "merge_pivot")
    M = int(dl.value)
    return ko, vo, pivot.view(-1)[: H * M].view(H, M).clone()


def think_prune(kc, q, recent_size, ratio):
    """key_pruner_query_driven(kc, q, recent_size, ratio) (pyramidkv_utils.py:13-26): kc [1, H, S, D] compressed keys, q [1, H, L, D].
    Returns (kv_pruned [1, H, S - recent, D - k], kv_recent [1, H, recent, D], keep mask bool [1, H, D] (= the reference's ~mask),
    channel scores [H, D])."""
    _check_inner(q)
    kc = kc.contiguous()
    H, S, D = kc.shape[1], kc.shape[2], kc.shape[3]
    n_prune = int(D * ratio)
    p = Params()
    p.dtype = _DTYPE[kc.dtype]
    p.n_q_heads = p.n_kv_heads = H
    p.q_len, p.head_dim = q.shape[2], D
    p.q_stride_h, p.q_stride_l = q.stride(1), q.stride(2)
    pruned = torch.empty(1, H, S - recent_size, D - n_prune, dtype=kc.dtype)
    keep = torch.empty(1, H, D, dtype=torch.uint8)
    sc = torch.empty(H, D, dtype=kc.dtype)
    _rc(lib().kvco_think_prune(ctypes.byref(p), _ptr(q), _ptr(kc), S, recent_size, n_prune, _ptr(pruned), _ptr(keep), _ptr(sc)), "think_prune")
    return pruned, kc[:, :, S - recent_size:, :], keep.bool(), sc


def l2norm(k, v, max_capacity_prompt, n_q_heads, n_threads=0):
    """L2NormCluster.update_kv (pyramidkv_utils.py:394-429) on k, v [1, Hq or Hkv, L, D]: the max_capacity_prompt smallest-norm
    keys per head, ascending.  Returns (k_out [1, Hq, cap, D], v_out, idx int64 [Hq, cap], norms [Hq, L] dtype)."""
    _check_inner(k), _check_inner(v)
    p = Params()
    p.dtype = _DTYPE[k.dtype]
    p.n_q_heads, p.n_kv_heads = n_q_heads, k.shape[1]
    p.q_len, p.head_dim = k.shape[2], k.shape[3]
    p.window, p.k, p.n_threads = 0, max_capacity_prompt, n_threads
    p.k_stride_h, p.k_stride_l = k.stride(1), k.stride(2)
    p.v_stride_h, p.v_stride_l = v.stride(1), v.stride(2)
    L, D, cap = k.shape[2], k.shape[3], max_capacity_prompt
    norms = torch.empty(n_q_heads, L, dtype=k.dtype)
    idx = torch.empty(n_q_heads, cap, dtype=torch.int64)
    ko = torch.empty(1, n_q_heads, cap, D, dtype=k.dtype)
    vo = torch.empty_like(ko)
    _rc(lib().kvco_l2norm(ctypes.byref(p), _ptr(k), _ptr(v), _ptr(norms), _ptr(idx), _ptr(ko), _ptr(vo)), "l2norm")
    return ko, vo, idx, norms


def pyramid_k(cap, window, q_len, layer_idx, n_layers, beta=20):
    """A9 schedule (pyramidkv_utils.py:205-215): k for this layer, or -1 for pass-through."""
    return int(lib().kvco_pyramid_k(cap, window, q_len, layer_idx, n_layers, beta))


# ---- CAM (pyramidkv_utils.py:431-513), restated with explicit fp32 operations and dtype roundings ----------------
def _rnd(x, dtype):
    return x.to(dtype).float()


def _cascade_sum(rows):
    """fp32 sum of a list of tensors in the order of torch's multi_row_sum (SumKernel.cpp; kvc_oracle.cpp / kvc_common.h CascadeSum)."""
    size, cl = len(rows), 0
    while (1 << cl) < size:
        cl += 1
    level_power = max(4, cl // 4)
    level_step = 1 << level_power
    full = size - size % level_step
    a0 = a1 = a2 = a3 = torch.zeros_like(rows[0])
    i = in_step = 0
    for v in rows:
        a0 = a0 + v
        i += 1
        in_step += 1
        if in_step == level_step and i <= full:
            in_step, mask = 0, level_step - 1
            a1, a0 = a1 + a0, torch.zeros_like(a0)
            if (i & (mask << level_power)) == 0:
                a2, a1 = a2 + a1, torch.zeros_like(a0)
                if (i & (mask << (2 * level_power))) == 0:
                    a3, a2 = a3 + a2, torch.zeros_like(a0)
    return ((a0 + a1) + a2) + a3


def cam_plan(q, k, window, start_budget, **modes):
    """(window-sum scores [H, n] = the reference's attn_cache (:476), merge_prob [H, L - W - start] = what it hands to
    torch.bernoulli at token s = start + column (:493-499)).  colmean = torch.mean(dim=-2) of the dtype probabilities: fp32 sum
    of the W rows in torch's cascade order, / W, one rounding; merge_prob = colmean[s] / max(max(colmean[:start]), max(colmean[s:s+W])) in
    dtype (fp32 quotient, one rounding), NaN -> 0, inf -> 1, clamp [0, 1]."""
    sc, lg, pr, ws = scores(q, k, window, 1, "maxpool", want_intermediates=True, **modes)
    dtype, (H, W, L) = q.dtype, pr.shape
    cm = _rnd(_cascade_sum([pr[:, w].float() for w in range(W)]) / float(W), dtype)
    m0 = cm[:, :start_budget].max(dim=-1).values
    nT = L - window - start_budget
    prob = torch.empty(H, nT)
    for t in range(nT):
        s = start_budget + t
        mean_attn = torch.maximum(m0, cm[:, s:s + window].max(dim=-1).values)
        prob[:, t] = _rnd(cm[:, s] / mean_attn, dtype)
    prob[torch.isnan(prob)] = 0.0
    prob[torch.isinf(prob)] = 1.0
    return ws, prob.clamp(0.0, 1.0).to(dtype)


def cam_merge(v, mask, window, start_budget):
    """value[s + 1 .. s + W] += value[s] * merge_mask / W for s = start .. L - W - 1, in order, every operation rounded to the dtype
    (:500-501).  v [1, H, L, D] (one copy per query head), mask [H, L - W - start] 0 / 1.  Returns the merged copy."""
    dtype = v.dtype
    out = v[0].float().clone()
    H, L, D = out.shape
    for t in range(L - window - start_budget):
        s = start_budget + t
        score1 = _rnd(_rnd(out[:, s] * mask[:, t, None].float(), dtype) / float(window), dtype)
        out[:, s + 1:s + window + 1] = _rnd(out[:, s + 1:s + window + 1] + score1[:, None], dtype)
    return out.to(dtype)[None]


def cam(q, k, v, window, n_keep, start_budget, mask, **modes):
    """CAMKVCluster.update_kv given the draws: (k_out, v_out, indices, scores, merge_prob, merged values)."""
    g = q.shape[1] // v.shape[1]
    vx = v.repeat_interleave(g, dim=1) if g > 1 else v
    sc, prob = cam_plan(q, k, window, start_budget, **modes)
    merged = cam_merge(vx, mask, window, start_budget)
    idx, _ = topk(sc.contiguous(), n_keep, modes.get("tie_mode", TIES_TORCH)) if False else topk(sc.contiguous(), n_keep, TIES_TORCH)
    return gather(k, idx, window, q.shape[1]), gather(merged, idx, window, q.shape[1]), idx, sc, prob, merged
