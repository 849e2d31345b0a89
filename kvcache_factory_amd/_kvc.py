"""ctypes binding of libkvc_hip.so (C-ABI: include/kvc.h).  PyTorch is used only for device memory
and the current HIP stream.  Fails loudly when the extension is absent — there is no fallback."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KVC_LIB_PATH", os.path.join(_HERE, "libkvc_hip.so"))   # override only for diagnostic builds

SNAPKV, PYRAMIDKV, H2O, STREAMINGLLM, ADAKV, HEADKV, L2NORM = 0, 1, 2, 3, 4, 5, 6
BF16, FP16, FP32 = 0, 1, 2
POOL_NONE, POOL_AVG, POOL_MAX = 0, 1, 2
TIES_TORCH_CPU, TIES_CANONICAL = 0, 1
OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_WORKSPACE, ERR_ALIGNMENT, ERR_HIP = 0, -1, -2, -3, -4, -5

DTYPES = {torch.bfloat16: BF16, torch.float16: FP16, torch.float32: FP32}
CAM = 100       # host-side method id (CAMKVCluster): no kvc_method of its own — kvc_cam_plan / kvc_cam_merge + select + gather
POOLINGS = {"avgpool": POOL_AVG, "maxpool": POOL_MAX, None: POOL_NONE, "none": POOL_NONE}
TIE_MODES = {"torch_cpu": TIES_TORCH_CPU, "canonical": TIES_CANONICAL}
DOT_EXACT, DOT_MFMA16 = 0, 1
DOT_MODES = {"exact": DOT_EXACT, "mfma16": DOT_MFMA16}
DOT_MODE = os.environ.get("KVC_DOT_MODE", "exact")      # default of every helper below

EXPORTS = ("kvc_version", "kvc_last_error", "kvc_workspace_bytes", "kvc_compress", "kvc_scores", "kvc_select",
           "kvc_gather", "kvc_pyramid_k", "kvc_workspace_layout", "kvc_workspace_bytes_batch", "kvc_compress_batch",
           "kvc_select_workspace_bytes", "kvc_decode_step", "kvc_ragged_workspace_bytes", "kvc_ragged_plan",
           "kvc_ragged_compact", "kvc_ragged_decode_step", "kvc_l2norm_workspace_bytes", "kvc_l2norm_compress",
           "kvc_merge_workspace_bytes", "kvc_merge_pivot", "kvc_think_prune", "kvc_decode_workspace_bytes", "kvc_decode_step2",
           "kvc_cam_workspace_bytes", "kvc_cam_plan", "kvc_cam_merge")


class KvcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"kvc error {code}: {msg}")
        self.code = code


class Params(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "method", "dtype", "bsz", "n_q_heads", "n_kv_heads", "q_len", "head_dim", "window", "k",
        "kernel_size", "pooling", "tie_mode", "debug_stage_mask", "dot_mode")] + [
        (n, ctypes.c_int64) for n in (
            "q_stride_b", "q_stride_h", "q_stride_l", "k_stride_b", "k_stride_h", "k_stride_l",
            "v_stride_b", "v_stride_h", "v_stride_l", "out_stride_h")]


class DecodeParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("dtype", "bsz", "n_q_heads", "n_kv_heads", "head_dim", "prefix_rows", "tail_rows",
                                               "new_rows")] + [("scaling", ctypes.c_float), ("reserved", ctypes.c_int32)] + [
        (n, ctypes.c_int64) for n in ("q_stride_b", "q_stride_h", "q_stride_l", "new_stride_b", "new_stride_h", "new_stride_l",
                                      "prefix_stride_b", "prefix_stride_h", "tail_stride_b", "tail_stride_h",
                                      "out_stride_b", "out_stride_h", "out_stride_l")]


class ThinkParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("dtype", "bsz", "n_heads", "q_len", "head_dim", "rows", "recent", "n_prune")] + \
               [(n, ctypes.c_int64) for n in ("q_stride_b", "q_stride_h", "q_stride_l", "kc_stride_h")]


class RaggedDecodeParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("dtype", "bsz", "n_q_heads", "n_kv_heads", "head_dim", "appended", "new_rows")] + [
        ("scaling", ctypes.c_float)] + [(n, ctypes.c_int64) for n in (
            "q_stride_b", "q_stride_h", "q_stride_l", "new_stride_b", "new_stride_h", "new_stride_l",
            "out_stride_b", "out_stride_h", "out_stride_l")] + [("slack_rows", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class DecodeExt(ctypes.Structure):
    _fields_ = [("v_prefix_stride_b", ctypes.c_int64), ("v_prefix_stride_h", ctypes.c_int64), ("pruned_rows", ctypes.c_int32),
                ("pruned_dim", ctypes.c_int32), ("pruned_stride_b", ctypes.c_int64), ("pruned_stride_h", ctypes.c_int64),
                ("chan_stride_b", ctypes.c_int64), ("chan_stride_h", ctypes.c_int64)]


_lib = None


def lib():
    """Load libkvc_hip.so.  Raises if it has not been built (python __graft_entry__.py / make -C csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C kvcache_factory_amd/csrc` "
                "(there is no CPU or PyTorch fallback for this path)")
        L = ctypes.CDLL(LIB_PATH)
        vp, pp, sz = ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_size_t
        L.kvc_version.restype = ctypes.c_int
        L.kvc_last_error.restype = ctypes.c_char_p
        L.kvc_workspace_bytes.argtypes = [pp]
        L.kvc_workspace_bytes.restype = sz
        L.kvc_compress.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        L.kvc_scores.argtypes = [pp, vp, vp, vp, vp, sz, vp]
        L.kvc_select.argtypes = [pp, vp, vp, vp, sz, vp]
        L.kvc_gather.argtypes = [pp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, vp, vp, vp]
        L.kvc_pyramid_k.argtypes = [ctypes.c_int64] * 6
        L.kvc_pyramid_k.restype = ctypes.c_int64
        L.kvc_workspace_layout.argtypes = [pp, ctypes.POINTER(sz * 3)]
        L.kvc_workspace_bytes_batch.argtypes = [pp, ctypes.c_int, vp]
        L.kvc_workspace_bytes_batch.restype = sz
        L.kvc_compress_batch.argtypes = [pp, ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        L.kvc_decode_step.argtypes = [ctypes.POINTER(DecodeParams)] + [vp] * 9
        L.kvc_cam_workspace_bytes.argtypes = [pp]
        L.kvc_cam_workspace_bytes.restype = sz
        L.kvc_cam_plan.argtypes = [pp, vp, vp, ctypes.c_int, vp, vp, vp, sz, vp]
        L.kvc_cam_merge.argtypes = [pp, vp, vp, ctypes.c_int, vp]
        L.kvc_decode_workspace_bytes.argtypes = [ctypes.POINTER(DecodeParams), ctypes.POINTER(DecodeExt)]
        L.kvc_decode_workspace_bytes.restype = sz
        L.kvc_decode_step2.argtypes = [ctypes.POINTER(DecodeParams), ctypes.POINTER(DecodeExt)] + [vp] * 11 + [sz, vp]
        L.kvc_ragged_workspace_bytes.argtypes = [pp]
        L.kvc_ragged_workspace_bytes.restype = sz
        L.kvc_ragged_plan.argtypes = [pp, vp, vp, ctypes.c_float, ctypes.c_int, vp, vp, vp, vp, sz, vp]
        L.kvc_ragged_compact.argtypes = [pp, vp, vp, vp, vp, ctypes.c_int, vp, vp, vp, ctypes.c_int64, vp, sz, vp]
        L.kvc_ragged_decode_step.argtypes = [ctypes.POINTER(RaggedDecodeParams)] + [vp] * 9
        L.kvc_select_workspace_bytes.argtypes = [pp]
        L.kvc_select_workspace_bytes.restype = sz
        L.kvc_l2norm_workspace_bytes.argtypes = [pp]
        L.kvc_l2norm_workspace_bytes.restype = sz
        L.kvc_l2norm_compress.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        L.kvc_merge_workspace_bytes.argtypes = [pp]
        L.kvc_merge_workspace_bytes.restype = sz
        L.kvc_merge_pivot.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        L.kvc_think_prune.argtypes = [ctypes.POINTER(ThinkParams), vp, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise KvcError(rc, lib().kvc_last_error().decode())


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _one_device(*ts):
    """Every tensor of a call lives on ONE GPU; returns it.  (HF device_map="auto" puts layers on different GPUs,
    run_longbench.py:390: each layer's call then runs on that layer's device — see _call.)"""
    devs = {t.device for t in ts if t is not None}
    if len(devs) != 1:
        raise RuntimeError(f"kvcache_factory_amd: the tensors of one call must share one device, got {sorted(map(str, devs))}")
    return next(iter(devs))


def _call(dev, fn, *args):
    """Run a library entry point with `dev` as the CURRENT HIP device.  The library enqueues on the stream it is handed —
    torch's default stream has handle 0 on every device, which HIP resolves against the current device — and raises
    kernel attributes (hipFuncSetAttribute) for the current device, so a layer living on cuda:1 while cuda:0 is current
    must switch first (the reference gets this for free from torch's own per-op device guards)."""
    if torch.cuda.current_device() == dev.index:
        return fn(*args)
    with torch.cuda.device(dev):
        return fn(*args)


_workspaces = {}


def workspace(device, nbytes):
    """Grow-only per-(device, stream) scratch; reuse is safe because every user is stream-ordered."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _last_dim_contig(t):
    return t if t.stride(-1) == 1 else t.contiguous()


def make_params(method, q, k, v, window, n_keep, kernel_size=5, pooling="avgpool", tie_mode="torch_cpu", dot_mode=None):
    ref = k if k is not None else q
    p = Params()
    p.method = method
    p.dtype = DTYPES[ref.dtype]
    p.bsz = ref.shape[0]
    p.n_q_heads = q.shape[1] if q is not None else ref.shape[1]
    p.n_kv_heads = k.shape[1] if k is not None else p.n_q_heads
    p.q_len, p.head_dim = ref.shape[2], ref.shape[3]
    p.window, p.k = window, n_keep
    p.kernel_size = kernel_size
    if pooling not in POOLINGS:
        raise ValueError('Pooling method not supported')      # pyramidkv_utils.py:333
    p.pooling = POOLINGS[pooling]
    p.tie_mode = TIE_MODES[tie_mode] if isinstance(tie_mode, str) else tie_mode
    dm = DOT_MODE if dot_mode is None else dot_mode
    p.dot_mode = DOT_MODES[dm] if isinstance(dm, str) else dm
    if q is not None:
        p.q_stride_b, p.q_stride_h, p.q_stride_l = q.stride(0), q.stride(1), q.stride(2)
    if k is not None:
        p.k_stride_b, p.k_stride_h, p.k_stride_l = k.stride(0), k.stride(1), k.stride(2)
    if v is not None:
        p.v_stride_b, p.v_stride_h, p.v_stride_l = v.stride(0), v.stride(1), v.stride(2)
    return p


def _require_gpu(*ts):
    for t in ts:
        if t is not None and t.device.type != "cuda":
            raise RuntimeError("kvcache_factory_amd runs on the GPU only (HIP kernels); got a %s tensor — "
                               "there is no CPU fallback" % t.device.type)


def _out_views(out, bsz, hq, rows, D, dtype, dev):
    """Caller-provided output buffers [bsz, H_q, capacity >= rows, D] (e.g. a decode cache with spare rows per head):
    returns (k_buf, v_buf, head stride in elements); None -> fresh dense tensors, stride 0 (= dense)."""
    if out is None:
        k_out = torch.empty(bsz, hq, rows, D, dtype=dtype, device=dev)
        return k_out, torch.empty_like(k_out), 0
    kb, vb = out
    for t in (kb, vb):
        assert t.dim() == 4 and t.shape[0] == bsz and t.shape[1] == hq and t.shape[2] >= rows and t.shape[3] == D and \
            t.dtype == dtype and t.device == dev and t.stride(3) == 1 and t.stride(2) == D and \
            t.stride(0) == hq * t.stride(1), "out buffers must be [bsz, H_q, capacity, D] with dense rows"
    assert kb.stride(1) == vb.stride(1)
    return kb, vb, kb.stride(1)


def compress(method, q, k, v, window, n_keep, kernel_size=5, pooling="avgpool", tie_mode="torch_cpu",
             n_q_heads=None, return_indices=False, return_scores=False, dot_mode=None, out=None):
    """One update_kv body on the GPU: returns (k_out, v_out[, idx][, scores]).
    out=(k_buf, v_buf): write K' / V' into the first k + window rows of each head of these [bsz, H_q, capacity, D]
    buffers (kvc_params.out_stride_h) and return views of them."""
    _require_gpu(q, k, v)
    k, v = _last_dim_contig(k), _last_dim_contig(v)
    if q is not None:
        q = _last_dim_contig(q)
    dev = _one_device(q, k, v, *(out if out is not None else ()))
    if method == STREAMINGLLM:
        hq = n_q_heads if n_q_heads is not None else (q.shape[1] if q is not None else k.shape[1])
        p = make_params(method, None, k, v, window, n_keep, kernel_size, None, tie_mode)
        p.n_q_heads = hq
        qq = None
    else:
        p = make_params(method, q, k, v, window, n_keep, kernel_size, pooling, tie_mode, dot_mode)
        hq, qq = q.shape[1], q
    bsz, L, D = k.shape[0], k.shape[2], k.shape[3]
    k_out, v_out, p.out_stride_h = _out_views(out, bsz, hq, n_keep + window, D, k.dtype, dev)
    scoring = method != STREAMINGLLM
    idx = torch.empty(bsz, hq, n_keep, dtype=torch.int64, device=dev) if (return_indices and scoring) else None
    sc = torch.empty(bsz, hq, L - window, dtype=k.dtype, device=dev) if (return_scores and scoring) else None
    nbytes = lib().kvc_workspace_bytes(ctypes.byref(p))
    if nbytes == 0 and scoring:
        raise KvcError(ERR_INVALID, lib().kvc_last_error().decode())
    ws = workspace(dev, nbytes) if nbytes else None
    _check(_call(dev, lib().kvc_compress, ctypes.byref(p), _ptr(qq), _ptr(k), _ptr(v), _ptr(k_out), _ptr(v_out), _ptr(idx),
                 _ptr(sc), _ptr(ws), nbytes, _stream(dev)))
    out = [k_out[:, :, :n_keep + window], v_out[:, :, :n_keep + window]]
    if return_indices:
        if not scoring:
            idx = torch.arange(n_keep, device=dev).expand(bsz, hq, n_keep).contiguous()
        out.append(idx)
    if return_scores:
        out.append(sc)
    return tuple(out)


def scores(method, q, k, window, kernel_size=5, pooling="avgpool", want_intermediates=False, dot_mode=None,
           softmax_path=None, debug_mask=0):
    """Stage A1-A5 only.  Returns pooled scores [bsz,Hq,L-W] (+ logits [bsz,Hq,L,W], rowmax, rowsum).
    softmax_path: None (library's choice) | "split" | "fused" | "fused_r2" (the fused form without the chain / worker wave split) —
    identical results (debug_stage_mask bits 3 / 4 / 10)."""
    _require_gpu(q, k)
    q, k = _last_dim_contig(q), _last_dim_contig(k)
    p = make_params(method, q, k, None, window, 0, kernel_size, pooling, dot_mode=dot_mode)
    p.debug_stage_mask = {None: 0, "split": 8, "fused": 16, "fused_r2": 16 | 1024 | 4096}[softmax_path] | debug_mask
    bsz, hq, L = q.shape[0], q.shape[1], q.shape[2]
    sc = torch.empty(bsz, hq, L - window, dtype=q.dtype, device=q.device)
    nbytes = lib().kvc_workspace_bytes(ctypes.byref(p))
    if nbytes == 0:
        raise KvcError(ERR_INVALID, lib().kvc_last_error().decode())
    dev = _one_device(q, k)
    ws = workspace(dev, nbytes)
    _check(_call(dev, lib().kvc_scores, ctypes.byref(p), _ptr(q), _ptr(k), _ptr(sc), _ptr(ws), nbytes, _stream(dev)))
    if not want_intermediates:
        return sc
    offs = (ctypes.c_size_t * 3)()
    _check(lib().kvc_workspace_layout(ctypes.byref(p), ctypes.byref(offs)))
    es = q.element_size()
    nlog = bsz * hq * L * window
    logits = ws[offs[0]:offs[0] + nlog * es].view(q.dtype).view(bsz, hq, L, window).clone()
    rowmax = ws[offs[1]:offs[1] + bsz * hq * window * 4].view(torch.float32).view(bsz, hq, window).clone()
    rowsum = ws[offs[2]:offs[2] + bsz * hq * window * 4].view(torch.float32).view(bsz, hq, window).clone()
    return sc, logits, rowmax, rowsum


def select(scores_t, n_keep, tie_mode="torch_cpu"):
    """Stage A7 only: scores [bsz,H,n] -> indices int64 [bsz,H,k]."""
    _require_gpu(scores_t)
    assert scores_t.dim() == 3 and scores_t.is_contiguous()
    p = Params()
    p.method, p.dtype = SNAPKV, DTYPES[scores_t.dtype]
    p.bsz, p.n_q_heads, p.n_kv_heads = scores_t.shape[0], scores_t.shape[1], scores_t.shape[1]
    p.window, p.head_dim, p.q_len, p.k = 1, 8, scores_t.shape[2] + 1, n_keep
    p.pooling = POOL_NONE
    p.tie_mode = TIE_MODES[tie_mode] if isinstance(tie_mode, str) else tie_mode
    idx = torch.empty(scores_t.shape[0], scores_t.shape[1], n_keep, dtype=torch.int64, device=scores_t.device)
    nbytes = lib().kvc_select_workspace_bytes(ctypes.byref(p))     # 0 unless the exact tie mode keeps lists / arrays there
    ws = workspace(scores_t.device, nbytes) if nbytes else None
    _check(_call(scores_t.device, lib().kvc_select, ctypes.byref(p), _ptr(scores_t), _ptr(idx), _ptr(ws), nbytes,
                 _stream(scores_t.device)))
    return idx


def gather(src, idx, window, n_q_heads):
    """Stage A8 only for one tensor: src [bsz,Hs,L,D], idx [bsz,Hq,k] or None -> [bsz,Hq,k+W,D]."""
    _require_gpu(src, idx)
    src = _last_dim_contig(src)
    n_keep = idx.shape[2] if idx is not None else 0
    p = make_params(SNAPKV, None, src, None, window, n_keep, 1, None)
    p.n_q_heads = n_q_heads
    out = torch.empty(src.shape[0], n_q_heads, n_keep + window, src.shape[3], dtype=src.dtype, device=src.device)
    dev = _one_device(src, idx)
    _check(_call(dev, lib().kvc_gather, ctypes.byref(p), _ptr(src), src.stride(0), src.stride(1), src.stride(2),
                 _ptr(idx.contiguous() if idx is not None else None), _ptr(out), _stream(dev)))
    return out


def decode_step(q, k_new, v_new, k_prefix, v_prefix, prefix_rows, k_tail, v_tail, tail_rows, scaling, k_pruned=None, chan=None,
                pruned_rows=0):
    """One decode step over the compacted cache (kvc_decode_step2): appends k_new / v_new [b, H_kv, t, D] to the tail buffers
    [b, H_kv, capacity, D] at row tail_rows and returns the attention output [b, t, H_q, D] of q [b, H_q, t, D] over the
    first prefix_rows rows per query head of k_prefix / v_prefix [b, H_q, >= rows, D] and the tail.
    ThinK (k_pruned [b, H_q, pruned_rows, Dk_padded], chan [b, H_q, Dk_padded] int16): the first pruned_rows rows' keys hold only
    the kept channels; their values are rows [0, pruned_rows) of v_prefix and k_prefix's rows pair with the v_prefix rows behind."""
    _require_gpu(q, k_new, v_new, k_prefix, v_prefix, k_tail, v_tail)
    dev = _one_device(q, k_new, v_new, k_prefix, v_prefix, k_tail, v_tail, k_pruned, chan)
    q, k_new, v_new = _last_dim_contig(q), _last_dim_contig(k_new), _last_dim_contig(v_new)
    if k_new.stride() != v_new.stride():        # K comes out of RoPE contiguous, V is the projection's transposed view
        k_new, v_new = k_new.contiguous(), v_new.contiguous()
    b, hq, t, D = q.shape
    assert k_tail.shape[2] >= tail_rows + t and k_new.shape == v_new.shape and k_new.shape[2] == t
    assert k_tail.stride() == v_tail.stride()
    for x in (k_prefix, v_prefix, k_tail, v_tail):
        assert x.stride(3) == 1 and x.stride(2) == D, "cache rows must be dense"
    out = torch.empty(b, t, hq, D, dtype=q.dtype, device=dev)
    p = DecodeParams()
    p.dtype, p.bsz, p.n_q_heads, p.n_kv_heads, p.head_dim = DTYPES[q.dtype], b, hq, k_new.shape[1], D
    p.prefix_rows, p.tail_rows, p.new_rows, p.scaling = prefix_rows, tail_rows, t, scaling
    p.q_stride_b, p.q_stride_h, p.q_stride_l = q.stride(0), q.stride(1), q.stride(2)
    p.new_stride_b, p.new_stride_h, p.new_stride_l = k_new.stride(0), k_new.stride(1), k_new.stride(2)
    p.prefix_stride_b, p.prefix_stride_h = k_prefix.stride(0), k_prefix.stride(1)
    p.tail_stride_b, p.tail_stride_h = k_tail.stride(0), k_tail.stride(1)
    p.out_stride_b, p.out_stride_h, p.out_stride_l = out.stride(0), out.stride(2), out.stride(1)
    x = DecodeExt()
    x.v_prefix_stride_b, x.v_prefix_stride_h = v_prefix.stride(0), v_prefix.stride(1)
    if pruned_rows:
        assert k_pruned.is_contiguous() and chan.is_contiguous() and chan.dtype == torch.int16 and chan.shape[-1] == k_pruned.shape[-1]
        x.pruned_rows, x.pruned_dim = pruned_rows, k_pruned.shape[3]
        x.pruned_stride_b, x.pruned_stride_h = k_pruned.stride(0), k_pruned.stride(1)
        x.chan_stride_b, x.chan_stride_h = chan.stride(0), chan.stride(1)
    nbytes = lib().kvc_decode_workspace_bytes(ctypes.byref(p), ctypes.byref(x))
    ws = workspace(dev, nbytes) if nbytes else None
    _check(_call(dev, lib().kvc_decode_step2, ctypes.byref(p), ctypes.byref(x), _ptr(q), _ptr(k_new), _ptr(v_new), _ptr(k_prefix),
                 _ptr(v_prefix), _ptr(k_tail), _ptr(v_tail), _ptr(k_pruned if pruned_rows else None), _ptr(chan if pruned_rows else None),
                 _ptr(out), _ptr(ws), nbytes, _stream(dev)))
    return out


def ragged_compress(method, q, k, v, window, base_capacity, kernel_size=7, pooling="maxpool", floor=0.2, normalize=True,
                    head_capacity=None, slack=0, return_indices=False, return_scores=False, tie_mode="torch_cpu"):
    """AdaKV / HeadKV on the GPU (kvc_ragged_plan + kvc_ragged_compact): per-head budgets and the flattened cache.
    Returns a dict: k_flat / v_flat [rows, D] (head h at rows seg_off[h] .. seg_off[h] + lens[h], `slack` spare rows behind
    every head), lens (host list, cap_h + W), caps / seg_off / seg_len (device), flag (AdaKV: 1 if a tie at the global
    threshold spanned several heads), idx [b*H, max cap] and scores [b, H, n] on request.
    One host synchronisation, where the reference has its own (the capacities become Python integers, :721-741)."""
    _require_gpu(q, k, v)
    q, k, v = _last_dim_contig(q), _last_dim_contig(k), _last_dim_contig(v)
    dev = _one_device(q, k, v)
    p = make_params(method, q, k, v, window, base_capacity, kernel_size, pooling, tie_mode)
    b, hq, L, D = q.shape
    heads, n = b * hq, L - window
    nbytes = lib().kvc_ragged_workspace_bytes(ctypes.byref(p))
    if nbytes == 0:
        raise KvcError(ERR_INVALID, lib().kvc_last_error().decode())
    ws = workspace(dev, nbytes)
    sc = torch.empty(b, hq, n, dtype=q.dtype, device=dev) if return_scores else None
    flags = torch.zeros(b, 2, dtype=torch.int32, device=dev)
    if method == ADAKV:
        caps = torch.empty(heads, dtype=torch.int32, device=dev)
    else:
        caps = torch.as_tensor(head_capacity, dtype=torch.int32).reshape(-1).to(dev)
        assert caps.numel() == heads, "head_capacity needs one entry per (batch, head)"
        caps = caps.clamp(max=n).contiguous()
    _check(_call(dev, lib().kvc_ragged_plan, ctypes.byref(p), _ptr(q), _ptr(k), float(floor), int(bool(normalize)),
                 _ptr(caps if method == ADAKV else None), _ptr(flags), _ptr(sc), _ptr(ws), nbytes, _stream(dev)))
    caps_host = caps.cpu()                                     # the synchronisation
    lens = (caps_host + window).tolist()
    offs, row = [], 0
    for ln in lens:
        offs.append(row)
        row += ln + slack
    seg_off = torch.tensor(offs, dtype=torch.int64, device=dev)
    k_flat = torch.empty(row, D, dtype=k.dtype, device=dev)
    v_flat = torch.empty_like(k_flat)
    cap_max = int(caps_host.max()) if heads else 0
    idx = torch.full((heads, max(cap_max, 1)), -1, dtype=torch.int64, device=dev) if return_indices else None
    _check(_call(dev, lib().kvc_ragged_compact, ctypes.byref(p), _ptr(k), _ptr(v), _ptr(caps), _ptr(seg_off), cap_max + window,
                 _ptr(k_flat), _ptr(v_flat), _ptr(idx), idx.shape[1] if idx is not None else 0, _ptr(ws), nbytes, _stream(dev)))
    return dict(k_flat=k_flat, v_flat=v_flat, lens=lens, caps=caps, seg_off=seg_off,
                seg_len=(caps + window).to(torch.int32), flag=flags[:, 0], idx=idx, scores=sc, slack=slack)


def ragged_decode_step(q, k_new, v_new, k_flat, v_flat, seg_off, seg_len, appended, scaling, slack=0):
    """One decode step over the ragged cache (kvc_ragged_decode_step): k_new / v_new [b, H_kv, t, D] are written behind every
    query head's segment (row seg_off + seg_len + appended) and the attention output [b, t, H_q, D] comes back."""
    _require_gpu(q, k_new, v_new, k_flat, v_flat)
    dev = _one_device(q, k_new, v_new, k_flat, v_flat, seg_off, seg_len)
    q, k_new, v_new = _last_dim_contig(q), _last_dim_contig(k_new), _last_dim_contig(v_new)
    if k_new.stride() != v_new.stride():
        k_new, v_new = k_new.contiguous(), v_new.contiguous()
    b, hq, t, D = q.shape
    out = torch.empty(b, t, hq, D, dtype=q.dtype, device=dev)
    p = RaggedDecodeParams()
    p.dtype, p.bsz, p.n_q_heads, p.n_kv_heads, p.head_dim = DTYPES[q.dtype], b, hq, k_new.shape[1], D
    p.appended, p.new_rows, p.scaling, p.slack_rows = appended, t, scaling, slack
    p.q_stride_b, p.q_stride_h, p.q_stride_l = q.stride(0), q.stride(1), q.stride(2)
    p.new_stride_b, p.new_stride_h, p.new_stride_l = k_new.stride(0), k_new.stride(1), k_new.stride(2)
    p.out_stride_b, p.out_stride_h, p.out_stride_l = out.stride(0), out.stride(2), out.stride(1)
    _check(_call(dev, lib().kvc_ragged_decode_step, ctypes.byref(p), _ptr(q), _ptr(k_new), _ptr(v_new), _ptr(k_flat), _ptr(v_flat),
                 _ptr(seg_off), _ptr(seg_len), _ptr(out), _stream(dev)))
    return out


def compress_merge(method, q, k, v, window, n_keep, kernel_size=5, pooling="avgpool", tie_mode="torch_cpu", n_q_heads=None):
    """update_kv with merge="pivot": the method's scores and top-k (StreamingLLM: the first n_keep positions, :610-611), then
    merge_kv instead of the gather (pyramidkv_utils.py:337-339).  Returns (key_states', value_states')."""
    if method == STREAMINGLLM:
        hq = n_q_heads if n_q_heads is not None else k.shape[1]
        idx = torch.arange(n_keep, device=k.device, dtype=torch.int64).expand(k.shape[0], hq, n_keep).contiguous()
    else:
        sc = scores(method, q, k, window, kernel_size, pooling)
        idx = select(sc, n_keep, tie_mode)
    return merge_pivot(method, k, v, idx, window)


def merge_pivot(method, k, v, idx, window, return_pivot=False, out=None):
    """merge_kv(key_states, value_states, indices, window_size, "pivot") (pyramidkv_utils.py:119-170) on the GPU (kvc_merge_pivot)
    for idx [b, H_q, n_keep] int64, the indices the method's top-k selected.  Returns (k_out [b, H_q, n_keep + W, D] in the
    reference's key order [window, selected], v_out in its value order [selected, window][, pivot int32 [b, H_q, drop_len]])."""
    _require_gpu(k, v, idx)
    k, v = _last_dim_contig(k), _last_dim_contig(v)
    dev = _one_device(k, v, idx, *(out if out is not None else ()))
    bsz, hq, n_keep = idx.shape
    idx = idx.contiguous()
    p = make_params(method, None, k, v, window, n_keep, 1, None)
    p.n_q_heads = hq
    L, D = k.shape[2], k.shape[3]
    k_out, v_out, p.out_stride_h = _out_views(out, bsz, hq, n_keep + window, D, k.dtype, dev)
    nbytes = lib().kvc_merge_workspace_bytes(ctypes.byref(p))
    if nbytes == 0:
        raise KvcError(ERR_INVALID, lib().kvc_last_error().decode())
    ws = workspace(dev, nbytes)
    pivot = torch.empty(bsz * hq, L, dtype=torch.int32, device=dev) if return_pivot else None
    dl = torch.empty(bsz, dtype=torch.int32, device=dev) if return_pivot else None
    _check(_call(dev, lib().kvc_merge_pivot, ctypes.byref(p), _ptr(k), _ptr(v), _ptr(idx), _ptr(k_out), _ptr(v_out), _ptr(pivot), _ptr(dl),
                 _ptr(ws), nbytes, _stream(dev)))
    res = [k_out[:, :, :n_keep + window], v_out[:, :, :n_keep + window]]
    if return_pivot:
        m = int(dl.max())
        res.append(pivot.view(bsz, hq, L)[:, :, :m])
    return tuple(res)


def think_prune(kc, q, recent_size, ratio, return_scores=False):
    """key_pruner_query_driven(kc, q, recent_size, ratio) (pyramidkv_utils.py:13-26) on the GPU (kvc_think_prune): kc [b, H, S, D]
    the compressed keys (rows dense; a view with a head stride is fine), q [b, H, L, D].  Returns (kv_pruned
    [b, H, S - recent, D - int(D * ratio)], kv_recent = kc[:, :, S - recent:], keep mask bool [b, H, D] (the reference's ~mask)
    [, channel scores [b, H, D]])."""
    _require_gpu(kc, q)
    q = _last_dim_contig(q)
    dev = _one_device(kc, q)
    b, H, S, D = kc.shape
    if kc.stride(3) != 1 or kc.stride(2) != D or kc.stride(0) != H * kc.stride(1):
        kc = kc.contiguous()
    n_prune = int(D * ratio)
    p = ThinkParams()
    p.dtype, p.bsz, p.n_heads, p.q_len, p.head_dim = DTYPES[kc.dtype], b, H, q.shape[2], D
    p.rows, p.recent, p.n_prune = S, recent_size, n_prune
    p.q_stride_b, p.q_stride_h, p.q_stride_l = q.stride(0), q.stride(1), q.stride(2)
    p.kc_stride_h = kc.stride(1)
    pruned = torch.empty(b, H, S - recent_size, D - n_prune, dtype=kc.dtype, device=dev)
    keep = torch.empty(b, H, D, dtype=torch.uint8, device=dev)
    sc = torch.empty(b, H, D, dtype=kc.dtype, device=dev) if return_scores else None
    _check(_call(dev, lib().kvc_think_prune, ctypes.byref(p), _ptr(q), _ptr(kc), _ptr(pruned), _ptr(keep), _ptr(sc), _stream(dev)))
    res = [pruned, kc[:, :, S - recent_size:, :], keep.bool()]
    if return_scores:
        res.append(sc)
    return tuple(res)


def l2norm_compress(k, v, rows, n_q_heads=None, return_indices=False, return_norms=False, out=None):
    """L2NormCluster.update_kv's kept branch (pyramidkv_utils.py:418-429) on the GPU (kvc_l2norm_compress): the `rows` keys of
    every head with the smallest L2 norm, ascending, and their values.  k, v: [b, H_kv (or H_q), L, D]; returns
    (k_out [b, H_q, rows, D], v_out[, idx int64 [b, H_q, rows]][, norms [b, H_q, L]])."""
    _require_gpu(k, v)
    k, v = _last_dim_contig(k), _last_dim_contig(v)
    dev = _one_device(k, v, *(out if out is not None else ()))
    hq = n_q_heads if n_q_heads is not None else k.shape[1]
    p = make_params(L2NORM, None, k, v, 0, rows, 1, None)
    p.n_q_heads = hq
    bsz, L, D = k.shape[0], k.shape[2], k.shape[3]
    k_out, v_out, p.out_stride_h = _out_views(out, bsz, hq, rows, D, k.dtype, dev)
    idx = torch.empty(bsz, hq, rows, dtype=torch.int64, device=dev) if return_indices else None
    norms = torch.empty(bsz, hq, L, dtype=k.dtype, device=dev) if return_norms else None
    nbytes = lib().kvc_l2norm_workspace_bytes(ctypes.byref(p))
    if nbytes == 0:
        raise KvcError(ERR_INVALID, lib().kvc_last_error().decode())
    ws = workspace(dev, nbytes)
    _check(_call(dev, lib().kvc_l2norm_compress, ctypes.byref(p), _ptr(k), _ptr(v), _ptr(k_out), _ptr(v_out), _ptr(idx), _ptr(norms),
                 _ptr(ws), nbytes, _stream(dev)))
    res = [k_out[:, :, :rows], v_out[:, :, :rows]]
    if return_indices:
        res.append(idx)
    if return_norms:
        res.append(norms)
    return tuple(res)


def cam_plan(q, k, window, start_budget):
    """CAM (pyramidkv_utils.py:461-499) up to its draw: (window-sum scores [b, H_q, n], merge_prob [b, H_q, n - start_budget])."""
    _require_gpu(q, k)
    q, k = _last_dim_contig(q), _last_dim_contig(k)
    dev = _one_device(q, k)
    p = make_params(SNAPKV, q, k, None, window, 0, 1, None)
    b, hq, L = q.shape[0], q.shape[1], q.shape[2]
    sc = torch.empty(b, hq, L - window, dtype=q.dtype, device=dev)
    prob = torch.empty(b, hq, L - window - start_budget, dtype=q.dtype, device=dev)
    nbytes = lib().kvc_cam_workspace_bytes(ctypes.byref(p))
    if nbytes == 0:
        raise KvcError(ERR_INVALID, lib().kvc_last_error().decode())
    ws = workspace(dev, nbytes)
    _check(_call(dev, lib().kvc_cam_plan, ctypes.byref(p), _ptr(q), _ptr(k), int(start_budget), _ptr(sc), _ptr(prob), _ptr(ws), nbytes, _stream(dev)))
    return sc, prob


def cam_merge(v, merge_mask, window, start_budget):
    """The CAM value recurrence (:500-501) in place on v [b, H_q, L, D] given merge_mask [b, H_q, L - window - start_budget] (0 / 1)."""
    _require_gpu(v, merge_mask)
    assert v.stride(3) == 1 and merge_mask.dtype == torch.uint8 and merge_mask.is_contiguous()
    dev = _one_device(v, merge_mask)
    p = make_params(SNAPKV, None, v, v, window, 0, 1, None)
    _check(_call(dev, lib().kvc_cam_merge, ctypes.byref(p), _ptr(v), _ptr(merge_mask), int(start_budget), _stream(dev)))
    return v


def pyramid_k(cap, window, q_len, layer_idx, n_layers, beta=20):
    """PyramidKV per-layer budget (pyramidkv_utils.py:205-215): k, or -1 for pass-through."""
    return int(lib().kvc_pyramid_k(cap, window, q_len, layer_idx, n_layers, beta))


class CompressPlan:
    """Everything one update_kv needs, resolved once: params, pointers, outputs and workspace.  `run()` is a single
    C call that only enqueues kernels on the current stream (used by bench.py and by callers that recompress the
    same shapes, e.g. one plan per layer)."""

    def __init__(self, method, q, k, v, window, n_keep, kernel_size=5, pooling="avgpool", tie_mode="torch_cpu",
                 n_q_heads=None, want_indices=False, want_scores=False, dot_mode=None):
        _require_gpu(q, k, v)
        self.q = _last_dim_contig(q) if q is not None else None
        self.k, self.v = _last_dim_contig(k), _last_dim_contig(v)
        dev = _one_device(self.q, self.k, self.v)
        scoring = method != STREAMINGLLM
        if scoring:
            self.p = make_params(method, self.q, self.k, self.v, window, n_keep, kernel_size, pooling, tie_mode, dot_mode)
            hq = self.q.shape[1]
        else:
            hq = n_q_heads if n_q_heads is not None else self.k.shape[1]
            self.p = make_params(method, None, self.k, self.v, window, n_keep, kernel_size, None, tie_mode)
            self.p.n_q_heads = hq
        bsz, L, D = self.k.shape[0], self.k.shape[2], self.k.shape[3]
        self.k_out = torch.empty(bsz, hq, n_keep + window, D, dtype=self.k.dtype, device=dev)
        self.v_out = torch.empty_like(self.k_out)
        self.idx = torch.empty(bsz, hq, n_keep, dtype=torch.int64, device=dev) if (want_indices and scoring) else None
        self.scores = torch.empty(bsz, hq, L - window, dtype=self.k.dtype, device=dev) if (want_scores and scoring) else None
        self.nbytes = lib().kvc_workspace_bytes(ctypes.byref(self.p))
        if self.nbytes == 0 and scoring:
            raise KvcError(ERR_INVALID, lib().kvc_last_error().decode())
        self.ws = workspace(dev, self.nbytes) if self.nbytes else None
        self.dev = dev
        self._args = (ctypes.byref(self.p), _ptr(self.q if scoring else None), _ptr(self.k), _ptr(self.v),
                      _ptr(self.k_out), _ptr(self.v_out), _ptr(self.idx), _ptr(self.scores), _ptr(self.ws), self.nbytes)
        self._fn = lib().kvc_compress

    def run(self, stream=None):
        rc = _call(self.dev, self._fn, *self._args, stream if stream is not None else _stream(self.dev))
        if rc:
            _check(rc)
        return self.k_out, self.v_out


class BatchPlan:
    """kvc_compress_batch over a list of (q, k, v) of identical shape — e.g. the layers of one prompt — with per-item
    budgets.  Everything is resolved once; run() is one C call that enqueues each kernel once for all items.

    q_rows="window": each q holds ONLY the scoring rows, [bsz, H_q, W, D] (the caller kept the last W query rows of a
    layer instead of the whole projection).  The library addresses row l at q + l*q_stride_l and reads rows L-W..L-1
    only (include/kvc.h), so the base handed over is window_base - (L-W)*q_stride_l.  Not for H2O (all rows score).
    q may be None for StreamingLLM (n_q_heads then names the output heads)."""

    def __init__(self, method, qkv, window, n_keep, kernel_size=5, pooling="avgpool", tie_mode="torch_cpu",
                 want_indices=False, dot_mode=None, q_rows="all", n_q_heads=None, outs=None):
        n = len(qkv)
        q0, k0, v0 = qkv[0]
        _require_gpu(q0, k0, v0)
        scoring = method != STREAMINGLLM
        cq = (lambda t: _last_dim_contig(t)) if scoring else (lambda t: None)
        self.items = [(cq(q), _last_dim_contig(k), _last_dim_contig(v)) for q, k, v in qkv]
        i0 = self.items[0]
        for q, k, v in self.items:
            assert k.shape == i0[1].shape and k.stride() == i0[1].stride() and v.stride() == i0[2].stride() and \
                k.dtype == i0[1].dtype and (not scoring or (q.shape == i0[0].shape and q.stride() == i0[0].stride())), \
                "items must share one layout"
        keeps = list(n_keep) if hasattr(n_keep, "__len__") else [int(n_keep)] * n
        q, k, v = i0
        self.p = make_params(method, q, k, v, window, max(keeps), kernel_size, pooling if scoring else None, tie_mode,
                             dot_mode)
        if not scoring:
            self.p.n_q_heads = n_q_heads if n_q_heads is not None else k.shape[1]
        q_off = 0
        if q_rows == "window":
            assert scoring and method != H2O and q.shape[2] == window, "q_rows='window' needs [bsz,H,W,D] queries"
            q_off = (k.shape[2] - window) * q.stride(2) * q.element_size()
        else:
            assert q_rows == "all" and (not scoring or q.shape[2] == k.shape[2])
        dev = _one_device(*[t for it in self.items for t in it], *[t for o in (outs or ()) for t in o])
        bsz, hq, D = k.shape[0], self.p.n_q_heads, k.shape[3]
        self._strides = None
        if outs is None:
            self.k_out = [torch.empty(bsz, hq, kk + window, D, dtype=k.dtype, device=dev) for kk in keeps]
            self.v_out = [torch.empty_like(t) for t in self.k_out]
            kbufs, vbufs = self.k_out, self.v_out
        else:                              # caller's buffers, one (k_buf, v_buf) per item, each with its own head stride
            trip = [_out_views(o, bsz, hq, kk + window, D, k.dtype, dev) for o, kk in zip(outs, keeps)]
            self._strides = (ctypes.c_int64 * n)(*[t[2] for t in trip])
            kbufs, vbufs = [t[0] for t in trip], [t[1] for t in trip]
            self.k_out = [t[0][:, :, :kk + window] for t, kk in zip(trip, keeps)]
            self.v_out = [t[1][:, :, :kk + window] for t, kk in zip(trip, keeps)]
        self.idx = [torch.empty(bsz, hq, kk, dtype=torch.int64, device=dev) for kk in keeps] if (want_indices and scoring) else None
        arr = ctypes.c_void_p * n
        self._keep = (ctypes.c_int32 * n)(*keeps)
        self._q = arr(*[t[0].data_ptr() - q_off for t in self.items]) if scoring else None
        self._k = arr(*[t[1].data_ptr() for t in self.items])
        self._v = arr(*[t[2].data_ptr() for t in self.items])
        self._ko = arr(*[t.data_ptr() for t in kbufs])
        self._vo = arr(*[t.data_ptr() for t in vbufs])
        self._ix = arr(*[t.data_ptr() for t in self.idx]) if self.idx is not None else None
        self.nbytes = lib().kvc_workspace_bytes_batch(ctypes.byref(self.p), n, self._keep)
        if self.nbytes == 0 and scoring:
            raise KvcError(ERR_INVALID, lib().kvc_last_error().decode())
        self.ws = workspace(dev, self.nbytes) if self.nbytes else None
        self.dev, self.n = dev, n

    def call(self, params=None, scores_out=None, stream=None):
        """The raw kvc_compress_batch call of this plan: `params` overrides the parameter block (e.g. debug_stage_mask for
        profiling one kernel), `scores_out` is a ctypes array of per-item score buffers; returns the status code."""
        return _call(self.dev, lib().kvc_compress_batch, ctypes.byref(params if params is not None else self.p), self.n,
                     self._keep, self._q, self._k, self._v, self._ko, self._vo, self._ix, scores_out, self._strides,
                     _ptr(self.ws), self.nbytes, stream if stream is not None else _stream(self.dev))

    def run(self, stream=None):
        rc = self.call(stream=stream)
        if rc:
            _check(rc)
        return self.k_out, self.v_out


class GroupPlan:
    """kvc_compress_batch for a group of layers that RECURS (the same layout and budgets for every prompt): parameter block,
    ctypes tables and workspace size are built once; `run` writes this call's pointers into the tables and makes the one C call.
    What a prompt pays per group on the host is then ~40 pointer reads instead of building a BatchPlan (the reference's call
    site runs once per layer, llama_model.py:283-286; PrefillBatch.flush keeps these plans per (layout, budgets))."""

    def __init__(self, method, q0, k0, v0, window, keeps, kernel_size, pooling, tie_mode, q_rows="window", n_q_heads=None,
                 out_strides=None, dot_mode=None):
        scoring = method != STREAMINGLLM
        self.n = n = len(keeps)
        self.p = make_params(method, q0 if scoring else None, k0, v0, window, max(keeps), kernel_size, pooling if scoring else None,
                             tie_mode, dot_mode)
        if not scoring:
            self.p.n_q_heads = n_q_heads if n_q_heads is not None else k0.shape[1]
        self.q_off = 0
        if scoring and q_rows == "window":
            assert method != H2O and q0.shape[2] == window, "q_rows='window' needs [bsz,H,W,D] queries"
            self.q_off = (k0.shape[2] - window) * q0.stride(2) * q0.element_size()
        arr = ctypes.c_void_p * n
        self._keep = (ctypes.c_int32 * n)(*keeps)
        self._q = arr() if scoring else None
        self._k, self._v, self._ko, self._vo = arr(), arr(), arr(), arr()
        self._strides = (ctypes.c_int64 * n)(*out_strides) if out_strides is not None else None
        self.nbytes = lib().kvc_workspace_bytes_batch(ctypes.byref(self.p), n, self._keep)
        if self.nbytes == 0 and scoring:
            raise KvcError(ERR_INVALID, lib().kvc_last_error().decode())
        self._pp = ctypes.byref(self.p)
        self._fn = lib().kvc_compress_batch

    def run(self, dev, qs, ks, vs, kbufs, vbufs):
        """Pointers only (tensors of the plan's layout, checked by the caller's layout key); enqueues on the current stream."""
        if self._q is not None:
            off = self.q_off
            self._q[:] = [t.data_ptr() - off for t in qs]
        self._k[:] = [t.data_ptr() for t in ks]
        self._v[:] = [t.data_ptr() for t in vs]
        self._ko[:] = [t.data_ptr() for t in kbufs]
        self._vo[:] = [t.data_ptr() for t in vbufs]
        ws = workspace(dev, self.nbytes) if self.nbytes else None
        rc = _call(dev, self._fn, self._pp, self.n, self._keep, self._q, self._k, self._v, self._ko, self._vo, None, None,
                   self._strides, _ptr(ws), self.nbytes, _stream(dev))
        if rc:
            _check(rc)


_GROUP_PLANS = {}                                        # layout key -> GroupPlan (compress_batch(reuse=True))


def compress_batch(method, qs, ks, vs, window, keeps, kernel_size=5, pooling="avgpool", tie_mode="torch_cpu",
                   q_rows="all", n_q_heads=None, return_indices=False, dot_mode=None, outs=None, reuse=False):
    """n independent update_kv bodies of one layout in ONE library call (kvc_compress_batch): lists of k_out, v_out
    (and indices).  See BatchPlan for q_rows.
    reuse=True (PrefillBatch.flush: the same group of layers comes back with every prompt): the plan of this layout and
    these budgets is kept (GroupPlan) and only the pointers are refreshed; the caller vouches that the items share the
    first item's shape / strides / dtype / device (it grouped them by exactly that)."""
    if reuse and not return_indices:
        scoring = method != STREAMINGLLM
        q0, k0, v0 = (qs[0] if scoring else None), ks[0], vs[0]
        _require_gpu(q0, k0, v0)
        dev = k0.device
        keeps = tuple(keeps)
        hq = q0.shape[1] if scoring else (n_q_heads if n_q_heads is not None else k0.shape[1])
        if outs is None:
            trip = [_out_views(None, k0.shape[0], hq, kk + window, k0.shape[3], k0.dtype, dev) for kk in keeps]
            strides = None
        else:
            trip = [(o[0], o[1], o[0].stride(1)) for o in outs]
            strides = tuple(t[2] for t in trip)
        key = (method, tuple(k0.shape), k0.stride(), v0.stride(), q0.stride() if scoring else None, k0.dtype, dev, hq, window, keeps,
               kernel_size, pooling, tie_mode, q_rows, dot_mode, strides)
        plan = _GROUP_PLANS.get(key)
        if plan is None:
            if len(_GROUP_PLANS) >= 64:
                _GROUP_PLANS.clear()
            plan = _GROUP_PLANS[key] = GroupPlan(method, q0, k0, v0, window, list(keeps), kernel_size, pooling, tie_mode, q_rows, hq,
                                                 strides, dot_mode)
        plan.run(dev, qs if scoring else None, ks, vs, [t[0] for t in trip], [t[1] for t in trip])
        return ([t[0][:, :, :kk + window] for t, kk in zip(trip, keeps)], [t[1][:, :, :kk + window] for t, kk in zip(trip, keeps)])
    qs = qs if qs is not None else [None] * len(ks)
    bp = BatchPlan(method, list(zip(qs, ks, vs)), window, keeps, kernel_size, pooling, tie_mode, return_indices,
                   dot_mode, q_rows, n_q_heads, outs)
    bp.run()
    return (bp.k_out, bp.v_out, bp.idx) if return_indices else (bp.k_out, bp.v_out)
