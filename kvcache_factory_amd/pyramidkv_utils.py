"""KV-cluster layer — drop-in for the reference's pyramidkv/pyramidkv_utils.py, compute on MI355X.

Same class names, constructor arguments, `update_kv` signature / return values and error behaviour as
the reference (file:line below); the ~60 lines of torch ops in each reference `update_kv` are replaced
by one call into the HIP library (include/kvc.h: kvc_compress).  Differences, all deliberate:

  * `key_states` / `value_states` may carry either H_q heads (the reference's caller repeat_kv()-expands
    them first, llama_model.py:277-278) or the model's H_kv heads with `num_key_value_groups` =
    H_q / H_kv — the outputs are identical ([bsz, H_q, cap, D]); the second form reads each KV head once.
  * the reference prints its budget on every call (pyramidkv_utils.py:217,312,539,601); here that is
    opt-in: set `kvcache_factory_amd.pyramidkv_utils.VERBOSE = True` (or KVC_VERBOSE=1).
  * `merge="pivot"` (LOOK-M pivot merge, merge_kv :119-170), L2NormCluster (:394-429) and SnapKVCluster.update_think (ThinK
    channel pruning, :13-26, :349-392; model forward in attention.make_think_forward) are built; CAM is not.
  * tensors must live on the GPU: there is no CPU / eager fallback.
  * `PrefillBatch` (not in the reference): the patched forwards hand every layer's (K, window-Q, V) to it and every
    GROUP_LAYERS-th layer (and the last) flushes the parked ones through ONE kvc_compress_batch call on a side stream,
    beside the model's prefill of the following layers.  Legal because the prefill attention of each layer runs on the
    uncompressed K/V (llama_model.py:306-313) and nothing reads the compressed cache before the first decode step;
    the bytes are those of per-layer update_kv calls.  `BATCH_LAYERS = False` (KVC_BATCH_LAYERS=0) restores one
    update_kv per layer.
"""
import contextlib
import math  # noqa: F401  (kept for parity with the reference's namespace)
import os

import torch

from . import _kvc

VERBOSE = os.environ.get("KVC_VERBOSE", "0") == "1"
TIE_MODE = os.environ.get("KVC_TIE_MODE", "torch_cpu")   # "torch_cpu" (reference-exact ties) | "canonical"
BATCH_LAYERS = os.environ.get("KVC_BATCH_LAYERS", "1") == "1"   # patched forwards compress the layers in batched calls
GROUP_LAYERS = int(os.environ.get("KVC_GROUP_LAYERS", "8"))      # ... of this many layers each (0: the whole prompt at once)
DECODE_KERNEL = os.environ.get("KVC_DECODE_KERNEL", "1") == "1"  # decode steps by kvc_decode_step over the split cache (N1)
OVERLAP = os.environ.get("KVC_OVERLAP", "1") == "1"              # grouped flushes run on a side stream beside the model's prefill
SPARE_ROWS = 256                                         # decode rows per head reserved behind the compressed prompt
RAGGED_SLACK = 256                                       # AdaKV / HeadKV: spare rows behind every head of the flattened cache


def _say(msg):
    if VERBOSE:
        print(msg)


def _check_merge(merge):
    if merge is not None and merge not in ("pivot",):
        raise ValueError('Merge method not supported')            # pyramidkv_utils.py:164


def _run(method, key_states, query_states, value_states, window, n_keep, kernel_size, pooling):
    if pooling not in ("avgpool", "maxpool") and method != _kvc.H2O:
        raise ValueError('Pooling method not supported')            # pyramidkv_utils.py:333
    k_out, v_out = _kvc.compress(method, query_states, key_states, value_states, window, n_keep,
                                 kernel_size, pooling if method != _kvc.H2O else None, TIE_MODE)
    return k_out, v_out


class _KVCluster:
    """State every reference cluster carries (ctor/reset of pyramidkv_utils.py:174-195, :286-304, :516-531,
    :579-593): the knobs only — clusters are stateless between calls."""

    _name = "KVCluster"

    def __init__(self, window_size=64, max_capacity_prompt=256 + 64, kernel_size=5, pooling='avgpool', merge=None):
        self.reset(window_size, max_capacity_prompt, kernel_size, pooling, merge)

    def reset(self, window_size=64, max_capacity_prompt=256 + 64, kernel_size=5, pooling='avgpool', merge=None):
        assert max_capacity_prompt - window_size > 0                       # :289
        self.window_size, self.max_capacity_prompt = window_size, max_capacity_prompt
        self.kernel_size, self.pooling, self.merge = kernel_size, pooling, merge

    def _prefill_shapes(self, key_states, query_states):
        assert key_states.shape[-2] == query_states.shape[-2]              # :309 "check if prefix phase"
        return query_states.shape                                          # bsz, num_heads, q_len, head_dim

    def _budget(self, q_len):
        return self.max_capacity_prompt - self.window_size                 # k of .topk(k) at :334

    _method = None

    def update_kv(self, key_states, query_states, value_states, attention_mask, num_key_value_groups):
        """(key_states', value_states') — `attention_mask` and `num_key_value_groups` are accepted and, as in the
        reference (:322 shadows the former, the latter is never read), do not influence the result."""
        bsz, num_heads, q_len, head_dim = self._prefill_shapes(key_states, query_states)
        n_keep = self._budget(q_len)
        _say(f"{self._name} max_capacity_prompt {n_keep if self._method == _kvc.PYRAMIDKV else self.max_capacity_prompt}")
        if q_len < self.max_capacity_prompt:                               # :314 pass-through, same objects
            return key_states, value_states
        _check_merge(self.merge)
        if self.merge is not None:                                         # :337-339 merge_kv instead of the gather
            _say("Pivot merge")                                            # :153
            if head_dim != 128:
                raise RuntimeError("merge_kv hard-codes head_dim 128 (pyramidkv_utils.py:149)")
            scoring = self._method != _kvc.STREAMINGLLM
            if scoring and self._method != _kvc.H2O and self.pooling not in ("avgpool", "maxpool"):
                raise ValueError('Pooling method not supported')            # :333
            return _kvc.compress_merge(self._method, query_states if scoring else None, key_states, value_states, self.window_size,
                                       n_keep, self.kernel_size, self.pooling if self._method != _kvc.H2O else None, TIE_MODE,
                                       n_q_heads=num_heads)
        if self._method == _kvc.STREAMINGLLM:
            return _kvc.compress(_kvc.STREAMINGLLM, None, key_states, value_states, self.window_size, n_keep,
                                 n_q_heads=num_heads)
        return _run(self._method, key_states, query_states, value_states, self.window_size, n_keep,
                    self.kernel_size, self.pooling)


class PrefillBatch:
    """The update_kv calls of one prompt, collected layer by layer and run as batched library calls.

    `add` does what update_kv does up to the compute (shape assert :309, budget, pass-through test :314, pooling /
    merge errors :333/:164) and keeps K, V and a copy of the W scoring query rows; `flush` groups entries of one
    layout, calls kvc_compress_batch per group and hands (key_states', value_states') to each entry's sink.

    Bounded and overlapped: once GROUP_LAYERS entries are parked the caller flushes them (`flush(overlap=True)`).  On
    the GPU that launch goes to a side stream which waits for the producing stream first, so the compression of
    layers 0..7 runs while the model computes layers 8..15 — the latency-bound exact top-k (~200 us per launch whatever
    the head count) disappears behind the model's own prefill, and at most GROUP_LAYERS layers of uncompressed K/V
    are alive at a time instead of the whole prompt's.  `settle()` makes the current stream wait for what the side
    stream produced; the patched forward calls it before anything reads the compressed cache."""

    def __init__(self):
        self.entries = []
        self._events = []               # (event recorded on the side stream, device, output tensors, input tensors kept alive)
        self._flushes = 0

    def __len__(self):
        return len(self.entries)

    def holds(self, tag):
        return any(e[0] is tag for e in self.entries)

    def add(self, cluster, key_states, query_states, value_states, sink, tag=None, alloc=None):
        """True: taken (sink(kc, vc) runs at flush).  False: not batchable here (pass-through, H2O — whose scoring
        needs every query row and is not launch-bound) — the caller runs cluster.update_kv itself.
        alloc(bsz, n_heads, capacity_rows, head_dim, dtype, device) -> (k_buf, v_buf): optional owner-provided output
        buffers (a decode cache with spare rows); K' / V' are then written straight into them and sink gets views."""
        bsz, num_heads, q_len, head_dim = cluster._prefill_shapes(key_states, query_states)
        if q_len < cluster.max_capacity_prompt or cluster._method in (_kvc.H2O, _kvc.L2NORM, _kvc.CAM) or cluster.merge is not None:
            return False                                             # (merge_kv takes the direct path: its own kernels)
        n_keep = cluster._budget(q_len)
        _say(f"{cluster._name} max_capacity_prompt {n_keep if cluster._method == _kvc.PYRAMIDKV else cluster.max_capacity_prompt}")
        _check_merge(cluster.merge)
        scoring = cluster._method != _kvc.STREAMINGLLM
        if scoring and cluster.pooling not in ("avgpool", "maxpool"):
            raise ValueError('Pooling method not supported')            # pyramidkv_utils.py:333
        W = cluster.window_size
        # the W scoring rows are read where the projection left them (a view: the library takes q_stride_h / q_stride_l);
        # round 2 copied them out per layer — a kernel launch and an allocation per update_kv on the host's critical path
        qw = _kvc._last_dim_contig(query_states)[:, :, q_len - W:, :] if scoring else None
        k, v = _kvc._last_dim_contig(key_states), _kvc._last_dim_contig(value_states)
        layout = (cluster._method, tuple(k.shape), k.stride(), v.stride(), qw.stride() if scoring else None, k.dtype, k.device,
                  num_heads, W, cluster.kernel_size if scoring else 0, cluster.pooling if scoring else None)
        self.entries.append((tag, layout, k, qw, v, n_keep, sink, alloc))
        return True

    def _release_done(self):
        """Inputs of earlier overlapped flushes whose side-stream work has finished are let go (their memory returns to the
        producing stream's pool; no record_stream bookkeeping per tensor)."""
        for i, (ev, dev, outs, held) in enumerate(self._events):
            if held is not None and ev.query():
                self._events[i] = (ev, dev, outs, None)

    def flush(self, overlap=False):
        entries, self.entries = self.entries, []
        groups = {}
        for e in entries:
            groups.setdefault(e[1], []).append(e)
        self._release_done()
        for layout, es in groups.items():
            method, kshape, _, _, _, dtype, dev, num_heads, W, kernel_size, pooling = layout
            scoring = method != _kvc.STREAMINGLLM
            keeps = tuple(e[5] for e in es)
            owned = all(e[7] is not None for e in es)
            side = None
            # overlapped only when the outputs go into owner-provided buffers: a sink that is handed FRESH tensors may read them
            # on its own stream right away (e.g. copy them into a cache) while the side stream is still writing
            if overlap and OVERLAP and owned and dev.type == "cuda":
                side = _side_stream(dev, self._flushes % N_SIDE_STREAMS)    # alternating: this group's latency-bound top-k runs
                self._flushes += 1                                          # beside the next group's K scan
                side.wait_stream(torch.cuda.current_stream(dev))            # K, V and the query rows are complete
                ctx = torch.cuda.stream(side)
            else:
                ctx = contextlib.nullcontext()
            with ctx:
                slab, outs = None, None
                if owned:                                   # every layer's cache sized for its OWN budget (per-item strides)
                    caps = [kk + W + SPARE_ROWS for kk in keeps]
                    owners = [getattr(e[7], "__self__", None) for e in es]
                    many = getattr(type(owners[0]), "reserve_many", None) if owners[0] is not None else None
                    if many is not None and all(type(o) is type(owners[0]) for o in owners):
                        slab, outs = many(owners, kshape[0], num_heads, caps, kshape[3], dtype, dev)
                    else:
                        outs = [e[7](kshape[0], num_heads, c, kshape[3], dtype, dev) for e, c in zip(es, caps)]
                kc, vc = _kvc.compress_batch(method, [e[3] for e in es] if scoring else None, [e[2] for e in es],
                                             [e[4] for e in es], W, keeps, kernel_size, pooling, TIE_MODE,
                                             q_rows="window" if scoring else "all", n_q_heads=num_heads, outs=outs, reuse=True)
            if side is not None:
                ev = torch.cuda.Event()
                ev.record(side)
                # inputs were allocated on the producing stream: they are held until the side stream has passed this event
                self._events.append((ev, dev, [slab] if slab is not None else [t for o in outs for t in o],
                                     [t for e in es for t in (e[2], e[3], e[4]) if t is not None]))
            for e, a, b in zip(es, kc, vc):
                e[6](a, b)

    def settle(self):
        """The current stream waits for every overlapped flush (no host synchronisation)."""
        for ev, dev, outs, held in self._events:
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ev)              # (whatever the current stream does to the held inputs' memory from here on is ordered
            for t in outs:                  # behind the side stream's reads: they may be dropped with this list)
                t.record_stream(cur)
        self._events = []


N_SIDE_STREAMS = 2
_SIDE = {}


def _side_stream(dev, i=0):
    if (dev, i) not in _SIDE:
        _SIDE[(dev, i)] = torch.cuda.Stream(device=dev)
    return _SIDE[(dev, i)]


class SnapKVCluster(_KVCluster):
    """pyramidkv_utils.py:285-347."""
    _name, _method = "SnapKV", _kvc.SNAPKV

    def __init__(self, window_size=64, max_capacity_prompt=256 + 64, kernel_size=5, pooling='avgpool', merge=None,
                 recent_size=32, ratio=0.4):
        super().__init__(window_size, max_capacity_prompt, kernel_size, pooling, merge)
        self.recent_size, self.ratio = recent_size, ratio                  # ThinK knobs (update_think)

    def update_think(self, key_states, query_states, value_states, attention_mask, num_key_value_groups):
        """pyramidkv_utils.py:349-392: update_kv, then key_pruner_query_driven (:13-26) on the compressed keys.
        Returns (kv_pruned [bsz, H, cap - recent_size, D - int(D * ratio)], kv_recent [bsz, H, recent_size, D], mask
        [bsz, H, D] bool — True = channel kept —, value_states'); the pass-through and merge branches return what update_kv
        returns (two tensors), like the reference (:360, :384)."""
        bsz, num_heads, q_len, head_dim = self._prefill_shapes(key_states, query_states)
        kc, vc = self.update_kv(key_states, query_states, value_states, attention_mask, num_key_value_groups)
        if q_len < self.max_capacity_prompt or self.merge is not None:
            return kc, vc
        kv_pruned, kv_recent, mask = _kvc.think_prune(kc, query_states, self.recent_size, self.ratio)
        return kv_pruned, kv_recent, mask, vc


class PyramidKVCluster(_KVCluster):
    """pyramidkv_utils.py:173-283: SnapKV's body with a per-layer budget (:205-215)."""
    _name, _method = "PyramidKV", _kvc.PYRAMIDKV

    def __init__(self, num_hidden_layers=32, window_size=64, max_capacity_prompt=256 + 64, kernel_size=5,
                 pooling='avgpool', beta=20, num_layers=80, layer_idx=None, merge=None):
        super().__init__(window_size, max_capacity_prompt, kernel_size, pooling, merge)
        self.layer_idx, self.num_hidden_layers = layer_idx, num_hidden_layers
        self.steps, self.beta = -1, beta

    def _budget(self, q_len):
        """k for this layer; -1 = pass-through.  Host arithmetic lives in the C library (kvc_pyramid_k)."""
        return _kvc.pyramid_k(self.max_capacity_prompt, self.window_size, q_len, self.layer_idx,
                              self.num_hidden_layers, self.beta)

    layer_budget = _budget


class H2OKVCluster(_KVCluster):
    """pyramidkv_utils.py:515-575: every query row scores; kernel_size / pooling are ignored (:555-561)."""
    _name, _method = "H2O", _kvc.H2O


class CAMKVCluster(_KVCluster):
    """pyramidkv_utils.py:431-513: SnapKV-style window scoring WITHOUT pooling (kernel_size / pooling are accepted
    and ignored, :468-476), a stochastic merge of every token's value into its `window_size` successors (:487-502), then top-k
    and gather of the keys and of the MERGED values.  On the GPU: kvc_cam_plan (scores + every token's merge probability — it
    does not depend on the merged values), the draw, kvc_cam_merge (the in-place value recurrence, bit-exact given the draws),
    kvc_select + kvc_gather.  `draw` is the reference's torch.bernoulli (:499) on the GPU: its random stream is not the CPU
    generator's, so outputs equal the reference's GIVEN THE SAME DRAWS (tests substitute the reference's recorded ones).
    Like the reference, value_states with H_q heads is merged IN PLACE (the caller's tensor; the reference's forward then
    attends over it); an H_kv-head tensor is expanded first (the draws differ per query head)."""
    _name, _method = "CAM", _kvc.CAM
    draw = staticmethod(torch.bernoulli)

    def __init__(self, start_budget_ratio=0.1, window_size=64, max_capacity_prompt=256 + 64, kernel_size=5, pooling='avgpool', merge=None):
        self.reset(start_budget_ratio, window_size, max_capacity_prompt, kernel_size, pooling, merge)

    def reset(self, start_budget_ratio=0.1, window_size=64, max_capacity_prompt=256 + 64, kernel_size=5, pooling='avgpool', merge=None):
        super().reset(window_size, max_capacity_prompt, kernel_size, pooling, merge)
        self.start_budget_ratio = start_budget_ratio

    def update_kv(self, key_states, query_states, value_states, attention_mask, num_key_value_groups):
        bsz, num_heads, q_len, head_dim = self._prefill_shapes(key_states, query_states)
        _say(f"CAM max_capacity_prompt {self.max_capacity_prompt}")                         # :455
        if q_len < self.max_capacity_prompt:                                                 # :457 pass-through, same objects
            return key_states, value_states
        W, n_keep = self.window_size, self.max_capacity_prompt - self.window_size
        start = math.ceil(self.start_budget_ratio * q_len)                                   # :479
        if value_states.shape[1] != num_heads:
            value_states = value_states.repeat_interleave(num_heads // value_states.shape[1], dim=1)
        elif value_states.stride(-1) != 1:
            raise RuntimeError("CAM merges value_states in place: the last dimension must be contiguous")
        if start + W < q_len:                                                                # the merge loop's range (:490) is not empty
            sc, prob = _kvc.cam_plan(query_states, key_states, W, start)
            mask = type(self).draw(prob.float()).to(torch.uint8).contiguous()
            _kvc.cam_merge(value_states, mask, W, start)
        else:
            sc = _kvc.scores(_kvc.SNAPKV, query_states, key_states, W, 1, None)
        idx = _kvc.select(sc, n_keep, TIE_MODE)
        return _kvc.gather(key_states, idx, W, num_heads), _kvc.gather(value_states, idx, W, num_heads)


class StreamingLLMKVCluster(_KVCluster):
    """pyramidkv_utils.py:578-620: no scoring, keeps the first cap-W and the last W tokens."""
    _name, _method = "StreamingLLM", _kvc.STREAMINGLLM


class L2NormCluster:
    """pyramidkv_utils.py:394-429: keep the max_capacity_prompt tokens of every head whose KEY has the smallest L2
    norm, in ascending norm order — no query, no window, no pooling.  Same constructor, reset and update_kv signature as the
    reference; the norms (torch-CPU's accumulation order), the ascending sort (torch-CPU's order of equal norms) and the
    gather run in the HIP library (kvc_l2norm_compress)."""
    _name, _method = "L2Norm", _kvc.L2NORM
    merge = None

    def __init__(self, max_capacity_prompt: int = 256 + 64, layer_idx: int = 0, skip_layers=()):
        self.reset(max_capacity_prompt, layer_idx, skip_layers)

    def reset(self, max_capacity_prompt: int = 256 + 64, layer_idx: int = 0, skip_layers=()):
        self.max_capacity_prompt, self.layer_idx, self.skip_layers = max_capacity_prompt, layer_idx, list(skip_layers)

    def _prefill_shapes(self, key_states, query_states):
        assert key_states.shape[-2] == query_states.shape[-2]              # :407 "check if prefix phase"
        return query_states.shape

    def update_kv(self, key_states, query_states, value_states, attention_mask, num_key_value_groups):
        bsz, num_heads, q_len, head_dim = self._prefill_shapes(key_states, query_states)
        _say(f"L2Norm max_capacity_prompt {self.max_capacity_prompt}")
        if q_len < self.max_capacity_prompt:                               # :413 pass-through, same objects
            return key_states, value_states
        if self.layer_idx in self.skip_layers:                             # :415
            return key_states, value_states
        return _kvc.l2norm_compress(key_states, value_states, self.max_capacity_prompt, n_q_heads=num_heads)


class _RaggedCluster:
    """AdaKV / HeadKV (pyramidkv_utils.py:622-878): `update_kv(key_states, query_states, value_states)` returns the FLATTENED
    compressed cache — `[sum_h (capacity_h + W), D]` keys and values, head after head — and leaves the reference's varlen
    metadata on the object: head_lens, cu_headlens, cu_klen, klen_sum, max_seqlen_k, layer_qlens, qlen_sum, cu_qlen,
    cu_offset, cu_head_offset (:684-699).  K/V may carry H_q heads (the reference) or the model's H_kv heads.
    `ragged` keeps the library's own handles (segment offsets / lengths on the device) for the in-place decode step."""

    _method = None

    def __init__(self, window_size=32, kernel_size=7, pooling='maxpool', max_capacity_prompt=None, layer_idx=None,
                 num_hidden_layers=None):
        self.window_size, self.kernel_size, self.pooling = window_size, kernel_size, pooling
        self.base_capacity = max_capacity_prompt - window_size
        self.num_hidden_layers, self.layer_idx = num_hidden_layers, layer_idx
        self.head_lens, self.max_seqlen_k, self.klen_sum, self.cu_klen = None, 0, 0, 0
        self.cu_offset = self.cu_headlens = None
        self.ragged = None

    # head_lens / cu_klen / max_seqlen_k / klen_sum are the PREFILL's values (:684-699).  The reference advances them per decode step
    # for its flash_attn_varlen call (llama_model.py:2371-2374); here the decode step reads seg_len + appended from the cache layer
    # (RaggedDynamicLayer), so they are not advanced.
    def _init_metadata(self, num_heads, k_lens, device):                       # :684-699
        i32 = dict(dtype=torch.int32, device=device)
        self.head_lens = torch.tensor(k_lens, **i32)
        self.klen_sum, self.max_seqlen_k = int(sum(k_lens)), int(max(k_lens))
        self.cu_headlens = torch.cumsum(self.head_lens, dim=0, dtype=torch.int32)
        self.cu_klen = torch.cat([self.cu_headlens - self.head_lens, torch.tensor([self.klen_sum], **i32)], dim=0)
        self.layer_qlens = torch.ones(num_heads, **i32)
        self.qlen_sum = num_heads
        self.cu_qlen = torch.cat([torch.cumsum(self.layer_qlens, dim=0, dtype=torch.int32) - self.layer_qlens,
                                  torch.tensor([self.qlen_sum], **i32)], dim=0)
        self.cu_offset = torch.arange(0, num_heads + 1, **i32)
        self.cu_head_offset = torch.arange(1, num_heads + 1, **i32)

    def _capacities(self):
        return None

    def update_kv(self, key_states, query_states, value_states, slack=0):
        bsz, num_heads, q_len, head_dim = query_states.shape
        assert bsz == 1                                                        # :719
        if self.pooling not in ("avgpool", "maxpool"):
            raise ValueError('Pooling method not supported')                   # :671
        if self.base_capacity > q_len - self.window_size:                      # :696 "not compress"
            g = num_heads // key_states.shape[1]
            k = key_states.repeat_interleave(g, 1) if g > 1 else key_states
            v = value_states.repeat_interleave(g, 1) if g > 1 else value_states
            self._init_metadata(num_heads, [q_len] * num_heads, key_states.device)
            kf, vf = k.reshape(-1, head_dim), v.reshape(-1, head_dim)
            if slack:                                                          # the same rows with spare rows behind every head
                kf = torch.cat([kf.view(num_heads, q_len, head_dim), kf.new_empty(num_heads, slack, head_dim)], 1).reshape(-1, head_dim)
                vf = torch.cat([vf.view(num_heads, q_len, head_dim), vf.new_empty(num_heads, slack, head_dim)], 1).reshape(-1, head_dim)
            dev = key_states.device
            self.ragged = dict(k_flat=kf, v_flat=vf, lens=[q_len] * num_heads, slack=slack,
                               seg_off=torch.arange(num_heads, dtype=torch.int64, device=dev) * (q_len + slack),
                               seg_len=torch.full((num_heads,), q_len, dtype=torch.int32, device=dev))
            return kf, vf
        r = _kvc.ragged_compress(self._method, query_states, key_states, value_states, self.window_size, self.base_capacity,
                                 self.kernel_size, self.pooling, getattr(self, "floor_ratio", 0.0), getattr(self, "normalize", False),
                                 self._capacities(), slack=slack, tie_mode=TIE_MODE)
        self._init_metadata(num_heads, r["lens"], key_states.device)
        self.ragged = r
        return r["k_flat"], r["v_flat"]


class AdaKVCluster(_RaggedCluster):
    """pyramidkv_utils.py:622-757: budgets from a global top-(H * base_capacity) over the heads' normalised scores, mixed with
    a floor: capacity_h = round(count_h * (1 - floor) + int(base_capacity * floor))."""
    _method = _kvc.ADAKV

    def __init__(self, window_size=32, kernel_size=7, pooling='maxpool', max_capacity_prompt=None, floor=None, normalize=None,
                 layer_idx=None, num_hidden_layers=None):
        super().__init__(window_size, kernel_size, pooling, max_capacity_prompt, layer_idx, num_hidden_layers)
        self.floor_ratio, self.normalize = floor, normalize
        self.floor_capacity = int(self.base_capacity * self.floor_ratio)
        self.adaptive_capacity = self.base_capacity - self.floor_capacity


class HeadKVCluster(_RaggedCluster):
    """pyramidkv_utils.py:760-878: capacities given per layer and head (`head_capacity[layer_idx][head]`)."""
    _method = _kvc.HEADKV

    def __init__(self, window_size=32, kernel_size=7, pooling='maxpool', max_capacity_prompt=None, layer_idx=None,
                 num_hidden_layers=None, head_capacity=None):
        super().__init__(window_size, kernel_size, pooling, max_capacity_prompt, layer_idx, num_hidden_layers)
        self.head_adaptive_capacity = head_capacity

    def _capacities(self):
        return [int(c) for c in self.head_adaptive_capacity[self.layer_idx]]


# ---- init_* factories (pyramidkv_utils.py:880-1031): default-fill self.config on first use, then rebuild
# ---- self.kv_cluster on EVERY forward so config edits between prompts take effect (clusters are stateless).
_DEFAULTS = (("window_size", 32), ("kernel_size", 5), ("pooling", "avgpool"), ("merge", None))


def _init(self, cls, default_cap, **extra):
    if not hasattr(self, "kv_cluster"):
        for name, val in _DEFAULTS + (("max_capacity_prompt", default_cap),):
            if not hasattr(self.config, name):
                setattr(self.config, name, val)
    c = self.config
    self.kv_cluster = cls(window_size=c.window_size, max_capacity_prompt=c.max_capacity_prompt,
                          kernel_size=c.kernel_size, pooling=c.pooling, merge=c.merge, **extra)


def init_pyramidkv(self, num_hidden_layers):
    """pyramidkv_utils.py:880-902 (default cap 2048)."""
    _init(self, PyramidKVCluster, 2048, num_hidden_layers=num_hidden_layers, layer_idx=self.layer_idx)


def init_snapkv(self):
    """pyramidkv_utils.py:904-924 (default cap 4096)."""
    _init(self, SnapKVCluster, 4096)


def init_H2O(self):
    """pyramidkv_utils.py:990-1009 (default cap 2048)."""
    _init(self, H2OKVCluster, 2048)


def init_CAM(self):
    """pyramidkv_utils.py:970-988 (default cap 2048; the reference reads config.merge without a default, :987: None here)."""
    _init(self, CAMKVCluster, 2048)


def init_StreamingLLM(self):
    """pyramidkv_utils.py:1011-1031 (default cap 2048)."""
    _init(self, StreamingLLMKVCluster, 2048)


def init_think(self):
    """pyramidkv_utils.py:926-952: SnapKV's defaults + recent_size 32, ratio 0.4; the caller uses kv_cluster.update_think."""
    _init(self, SnapKVCluster, 4096, **{name: _default(self.config, name, val) for name, val in (("recent_size", 32), ("ratio", 0.4))})


def _default(config, name, val):
    if not hasattr(config, name):
        setattr(config, name, val)
    return getattr(config, name)


def init_l2norm(self):
    """pyramidkv_utils.py:954-968 (defaults: cap 4096, skip_layers [0, 1]); the cluster carries the MODULE's layer index."""
    if not hasattr(self, "kv_cluster"):
        for name, val in (("max_capacity_prompt", 4096), ("layer_idx", 0), ("skip_layers", [0, 1])):
            if not hasattr(self.config, name):
                setattr(self.config, name, val)
    self.kv_cluster = L2NormCluster(max_capacity_prompt=self.config.max_capacity_prompt, layer_idx=self.layer_idx,
                                    skip_layers=self.config.skip_layers)


def init_adakv(self):
    """pyramidkv_utils.py:1033-1060.  (The reference fills `config.floor_ratio` but reads `config.floor`, :1043 / :1057: either
    name is honoured here.)  Built once per attention module, like the reference."""
    if not hasattr(self, "kv_cluster"):
        for name, val in (("window_size", 32), ("max_capacity_prompt", 2048), ("kernel_size", 5), ("pooling", "maxpool"),
                          ("floor_ratio", 0.2), ("normalize", True)):
            if not hasattr(self.config, name):
                setattr(self.config, name, val)
        c = self.config
        self.kv_cluster = AdaKVCluster(num_hidden_layers=c.num_hidden_layers, layer_idx=self.layer_idx, window_size=c.window_size,
                                       max_capacity_prompt=c.max_capacity_prompt, kernel_size=c.kernel_size, pooling=c.pooling,
                                       floor=getattr(c, "floor", c.floor_ratio), normalize=c.normalize)


def init_headkv(self):
    """pyramidkv_utils.py:1062-1084."""
    if not hasattr(self, "kv_cluster"):
        for name, val in (("window_size", 32), ("max_capacity_prompt", 2048), ("kernel_size", 5), ("pooling", "maxpool")):
            if not hasattr(self.config, name):
                setattr(self.config, name, val)
        if not hasattr(self.config, "head_capacity"):
            raise ValueError("Must have head_capacity")                           # :1073
        c = self.config
        self.kv_cluster = HeadKVCluster(num_hidden_layers=c.num_hidden_layers, layer_idx=self.layer_idx, window_size=c.window_size,
                                        max_capacity_prompt=c.max_capacity_prompt, kernel_size=c.kernel_size, pooling=c.pooling,
                                        head_capacity=c.head_capacity)
