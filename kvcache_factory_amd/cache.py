"""Cache layer that holds a compressed prefix but reports the TRUE sequence length.

Counterpart of the reference's `past_key_value._seen_tokens = self.kv_seq_len` (llama_model.py:290) and of the
`kv_seq_len` reset in `prepare_inputs_for_generation_llama_new` (llama_model.py:2609-2612): after compression the
cache stores `cap` tokens per head, but RoPE positions / cache_position for the decode steps must continue from
the real prompt length.  transformers 5.x derives both from `layer.get_seq_length()`, so the layer keeps them apart.
"""
import torch
from transformers.cache_utils import DynamicLayer


def _placeholder_only(name):
    """Setter of the computed `keys` / `values` properties: DynamicLayer's constructor assigns None (and its stock
    lazy_initialization an empty tensor); anything else would be silently lost, so it is refused."""
    def setter(self, value):
        if value is not None and getattr(value, "numel", lambda: 1)() != 0:
            raise AttributeError(f"{type(self).__name__}.{name} is computed from the layer's own buffers; DynamicLayer operations that "
                                 "reassign it are overridden (batch reordering) or refused (crop, offload)")
    return setter


class _BatchOps:
    """The DynamicLayer operations that work by REASSIGNING keys / values (transformers 5.15: reorder_cache, batch_repeat_interleave,
    batch_select_indices — beam search, num_return_sequences, contrastive search — and offload / prefetch / reset), restated on the
    layer's own buffers.  `_batch_tensors` names the attributes that carry a batch dimension in front."""
    _batch_tensors = ()

    def _map_batch(self, fn):
        for name in self._batch_tensors:
            t = getattr(self, name)
            if t is not None:
                setattr(self, name, fn(t))

    def reorder_cache(self, beam_idx):
        if self.get_seq_length() > 0:
            self._map_batch(lambda t: t.index_select(0, beam_idx.to(t.device)))

    def batch_repeat_interleave(self, repeats):
        if self.get_seq_length() > 0:
            self._map_batch(lambda t: t.repeat_interleave(repeats, dim=0))

    def batch_select_indices(self, indices):
        if self.get_seq_length() > 0:
            self._map_batch(lambda t: t[indices, ...])

    def offload(self):
        raise NotImplementedError("a compressed cache stays in the device buffers its kernels append to in place: no offloading")

    def prefetch(self):
        pass                                                            # never offloaded

    def lazy_initialization(self, key_states, value_states=None):
        """DynamicLayer's version also allocates two empty placeholder tensors for `keys` / `values`; here those are computed
        properties, so only the bookkeeping is kept (two allocations per layer and prompt less on the prefill path)."""
        self.dtype, self.device = key_states.dtype, key_states.device
        self.is_initialized = True

    def crop(self, *a, **kw):
        raise NotImplementedError("a compressed cache cannot be cropped by position")


class CompressedDynamicLayer(_BatchOps, DynamicLayer):
    """The compressed prompt of one layer plus the decoded tokens, in one of two forms.

    EXPANDED (the reference's form): `[bsz, H_q, stored, D]` — H_q heads like the reference's cache after repeat_kv.  The
    reference grows it by `torch.cat` on every generated token (`past_key_value.update` else-branch, llama_model.py:287-289:
    a copy of the whole [1, 32, cap + t, 128] tensor per layer per step); here the compressed prompt sits in a buffer with
    spare rows per head (doubled when exhausted) and a decode step writes its row(s) in place.  Same bytes.

    SPLIT (used by the fused decode step, `decode_attend`): the prefix `[bsz, H_q, k + W, D]` stays exactly
    where the compression kernels wrote it and the decoded tokens go to a TAIL `[bsz, H_kv, capacity, D]` — once per KV
    head, not H_q / H_kv times, with no repeat_kv and no copy of the prefix.  `keys` / `values` materialise the reference's
    view on demand (tests, tools); asking for them through `update` converts the layer to EXPANDED for good."""

    RESERVE = 256                       # spare rows allocated with the compressed prompt / tail rows allocated at a time

    def __init__(self):
        super().__init__()
        self.true_length = 0            # tokens the model has really seen for this layer
        self._kbuf = self._vbuf = None  # [bsz, H_q, capacity, D]
        self._stored = 0                # rows in use per head of _kbuf / _vbuf
        self._ktail = self._vtail = None    # SPLIT form: [bsz, H_kv, capacity, D]
        self._tail = 0
        self._split = False

    # ---- the reference's view --------------------------------------------------------------------------------------
    # (`keys` / `values` are plain attributes of DynamicLayer; here they are computed from the buffers)
    keys = property(lambda self: self._view(self._kbuf, self._ktail), _placeholder_only("keys"))
    values = property(lambda self: self._view(self._vbuf, self._vtail), _placeholder_only("values"))
    _batch_tensors = ("_kbuf", "_vbuf", "_ktail", "_vtail")

    def reset(self):
        """Back to an empty layer (the next forward is a prefill): buffers, lengths and the split state go."""
        self._kbuf = self._vbuf = self._ktail = self._vtail = None
        self._stored = self._tail = self.true_length = 0
        self._split = False

    def _view(self, buf, tail):
        if buf is None:
            return None
        pre = buf[:, :, :self._stored]
        if not self._split or self._tail == 0:
            return pre
        g = buf.shape[1] // tail.shape[1]
        return torch.cat([pre, tail[:, :, :self._tail].repeat_interleave(g, dim=1)], dim=2)

    # ---- prefill ------------------------------------------------------------------------------------------------------
    def _place(self, k, v, spare):
        b, h, n, d = k.shape
        kb = torch.empty(b, h, n + spare, d, dtype=k.dtype, device=k.device)
        vb = torch.empty_like(kb)
        kb[:, :, :n].copy_(k)
        vb[:, :, :n].copy_(v)
        self._kbuf, self._vbuf, self._stored = kb, vb, n

    def reserve(self, bsz, n_heads, capacity, head_dim, dtype, device):
        """Buffers [bsz, H_q, capacity, D] the compression kernels write K' / V' into directly (kvc_params.out_stride_h):
        the compressed prompt is produced in its final home, `prefill` then only records its length."""
        self._kbuf = torch.empty(bsz, n_heads, capacity, head_dim, dtype=dtype, device=device)
        self._vbuf = torch.empty_like(self._kbuf)
        self._stored = 0
        return self._kbuf, self._vbuf

    @classmethod
    def reserve_many(cls, layers, bsz, n_heads, capacities, head_dim, dtype, device):
        """`reserve` for a group of layers out of ONE allocation (the grouped flush of PrefillBatch: one torch.empty and one
        stream bookkeeping call per group instead of two per layer).  The slab lives as long as any layer's views do."""
        per = [bsz * n_heads * c * head_dim for c in capacities]
        slab = torch.empty(2 * sum(per), dtype=dtype, device=device)
        outs, off = [], 0
        for layer, c, n in zip(layers, capacities, per):
            layer._kbuf = slab[off:off + n].view(bsz, n_heads, c, head_dim)
            layer._vbuf = slab[off + n:off + 2 * n].view(bsz, n_heads, c, head_dim)
            layer._stored = 0
            outs.append((layer._kbuf, layer._vbuf))
            off += 2 * n
        return slab, outs

    def prefill(self, k_compressed, v_compressed, true_length):
        if not self.is_initialized:
            self.lazy_initialization(k_compressed, v_compressed)
        if self._kbuf is not None and k_compressed.data_ptr() == self._kbuf.data_ptr() and \
                v_compressed.data_ptr() == self._vbuf.data_ptr():      # already written into our own buffers
            self._stored = k_compressed.shape[2]
        else:
            self._place(k_compressed, v_compressed, self.RESERVE)
        self._split, self._tail, self._ktail, self._vtail = False, 0, None, None
        self.true_length = int(true_length)

    # ---- decode: the reference-shaped path ----------------------------------------------------------------------------
    def _expand(self):
        """SPLIT -> EXPANDED (a caller wants plain H_q-head tensors to attend over)."""
        if self._split:
            k, v = self.keys, self.values
            self._split, self._tail, self._ktail, self._vtail = False, 0, None, None
            self._place(k, v, self.RESERVE)

    def update(self, key_states, value_states, *args, **kwargs):
        """Append H_q-head rows (the caller repeat_kv-expanded them like the reference) and return the whole cache."""
        if not self.is_initialized:
            self.lazy_initialization(key_states, value_states)
        t = key_states.shape[-2]
        self._expand()
        if self._kbuf is None:                                    # a layer that never went through prefill()
            self._place(key_states, value_states, self.RESERVE)
        else:
            if self._stored + t > self._kbuf.shape[2]:            # out of spare rows: double
                self._place(self._kbuf[:, :, :self._stored], self._vbuf[:, :, :self._stored], max(self._kbuf.shape[2], t))
            n = self._stored
            self._kbuf[:, :, n:n + t].copy_(key_states)
            self._vbuf[:, :, n:n + t].copy_(value_states)
            self._stored = n + t
        self.true_length += t
        return self.keys, self.values

    # ---- decode: the fused step ---------------------------------------------------------------------------------------
    def decode_attend(self, query_states, key_states, value_states, scaling):
        """One decode step in the SPLIT form: key_states / value_states [bsz, H_kv, t, D] are appended to the tail by the
        kernel and the attention output [bsz, t, H_q, D] over prefix + tail comes back (kvc_decode_step)."""
        from . import _kvc
        t = key_states.shape[-2]
        b, hkv, _, d = key_states.shape
        if not self._split:
            self._split, self._tail = True, 0
            self._ktail = torch.empty(b, hkv, self.RESERVE, d, dtype=key_states.dtype, device=key_states.device)
            self._vtail = torch.empty_like(self._ktail)
        if self._tail + t > self._ktail.shape[2]:                 # tail full: double it
            cap = max(2 * self._ktail.shape[2], self._tail + t)
            kt = torch.empty(b, hkv, cap, d, dtype=self._ktail.dtype, device=self._ktail.device)
            vt = torch.empty_like(kt)
            kt[:, :, :self._tail].copy_(self._ktail[:, :, :self._tail])
            vt[:, :, :self._tail].copy_(self._vtail[:, :, :self._tail])
            self._ktail, self._vtail = kt, vt
        out = _kvc.decode_step(query_states, key_states, value_states, self._kbuf, self._vbuf, self._stored,
                               self._ktail, self._vtail, self._tail, scaling)
        self._tail += t
        self.true_length += t
        return out

    def can_split(self):
        """The fused decode step needs the prefix in this layer's own buffers (a layer that went through prefill)."""
        return self._kbuf is not None

    # ---- bookkeeping --------------------------------------------------------------------------------------------------
    def stored_length(self):
        if self._kbuf is None:
            return 0
        return self._stored + (self._tail if self._split else 0)

    def get_seq_length(self):
        return self.true_length

    def get_mask_sizes(self, query_length):
        # the mask must be as wide as what attention will really see: stored tokens + the new ones
        return self.stored_length() + query_length, 0


class RaggedDynamicLayer(_BatchOps, DynamicLayer):
    """The flattened cache of AdaKV / HeadKV for one layer — counterpart of the reference's DynamicCacheSplitHeadFlatten
    (pyramidkv_utils.py:28-102), whose every decode step allocates a new tensor and re-copies the whole cache with one row
    inserted per head (update_flatten_view, csrc/csrc/cuda_api.cu:12-85).  Here every head's segment has spare rows behind it
    and kvc_ragged_decode_step appends in place; when a segment fills up, the cache is re-laid out once with twice the slack."""

    def __init__(self):
        super().__init__()
        self.true_length = 0
        self.k_flat = self.v_flat = self.seg_off = self.seg_len = None
        self.lens, self.slack, self.appended = [], 0, 0

    keys = property(lambda self: self.k_flat, _placeholder_only("keys"))
    values = property(lambda self: self.v_flat, _placeholder_only("values"))

    def _no_batch(self, *a, **kw):
        # the flattened cache has no batch dimension to reorder (the reference's DynamicCacheSplitHeadFlatten has none either, and
        # its README lists batch inference as unsupported): refuse instead of decoding against un-reordered histories
        raise NotImplementedError("AdaKV / HeadKV keep one flattened cache per layer: beam search / num_return_sequences > 1 are not supported")
    reorder_cache = batch_repeat_interleave = batch_select_indices = _no_batch

    def reset(self):
        self.k_flat = self.v_flat = self.seg_off = self.seg_len = None
        self.lens, self.slack, self.appended, self.true_length = [], 0, 0, 0

    def prefill(self, ragged, true_length):
        if not self.is_initialized:
            self.lazy_initialization(ragged["k_flat"], ragged["v_flat"])
        self.k_flat, self.v_flat = ragged["k_flat"], ragged["v_flat"]
        self.seg_off, self.seg_len = ragged["seg_off"], ragged["seg_len"]
        self.lens, self.slack, self.appended = list(ragged["lens"]), ragged["slack"], 0
        self.true_length = int(true_length)

    def _regrow(self, need):
        slack = max(2 * self.slack, need)
        offs, row = [], 0
        for ln in self.lens:
            offs.append(row)
            row += ln + slack
        new_off = torch.tensor(offs, dtype=torch.int64, device=self.k_flat.device)
        kf = torch.empty(row, self.k_flat.shape[1], dtype=self.k_flat.dtype, device=self.k_flat.device)
        vf = torch.empty_like(kf)
        old = self.seg_off.tolist()
        for h, ln in enumerate(self.lens):
            m = ln + self.appended
            kf[offs[h]:offs[h] + m].copy_(self.k_flat[old[h]:old[h] + m])
            vf[offs[h]:offs[h] + m].copy_(self.v_flat[old[h]:old[h] + m])
        self.k_flat, self.v_flat, self.seg_off, self.slack = kf, vf, new_off, slack

    def decode_attend(self, query_states, key_states, value_states, scaling):
        from . import _kvc
        t = key_states.shape[-2]
        if self.appended + t > self.slack:
            self._regrow(self.appended + t)
        out = _kvc.ragged_decode_step(query_states, key_states, value_states, self.k_flat, self.v_flat, self.seg_off, self.seg_len,
                                      self.appended, scaling, slack=self.slack)
        self.appended += t
        self.true_length += t
        return out

    def segments(self):
        """Per head: (keys, values) views of its rows (prefill + appended) — for tests and tools."""
        offs = self.seg_off.tolist()
        return [(self.k_flat[o:o + ln + self.appended], self.v_flat[o:o + ln + self.appended]) for o, ln in zip(offs, self.lens)]

    def get_seq_length(self):
        return self.true_length

    def get_mask_sizes(self, query_length):
        return max(self.lens or [0]) + self.appended + query_length, 0


class ThinkDynamicLayer(_BatchOps, DynamicLayer):
    """ThinK's cache for one layer — counterpart of the reference's cache_utils_think.DynamicCache (key_cache_pruned, mask,
    key_cache, value_cache; `update_think`, cache_utils_think.py:390-424, llama_model_think.py:167-168): the compressed keys are
    stored WITHOUT their pruned channels except the last `recent_size` rows; decoded tokens join the full-channel rows.  A decode
    step attends over [pruned rows | recent + decoded rows] with the query's kept channels for the first part
    (llama_model_think.py:175-196) — here ONE kvc_decode_step2 call (round 2: two torch matmuls, a cat, a softmax and a matmul):
    the pruned keys are kept once more in the kernel's form (kept channels in ascending order, zero-padded to 16 bytes, with the
    channel list), the recent rows and all values stay where update_think wrote them, and the decoded tokens go to a tail stored
    once per KV head and appended in place, as in CompressedDynamicLayer."""
    RESERVE = 256

    def __init__(self):
        super().__init__()
        self.true_length = 0
        self.key_pruned = None          # [bsz, H, P1, Dk] (the reference's layout) or None (pass-through prompt: nothing pruned)
        self.mask = None                # [bsz, H, D] bool, True = channel kept
        self._kp = self._chan = None    # kernel form: [bsz, H, P1, Dk padded], [bsz, H, Dk padded] int16 (-1 = padding)
        self._kbuf = self._vbuf = None  # recent keys [bsz, H, Pr, D]; ALL values of the compressed prompt [bsz, H, P1 + Pr, D]
        self._ktail = self._vtail = None    # decoded tokens [bsz, H_kv, capacity, D]
        self._tail = 0

    def _expanded(self, buf, tail):
        if buf is None:
            return None
        if tail is None or self._tail == 0:
            return buf
        return torch.cat([buf, tail[:, :, :self._tail].repeat_interleave(buf.shape[1] // tail.shape[1], dim=1)], dim=2)

    # the reference's views: key_cache (full-channel rows: recent + decoded) and value_cache (every row)
    keys = property(lambda self: self._expanded(self._kbuf, self._ktail), _placeholder_only("keys"))
    values = property(lambda self: self._expanded(self._vbuf, self._vtail), _placeholder_only("values"))
    _batch_tensors = ("key_pruned", "mask", "_kp", "_chan", "_kbuf", "_vbuf", "_ktail", "_vtail")

    def reset(self):
        self.key_pruned = self.mask = self._kp = self._chan = self._kbuf = self._vbuf = self._ktail = self._vtail = None
        self._tail = self.true_length = 0

    def prefill_think(self, kv_pruned, kv_recent, mask, values, true_length):
        if not self.is_initialized:
            self.lazy_initialization(kv_recent, values)
        self.key_pruned, self.mask = kv_pruned, mask
        b, h, p1, dk = kv_pruned.shape
        per16 = 16 // kv_pruned.element_size()
        dkp = (dk + per16 - 1) // per16 * per16
        self._kp = torch.nn.functional.pad(kv_pruned, (0, dkp - dk)).contiguous()
        # kept channels in ascending order (the order of key_states[mask], pyramidkv_utils.py:24-25): a stable sort of the mask
        order = torch.sort(mask.to(torch.int8), dim=-1, descending=True, stable=True).indices[..., :dk]
        self._chan = torch.nn.functional.pad(order.to(torch.int16), (0, dkp - dk), value=-1).contiguous()
        self._kbuf, self._vbuf = kv_recent.contiguous(), values.contiguous()
        self._ktail = self._vtail = None
        self._tail = 0
        self.true_length = int(true_length)

    def prefill_plain(self, keys, values, true_length):
        """A prompt shorter than the budget (or a merged one): update_think returned plain (keys, values)."""
        if not self.is_initialized:
            self.lazy_initialization(keys, values)
        self.key_pruned = self.mask = self._kp = self._chan = None
        self._kbuf, self._vbuf = keys.contiguous(), values.contiguous()
        self._ktail = self._vtail = None
        self._tail = 0
        self.true_length = int(true_length)

    def attend(self, query_states, k_new, v_new, scaling):
        """The step's K/V rows [bsz, H_kv, t, D] go to the tail and the attention output [bsz, t, H, D] over pruned rows, recent
        rows and tail comes back — kvc_decode_step2 (llama_model_think.py:170-196)."""
        from . import _kvc
        b, hkv, t, d = k_new.shape
        if self._ktail is None:
            self._ktail = torch.empty(b, hkv, max(self.RESERVE, 2 * t), d, dtype=k_new.dtype, device=k_new.device)
            self._vtail = torch.empty_like(self._ktail)
        if self._tail + t > self._ktail.shape[2]:                 # tail full: double it
            cap = max(2 * self._ktail.shape[2], self._tail + t)
            kt = torch.empty(b, hkv, cap, d, dtype=self._ktail.dtype, device=self._ktail.device)
            vt = torch.empty_like(kt)
            kt[:, :, :self._tail].copy_(self._ktail[:, :, :self._tail])
            vt[:, :, :self._tail].copy_(self._vtail[:, :, :self._tail])
            self._ktail, self._vtail = kt, vt
        p1 = 0 if self._kp is None else self._kp.shape[2]
        out = _kvc.decode_step(query_states, k_new, v_new, self._kbuf, self._vbuf, self._kbuf.shape[2], self._ktail, self._vtail,
                               self._tail, scaling, k_pruned=self._kp, chan=self._chan, pruned_rows=p1)
        self._tail += t
        self.true_length += t
        return out

    def stored_length(self):
        return (0 if self._vbuf is None else self._vbuf.shape[2]) + self._tail

    def get_seq_length(self):
        return self.true_length

    def get_mask_sizes(self, query_length):
        return self.stored_length() + query_length, 0
