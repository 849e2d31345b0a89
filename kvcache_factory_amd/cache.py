"""Cache layer that holds a compressed prefix but reports the TRUE sequence length.

Counterpart of the reference's `past_key_value._seen_tokens = self.kv_seq_len` (llama_model.py:290) and of the
`kv_seq_len` reset in `prepare_inputs_for_generation_llama_new` (llama_model.py:2609-2612): after compression the
cache stores `cap` tokens per head, but RoPE positions / cache_position for the decode steps must continue from
the real prompt length.  transformers 5.x derives both from `layer.get_seq_length()`, so the layer keeps them apart.
"""
import torch
from transformers.cache_utils import DynamicLayer


class CompressedDynamicLayer(DynamicLayer):
    """Stores [bsz, H_q, stored, D] (H_q heads, like the reference's cache after repeat_kv) and counts real tokens.

    Decode side (first step): the reference's cache grows by `torch.cat` on every generated token
    (`past_key_value.update` else-branch, llama_model.py:287-289) — a copy of the whole [1, 32, cap + t, 128] tensor
    per layer per step.  Here the compressed prompt is placed once into a buffer with `RESERVE` spare rows per head
    (doubled when exhausted); a decode step writes its row(s) in place and hands out views of the filled part.  Same
    bytes, no per-token reallocation."""

    RESERVE = 256                       # spare rows per head allocated with the compressed prompt

    def __init__(self):
        super().__init__()
        self.true_length = 0            # tokens the model has really seen for this layer
        self._kbuf = self._vbuf = None  # [bsz, H_q, capacity, D]
        self._stored = 0

    def _place(self, k, v, spare):
        b, h, n, d = k.shape
        self._kbuf = torch.empty(b, h, n + spare, d, dtype=k.dtype, device=k.device)
        self._vbuf = torch.empty_like(self._kbuf)
        self._kbuf[:, :, :n].copy_(k)
        self._vbuf[:, :, :n].copy_(v)
        self._stored = n
        self.keys, self.values = self._kbuf[:, :, :n], self._vbuf[:, :, :n]

    def reserve(self, bsz, n_heads, capacity, head_dim, dtype, device):
        """Buffers [bsz, H_q, capacity, D] the compression kernels write K' / V' into directly (kvc_params.out_stride_h):
        the compressed prompt is produced in its final home, `prefill` then only records its length."""
        self._kbuf = torch.empty(bsz, n_heads, capacity, head_dim, dtype=dtype, device=device)
        self._vbuf = torch.empty_like(self._kbuf)
        self._stored = 0
        return self._kbuf, self._vbuf

    def prefill(self, k_compressed, v_compressed, true_length):
        if not self.is_initialized:
            self.lazy_initialization(k_compressed, v_compressed)
        if self._kbuf is not None and k_compressed.data_ptr() == self._kbuf.data_ptr() and \
                v_compressed.data_ptr() == self._vbuf.data_ptr():      # already written into our own buffers
            self._stored = k_compressed.shape[2]
            self.keys, self.values = self._kbuf[:, :, :self._stored], self._vbuf[:, :, :self._stored]
        else:
            self._place(k_compressed, v_compressed, self.RESERVE)
        self.true_length = int(true_length)

    def update(self, key_states, value_states, *args, **kwargs):
        if not self.is_initialized:
            self.lazy_initialization(key_states, value_states)
        t = key_states.shape[-2]
        if self._kbuf is None:                                    # a layer that never went through prefill()
            self._place(key_states, value_states, self.RESERVE)
        else:
            if self._stored + t > self._kbuf.shape[2]:            # out of spare rows: double
                self._place(self.keys, self.values, max(self._kbuf.shape[2], t))
            n = self._stored
            self._kbuf[:, :, n:n + t].copy_(key_states)
            self._vbuf[:, :, n:n + t].copy_(value_states)
            self._stored = n + t
            self.keys, self.values = self._kbuf[:, :, :n + t], self._vbuf[:, :, :n + t]
        self.true_length += t
        return self.keys, self.values

    def stored_length(self):
        return 0 if (not self.is_initialized or self.keys.numel() == 0) else self.keys.shape[-2]

    def get_seq_length(self):
        return self.true_length

    def get_mask_sizes(self, query_length):
        # the mask must be as wide as what attention will really see: stored tokens + the new ones
        return self.stored_length() + query_length, 0

    def crop(self, *a, **kw):
        raise NotImplementedError("a compressed cache cannot be cropped by position")
