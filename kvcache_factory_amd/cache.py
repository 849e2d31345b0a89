"""Cache layer that holds a compressed prefix but reports the TRUE sequence length.

Counterpart of the reference's `past_key_value._seen_tokens = self.kv_seq_len` (llama_model.py:290) and of the
`kv_seq_len` reset in `prepare_inputs_for_generation_llama_new` (llama_model.py:2609-2612): after compression the
cache stores `cap` tokens per head, but RoPE positions / cache_position for the decode steps must continue from
the real prompt length.  transformers 5.x derives both from `layer.get_seq_length()`, so the layer keeps them apart.
"""
import torch
from transformers.cache_utils import DynamicLayer


class CompressedDynamicLayer(DynamicLayer):
    """Stores [bsz, H_q, stored, D] (H_q heads, like the reference's cache after repeat_kv) and counts real tokens."""

    def __init__(self):
        super().__init__()
        self.true_length = 0            # tokens the model has really seen for this layer

    def prefill(self, k_compressed, v_compressed, true_length):
        if not self.is_initialized:
            self.lazy_initialization(k_compressed, v_compressed)
        self.keys, self.values = k_compressed, v_compressed
        self.true_length = int(true_length)

    def update(self, key_states, value_states, *args, **kwargs):
        keys, values = super().update(key_states, value_states, *args, **kwargs)
        self.true_length += key_states.shape[-2]
        return keys, values

    def stored_length(self):
        return 0 if (not self.is_initialized or self.keys.numel() == 0) else self.keys.shape[-2]

    def get_seq_length(self):
        return self.true_length

    def get_mask_sizes(self, query_length):
        # the mask must be as wide as what attention will really see: stored tokens + the new ones
        return self.stored_length() + query_length, 0

    def crop(self, *a, **kw):
        raise NotImplementedError("a compressed cache cannot be cropped by position")
