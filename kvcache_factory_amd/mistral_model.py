"""Mistral attention forwards (reference: pyramidkv/mistral_model.py, sdpa PyramidKV :1881-2023).  Same template as
Llama; the only difference in transformers 5.x is the `sliding_window` kwarg handed to the attention function."""
from transformers.models.mistral import modeling_mistral as _mm

from .attention import make_forward, make_ragged_forward


def _mk(method):
    return make_forward(method, _mm.apply_rotary_pos_emb, _mm.eager_attention_forward, _mm.repeat_kv,
                        pass_sliding_window=True)


mistral_attn_forward_PyramidKV = mistral_sdpa_attn_forward_PyramidKV = _mk("pyramidkv")
mistral_attn_forward_SnapKV = mistral_sdpa_attn_forward_SnapKV = _mk("snapkv")
mistral_attn_forward_H2O = mistral_sdpa_attn_forward_H2O = _mk("h2o")
mistral_attn_forward_StreamingLLM = mistral_sdpa_attn_forward_StreamingLLM = _mk("streamingllm")
mistral_attn_forward_CAM = mistral_sdpa_attn_forward_CAM = _mk("cam")                        # mistral_model.py CAM forwards (monkeypatch.py:39-43, :112-116)
mistral_attn_forward_L2Norm = mistral_sdpa_attn_forward_L2Norm = _mk("l2norm")        # mistral_model.py:513, :622
mistral_flash_attn2_forward_AdaKV = make_ragged_forward("adakv", _mm.apply_rotary_pos_emb, _mm.eager_attention_forward, _mm.repeat_kv)
mistral_flash_attn2_forward_HeadKV = make_ragged_forward("headkv", _mm.apply_rotary_pos_emb, _mm.eager_attention_forward, _mm.repeat_kv)
FORWARDS = {"pyramidkv": mistral_attn_forward_PyramidKV, "snapkv": mistral_attn_forward_SnapKV,
            "h2o": mistral_attn_forward_H2O, "streamingllm": mistral_attn_forward_StreamingLLM,
            "adakv": mistral_flash_attn2_forward_AdaKV, "headkv": mistral_flash_attn2_forward_HeadKV,
            "l2norm": mistral_attn_forward_L2Norm, "cam": mistral_attn_forward_CAM}
