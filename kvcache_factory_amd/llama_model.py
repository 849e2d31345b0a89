"""Llama attention forwards for the four in-scope methods (reference: pyramidkv/llama_model.py — eager :87,
sdpa :208/:1305/:1663/:2020 and flash-attn copies of one template).  transformers 5.x has ONE LlamaAttention class
whose backend is picked at call time, so one forward per method replaces the reference's three."""
from transformers.models.llama import modeling_llama as _ml

from .attention import make_forward, make_ragged_forward, make_think_forward


def _mk(method):
    return make_forward(method, _ml.apply_rotary_pos_emb, _ml.eager_attention_forward, _ml.repeat_kv)


llama_attn_forward_PyramidKV = llama_sdpa_attn_forward_PyramidKV = _mk("pyramidkv")
llama_attn_forward_SnapKV = llama_sdpa_attn_forward_SnapKV = _mk("snapkv")
llama_attn_forward_H2O = llama_sdpa_attn_forward_H2O = _mk("h2o")
llama_attn_forward_StreamingLLM = llama_sdpa_attn_forward_StreamingLLM = _mk("streamingllm")
llama_attn_forward_CAM = llama_sdpa_attn_forward_CAM = _mk("cam")                        # llama_model.py CAM forwards (monkeypatch.py:39-43, :112-116)
llama_attn_forward_L2Norm = llama_sdpa_attn_forward_L2Norm = _mk("l2norm")            # llama_model.py:455, :576
# AdaKV / HeadKV: the reference only has flash-attn forwards for them (llama_model.py:2255, :2400)
llama_flash_attn2_forward_AdaKV = make_ragged_forward("adakv", _ml.apply_rotary_pos_emb, _ml.eager_attention_forward, _ml.repeat_kv)
llama_flash_attn2_forward_HeadKV = make_ragged_forward("headkv", _ml.apply_rotary_pos_emb, _ml.eager_attention_forward, _ml.repeat_kv)
# ThinK: llama_model_think.py:86 (the reference patches Llama only, monkeypatch.py:80-83)
llama_attn_forward_SnapKV_ThinK = make_think_forward(_ml.apply_rotary_pos_emb, _ml.eager_attention_forward, _ml.repeat_kv)
FORWARDS = {"pyramidkv": llama_attn_forward_PyramidKV, "snapkv": llama_attn_forward_SnapKV,
            "h2o": llama_attn_forward_H2O, "streamingllm": llama_attn_forward_StreamingLLM,
            "adakv": llama_flash_attn2_forward_AdaKV, "headkv": llama_flash_attn2_forward_HeadKV,
            "l2norm": llama_attn_forward_L2Norm, "cam": llama_attn_forward_CAM, "think": llama_attn_forward_SnapKV_ThinK}
