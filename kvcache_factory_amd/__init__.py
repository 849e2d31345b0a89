"""MI355X-native KV-cache compression hot path (drop-in for KVCache-Factory's update_kv path).

Public surface mirrors the reference (assassin808/KVCache-Factory @ 2025-02-18):
    kvcache_factory_amd.monkeypatch.replace_llama / replace_mistral      (pyramidkv/monkeypatch.py:19,92)
    kvcache_factory_amd.pyramidkv_utils.{SnapKV,PyramidKV,H2OKV,StreamingLLMKV}Cluster.update_kv
                                                                         (pyramidkv/pyramidkv_utils.py:197,306,533,595)
    kvcache_factory_amd.pyramidkv_utils.init_{pyramidkv,snapkv,H2O,StreamingLLM}   (:880-1031)
All compute runs in hand-written HIP kernels (csrc/, libkvc_hip.so) through the C-ABI of include/kvc.h;
there is no CPU or eager-PyTorch fallback: if the extension is missing the import of `_kvc` raises.
"""
__version__ = "0.1.0"
