"""replace_llama / replace_mistral — drop-in for pyramidkv/monkeypatch.py:19-87, :92-145.

Call once BEFORE building the model, exactly like the reference (run_longbench.py:382-384); the per-layer knobs are
then written onto `model.model.layers[i].self_attn.config.*` (run_longbench.py:253-261) and read by init_*.
Methods in scope: "pyramidkv", "snapkv", "h2o", "streamingllm", "adakv", "headkv", (N4) "l2norm", "cam" and "think"
(Llama only, like the reference, :80-83); "fullkv" leaves the model untouched (:86).  The reference's remaining method string
(minference) is out of scope and raises.  (The reference's replace_mistral("cam") patches the LLAMA classes, monkeypatch.py:112-116 —
a slip not reproduced: Mistral gets the CAM forward.)
The reference also rebinds `prepare_inputs_for_generation` to reset `kv_seq_len` (llama_model.py:2598-2612); with
transformers 5.x that bookkeeping lives in the cache layer (cache.CompressedDynamicLayer), so nothing else is patched.
"""
import transformers

_IN_SCOPE = ("pyramidkv", "snapkv", "h2o", "streamingllm", "adakv", "headkv", "l2norm", "think", "cam")
_OUT_OF_SCOPE = ("minference",)
_ORIGINALS = {}


def _patch(cls, forwards, method):
    if method == "fullkv":
        if cls in _ORIGINALS:
            cls.forward = _ORIGINALS[cls]
        return
    if method in _OUT_OF_SCOPE:
        raise NotImplementedError(f"method {method!r} is outside this build's scope (SURVEY.md §2 rows 5-11)")
    if method not in _IN_SCOPE:
        return                       # reference: unknown strings silently patch nothing (monkeypatch.py:19-87)
    if method not in forwards:
        raise NotImplementedError(f"the reference has no {method!r} forward for {cls.__name__} (monkeypatch.py:80-83)")
    _ORIGINALS.setdefault(cls, cls.forward)
    cls.forward = forwards[method]


def replace_llama(method, model_name=None):
    from . import llama_model
    _patch(transformers.models.llama.modeling_llama.LlamaAttention, llama_model.FORWARDS, method)


def replace_mistral(method):
    from . import mistral_model
    _patch(transformers.models.mistral.modeling_mistral.MistralAttention, mistral_model.FORWARDS, method)
