"""Helpers shared by the parity tests: load tests/golden fixtures (generated from the imported reference by
oracle/gen_golden.py) and regenerate their inputs with the portable generator."""
import hashlib
import json
import os

import numpy as np
import torch

from kvcache_factory_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    MANIFEST = {k: v for k, v in json.load(_f).items() if not k.startswith("_")}


def names(pred=lambda m: True):
    return sorted(n for n, m in MANIFEST.items() if pred(m))


_INPUT_CACHE = {}


def _make_qkv_cached(Hq, Hkv, L, D, dtype, seed, peaky, expanded, device):
    """The big fixtures (8k x 32 heads, 32k) share their inputs (same seed and geometry): generate each once per process.
    Tests treat the tensors as read-only."""
    key = (Hq, Hkv, L, D, dtype, seed, peaky, expanded, str(device))
    if key not in _INPUT_CACHE:
        if L * Hq < 64 * 1024:                               # small cases are cheap: no point in holding them
            return synth.make_qkv(Hq, Hkv, L, D, dtype, seed, peaky=peaky, expanded=expanded, device=device)
        if len(_INPUT_CACHE) >= 4:
            _INPUT_CACHE.pop(next(iter(_INPUT_CACHE)))
        _INPUT_CACHE[key] = synth.make_qkv(Hq, Hkv, L, D, dtype, seed, peaky=peaky, expanded=expanded, device=device)
    return _INPUT_CACHE[key]


def inputs(meta, device="cpu", expanded=True):
    """Regenerate (q, k, v) of a fixture exactly as oracle/gen_golden.py fed them to the reference."""
    dtype = DT[meta["dtype"]]
    q, k, v = _make_qkv_cached(meta["Hq"], meta["Hkv"], meta["L"], meta["D"], dtype, meta["seed"],
                               meta.get("peaky", False), expanded, device)
    if meta.get("zero_q"):
        q = torch.zeros_like(q)
    if meta.get("scale_q"):
        q = (q.float() * meta["scale_q"]).to(dtype)
    return q, k, v


def arrays(name):
    p = os.path.join(GOLDEN, name + ".npz")
    return dict(np.load(p)) if os.path.exists(p) else {}


def from_bits(a, dtype):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.view(dtype)


def bits(t):
    t = t.detach().cpu().contiguous()
    return t.view(torch.int32 if t.dtype == torch.float32 else torch.int16)


def sha(t):
    return hashlib.sha256(bits(t).numpy().tobytes()).hexdigest()


def ulp_diff(a, b):
    """Elementwise distance in units of the dtype's last place (monotone integer mapping of the bit patterns)."""
    def key(t):
        i = bits(t).to(torch.int64)
        nb = 32 if t.dtype == torch.float32 else 16
        sign = 1 << (nb - 1)
        i = i & ((1 << nb) - 1)
        return torch.where(i >= sign, sign - 1 - (i - sign), i + sign - 1 + 1)
    return (key(a) - key(b)).abs()


def pool_name(meta):
    return None if meta["method"] == "h2o" else meta["pooling"]


def product_modes(oracle, meta):
    """The oracle arithmetic that mirrors the GPU product: fmaf-chain dot products (what the gfx950 f32-input MFMA
    computes) and softmax row sums in torch's own 16-lane order (kvc_score.hip torch16_rowsums, kvc_h2o.hip)."""
    return dict(dot_mode=oracle.DOT_CHAIN, sum_mode=oracle.SUM_TORCH16)


KILLERS = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("killer_") and f.endswith(".npz"))


def killer(name, dtype):
    """An adversarial score row (oracle/killer_adversary.h, tests/golden/killer_*.npz) in `dtype`, its budget k and the indices
    torch-CPU topk returned for it when the fixture was made.  Codes become fp32 integers, or consecutive positive normal
    bf16 / fp16 bit patterns: the order and the ties of the codes carry over exactly."""
    a = arrays(name)
    codes = torch.from_numpy(a["codes"].astype(np.int64))
    if dtype == torch.float32:
        sc = codes.float()
    else:
        assert int(codes.max()) + 0x0480 < 0x7800, "row too long for consecutive 16-bit codes"
        sc = (codes + 0x0480).to(torch.int16).view(dtype)
    return sc[None].contiguous(), int(a["k"]), torch.from_numpy(a["want"])[None]
