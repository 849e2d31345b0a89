"""Portable synthetic inputs for parity tests and bench.py.

A counter-based generator built only from 64-bit integer arithmetic and IEEE float64 additions, so
the same (seed, shape) gives bit-identical tensors on every machine — unlike torch.manual_seed
streams.  Values are Irwin-Hall(12) - 6 (mean 0, variance 1), then cast to the target dtype.
"""
import torch

_C1 = 0x9E3779B97F4A7C15 - (1 << 64)      # the splitmix64 constants as signed int64
_C2 = 0xBF58476D1CE4E5B9 - (1 << 64)
_C3 = 0x94D049BB133111EB - (1 << 64)


def _lsr(x, s):
    """Logical right shift of an int64 tensor (torch's >> is arithmetic)."""
    return (x >> s) & ((1 << (64 - s)) - 1)


def _splitmix64(x):
    z = x + _C1                     # int64 arithmetic wraps modulo 2^64 on CPU and GPU alike
    z = (z ^ _lsr(z, 30)) * _C2
    z = (z ^ _lsr(z, 27)) * _C3
    return z ^ _lsr(z, 31)


def _wrap64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def normal(shape, seed, stream=0, device="cpu", chunk=1 << 22):
    """float32 tensor of `shape`; element i depends only on (seed, stream, i), on any device."""
    n = 1
    for d in shape:
        n *= int(d)
    out = torch.empty(n, dtype=torch.float32, device=device)
    base = _splitmix64(torch.tensor([_wrap64(seed * 1000003 + stream * 7919 + 1)], dtype=torch.int64, device=device))
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        ctr = torch.arange(s, e, dtype=torch.int64, device=device) * 12 + base
        acc = torch.zeros(e - s, dtype=torch.float64, device=device)
        for j in range(12):
            h = _splitmix64(ctr + j)
            acc += _lsr(h, 11).to(torch.float64) * (1.0 / 9007199254740992.0)
        out[s:e] = (acc - 6.0).to(torch.float32)
    return out.view(*shape)


def make_qkv(n_q_heads, n_kv_heads, q_len, head_dim, dtype, seed, peaky=False, expanded=False, bsz=1, device="cpu"):
    """Q with the model's memory order [b, L, Hq, D] viewed as [b, Hq, L, D] (strides (L*Hq*D, D, Hq*D, 1),
    llama_model.py:244), K/V [b, Hkv, L, D] contiguous — or repeat_kv-expanded to Hq heads
    (llama_model.py:277-278) when expanded=True.
    peaky=True plants 64 heavy-hitter keys per KV head (K row = 4 * mean window query)."""
    q = normal((bsz, q_len, n_q_heads, head_dim), seed, 0, device).to(dtype).transpose(1, 2)
    k = normal((bsz, n_kv_heads, q_len, head_dim), seed, 1, device)
    v = normal((bsz, n_kv_heads, q_len, head_dim), seed, 2, device).to(dtype)
    if peaky:
        g = n_q_heads // n_kv_heads
        pos_src = normal((n_kv_heads, 64), seed, 3, device)
        for b in range(bsz):
            for h in range(n_kv_heads):
                pos = ((pos_src[h].double() + 6.0) / 12.0 * (q_len - 64)).long().clamp(0, q_len - 65).unique()
                qbar = q[b, h * g, -8:, :].float().mean(0)
                k[b, h, pos, :] = 4.0 * qbar
    k = k.to(dtype)
    if expanded:
        g = n_q_heads // n_kv_heads
        k = k[:, :, None].expand(bsz, n_kv_heads, g, q_len, head_dim).reshape(bsz, n_q_heads, q_len, head_dim)
        v = v[:, :, None].expand(bsz, n_kv_heads, g, q_len, head_dim).reshape(bsz, n_q_heads, q_len, head_dim)
    return q, k, v
