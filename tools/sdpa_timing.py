"""Time of the model's own prefill attention at the bench shapes (torch SDPA, bf16, causal, 32 query heads over 8 KV heads,
D = 128) — the yardstick for SURVEY 8f N2."""
import time, torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
for L in (8000, 32000):
    q = torch.randn(1, 32, L, 128, dtype=torch.bfloat16, device=dev)
    k = torch.randn(1, 8, L, 128, dtype=torch.bfloat16, device=dev)
    v = torch.randn(1, 8, L, 128, dtype=torch.bfloat16, device=dev)
    kk, vv = k.repeat_interleave(4, dim=1), v.repeat_interleave(4, dim=1)
    for _ in range(3):
        F.scaled_dot_product_attention(q, kk, vv, is_causal=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        F.scaled_dot_product_attention(q, kk, vv, is_causal=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    flops = 2 * 2 * 32 * L * L * 128 / 2
    print(f"SDPA causal bf16 L={L}: {ms:.3f} ms per layer ({flops / ms / 1e9:.0f} TFLOP/s)")
