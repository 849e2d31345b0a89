"""C3 (H2O 8k, 32 heads) residual: which head(s) differ from the reference's golden indices, where, and the GPU's pooled scores
of those heads (bits) -> gpurun_out/c3_residual.npz.  The reference's scores of the same heads come from oracle/gen_golden.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import golden_util as G
from kvcache_factory_amd import _kvc as kvc
dev = torch.device("cuda:0")
name = "C3_h2o_8k"
m, arr = G.MANIFEST[name], G.arrays(name)
q, k, v = G.inputs(m, device=dev, expanded=False)
ref = torch.from_numpy(arr["indices"])
out = {}
for dm in ("exact", "mfma16"):
    ko, vo, idx, sc = kvc.compress(kvc.H2O, q, k, v, m["W"], m["n_keep"], m["kernel"], None, "torch_cpu", return_indices=True, return_scores=True, dot_mode=dm)
    idx = idx[0].cpu(); sc = sc[0].cpu()
    bad = [h for h in range(ref.shape[0]) if not torch.equal(idx[h], ref[h])]
    print(dm, "heads differing from the reference:", bad)
    for h in bad:
        a, b = set(idx[h].tolist()), set(ref[h].tolist())
        print(f"  head {h}: GPU-only indices {sorted(a - b)}, reference-only indices {sorted(b - a)}; first differing rank {int((idx[h] != ref[h]).nonzero()[0])}")
        s = sc[h].view(torch.int16)
        for i in sorted((a - b) | (b - a)):
            print(f"    index {i}: GPU score bits 0x{int(s[i]) & 0xffff:04x} = {float(sc[h][i]):.6g}")
        kth = sorted(sc[h].float().tolist(), reverse=True)[m['n_keep'] - 1]
        print(f"    GPU k-th value {kth:.6g}; candidates equal to it: {int((sc[h].float() == kth).sum())}")
    if dm == "exact":
        out["gpu_scores_bits"] = sc.view(torch.int16).numpy(); out["gpu_indices"] = idx.numpy(); out["bad_heads"] = np.array(bad)
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/c3_residual.npz", **out)
