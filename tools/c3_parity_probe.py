"""C3 (H2O 8k, 32 heads) on the GPU against the reference's golden indices: how many heads / positions agree."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import golden_util as G
from kvcache_factory_amd import _kvc as kvc
dev = torch.device("cuda:0")
for name in ("C3_h2o_8k_2heads", "C3_h2o_8k"):
    m, arr = G.MANIFEST[name], G.arrays(name)
    q, k, v = G.inputs(m, device=dev, expanded=False)
    ref = torch.from_numpy(arr["indices"])
    for dm in ("exact", "mfma16"):
        ko, vo, idx, sc = kvc.compress(kvc.H2O, q, k, v, m["W"], m["n_keep"], m["kernel"], None, "torch_cpu", return_indices=True,
                                       return_scores=True, dot_mode=dm)
        idx = idx[0].cpu()
        heads_eq = int((idx == ref).all(-1).sum())
        sets_eq = sum(set(a.tolist()) == set(b.tolist()) for a, b in zip(idx, ref))
        overlap = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(idx, ref)) / ref.numel()
        refv = torch.from_numpy(arr["values"]).view(torch.bfloat16).float()
        gotv = torch.gather(sc[0].cpu().float(), 1, idx)
        print(f"{name} dot={dm}: heads identical (order too) {heads_eq}/{ref.shape[0]}, identical sets {sets_eq}, index overlap {overlap:.4f}, "
              f"selected values equal {float((gotv == refv).float().mean()):.4f}", flush=True)
