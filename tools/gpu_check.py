"""Ad-hoc GPU bring-up check (not a test): GPU stages vs the CPU oracle, plus a first timing."""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kvcache_factory_amd import _kvc
from oracle import kvc_oracle as O

dev = torch.device("cuda:0")
print("device", torch.cuda.get_device_name(0), "lib", _kvc.lib().kvc_version(), flush=True)

def bits(t):
    return t.contiguous().view(torch.int32 if t.dtype == torch.float32 else torch.int16)

def case(dtype, Hq, Hkv, L, D, W, ks, pooling, k, seed=0, expanded=False):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(1, L, Hq, D, generator=g).to(dtype).transpose(1, 2)      # strides like the model's q
    kk = torch.randn(1, Hkv, L, D, generator=g).to(dtype)
    vv = torch.randn(1, Hkv, L, D, generator=g).to(dtype)
    if expanded:
        kk = kk.repeat_interleave(Hq // Hkv, dim=1).contiguous(); vv = vv.repeat_interleave(Hq // Hkv, dim=1).contiguous()
    tag = f"{str(dtype)[6:]:9s} Hq{Hq} Hkv{kk.shape[1]} L{L} D{D} W{W} {pooling}{ks} k{k}"
    try:
        sc_o, lg_o, pr_o, ws_o = O.scores(q, kk, W, ks, pooling, want_intermediates=True, dot_mode=O.DOT_CHAIN, sum_mode=O.SUM_KVC)
        qd, kd, vd = q.to(dev), kk.to(dev), vv.to(dev)
        sc_g, lg_g, rmax, rsum = _kvc.scores(_kvc.SNAPKV, qd, kd, W, ks, pooling, want_intermediates=True)
        torch.cuda.synchronize()
        lg_g = lg_g[0].cpu().permute(0, 2, 1)       # [H][W][L]
        m_log = (bits(lg_g) != bits(lg_o)).sum().item()
        m_sc = (bits(sc_g[0].cpu()) != bits(sc_o)).sum().item()
        # select (canonical) on the GPU's own scores vs oracle canonical on the same scores
        idx_g = _kvc.select(sc_g, k, "canonical")[0].cpu()
        idx_o, _ = O.topk(sc_g[0].cpu(), k, O.TIES_CANON)
        m_idx = (idx_g != idx_o).sum().item()
        # full compress
        ko, vo, idx2, sc2 = _kvc.compress(_kvc.SNAPKV, qd, kd, vd, W, k, ks, pooling, "canonical", return_indices=True, return_scores=True)
        torch.cuda.synchronize()
        ko_o = O.gather(kk, idx2[0].cpu(), W, Hq); vo_o = O.gather(vv, idx2[0].cpu(), W, Hq)
        m_k = (bits(ko.cpu()) != bits(ko_o)).sum().item(); m_v = (bits(vo.cpu()) != bits(vo_o)).sum().item()
        m_idx2 = (idx2[0].cpu() != idx_o).sum().item()
        ok = (m_log == 0 and m_sc == 0 and m_idx == 0 and m_k == 0 and m_v == 0 and m_idx2 == 0)
        print(f"{'OK ' if ok else 'BAD'} {tag}: logits {m_log}/{lg_o.numel()} scores {m_sc}/{sc_o.numel()} idx {m_idx} idx(compress) {m_idx2} K {m_k} V {m_v}", flush=True)
        return ok
    except Exception as e:
        print(f"EXC {tag}: {type(e).__name__}: {e}", flush=True)
        traceback.print_exc()
        return False

allok = True
for dtype in (torch.bfloat16, torch.float16, torch.float32):
    allok &= case(dtype, 4, 4, 96, 128, 8, 5, "avgpool", 24)
    allok &= case(dtype, 4, 2, 257, 64, 32, 5, "avgpool", 64)
    allok &= case(dtype, 4, 1, 1024, 128, 8, 7, "maxpool", 120)
    allok &= case(dtype, 8, 2, 1000, 128, 16, 7, "maxpool", 40)
    allok &= case(dtype, 4, 4, 300, 64, 5, 3, "maxpool", 17)
    allok &= case(dtype, 2, 1, 640, 128, 64, 5, "avgpool", 100)
allok &= case(torch.bfloat16, 32, 8, 8000, 128, 8, 7, "maxpool", 120)
allok &= case(torch.bfloat16, 32, 8, 8000, 128, 8, 7, "maxpool", 120, expanded=True)
allok &= case(torch.float16, 32, 8, 8000, 128, 32, 7, "maxpool", 96)
print("ALL OK" if allok else "SOME BAD", flush=True)

# first timing at C2 (GQA-native and expanded), canonical ties
for expanded in (False, True):
    Hq, Hkv, L, D, W, k = 32, 8, 8000, 128, 8, 120
    q = torch.randn(1, L, Hq, D, device=dev).to(torch.bfloat16).transpose(1, 2)
    kk = torch.randn(1, Hkv, L, D, device=dev).to(torch.bfloat16); vv = torch.randn(1, Hkv, L, D, device=dev).to(torch.bfloat16)
    if expanded:
        kk = kk.repeat_interleave(4, dim=1).contiguous(); vv = vv.repeat_interleave(4, dim=1).contiguous()
    for _ in range(5): _kvc.compress(_kvc.SNAPKV, q, kk, vv, W, k, 7, "maxpool", "canonical")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 200
    for _ in range(N): _kvc.compress(_kvc.SNAPKV, q, kk, vv, W, k, 7, "maxpool", "canonical")
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
    print(f"C2 {'expanded' if expanded else 'gqa-native'}: {dt*1e6:.1f} us/layer-call  {L/dt/1e6:.1f} M tokens/s", flush=True)
