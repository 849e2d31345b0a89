"""Phase breakdown of h2o_fused_kernel from the -DKVC_STAMPS diagnostic build (tools/build_diag.sh stamps).
Run with KVC_LIB_PATH=tools/diag/libkvc_hip_stamps.so.  Shares only — never quote this build's run time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kvcache_factory_amd import _kvc, synth
dev = torch.device("cuda:0")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
q, k, v = synth.make_qkv(32, 8, L, 128, torch.bfloat16, 0, device=dev)
q = torch.randn(1, 32, L, 128, device=dev).to(torch.bfloat16) if q.shape[2] != L else q
p = _kvc.make_params(_kvc.H2O, q, k, None, 8, 0, 7, None)
nbytes = _kvc.lib().kvc_workspace_bytes(ctypes.byref(p)); ws = _kvc.workspace(dev, nbytes)
sc = torch.zeros(1 << 20, dtype=torch.int64, device=dev)     # stamps land here
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for it in range(3):
    sc.zero_()
    rc = _kvc.lib().kvc_scores(ctypes.byref(p), _kvc._ptr(q), _kvc._ptr(k), _kvc._ptr(sc), _kvc._ptr(ws), nbytes, st)
    assert rc == 0, _kvc.lib().kvc_last_error()
    torch.cuda.synchronize()
nwg = 32 * ((L + 255) // 256)
t = sc[: nwg * 16].view(nwg, 2, 8).cpu().double()
names = ["Q -> LDS + barrier", "A: logits (MFMA)", "B: row maxima", "C: exps + chains", "D: p, column sums"]
for w, nm in ((0, "wave 0 (chain wave)"), (1, "wave 5 (worker)")):
    d = t[:, w, :5]
    full = d.sum(1) > 0.9 * d.sum(1).max()
    print(nm, " ticks of the 100 MHz clock per workgroup (16 sub-blocks), full workgroups only:", int(full.sum()))
    tot = d[full].sum(1).mean()
    for i, n_ in enumerate(names):
        print(f"  {n_:28s} mean {d[full][:, i].mean():10.1f}  share {d[full][:, i].mean() / tot:6.3f}")
    print(f"  total {tot:10.1f} ticks = {tot / 100:8.1f} us per workgroup")
