"""Phase breakdown of logits_kernel from the -DKVC_STAMPS diagnostic build (tools/diag/libkvc_hip_stamps.so).
Run with KVC_LIB_PATH pointing at that library.  Shares only — never quote this build's run time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kvcache_factory_amd import _kvc, synth
dev = torch.device("cuda:0")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
q, k, v = synth.make_qkv(32, 8, 8000, 128, torch.bfloat16, 0, device=dev)
p = _kvc.make_params(_kvc.SNAPKV, q, k, None, W, 0, 7, "maxpool")
p.debug_stage_mask = 1
nbytes = _kvc.lib().kvc_workspace_bytes(ctypes.byref(p)); ws = _kvc.workspace(dev, nbytes)
sc = torch.zeros(1 << 20, dtype=torch.int64, device=dev)     # stamps land here
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for it in range(5):
    sc.zero_()
    rc = _kvc.lib().kvc_scores(ctypes.byref(p), _kvc._ptr(q), _kvc._ptr(k), _kvc._ptr(sc), _kvc._ptr(ws), nbytes, st)
    assert rc == 0, _kvc.lib().kvc_last_error()
    torch.cuda.synchronize()
nblk = 63 * 8
t = sc[: nblk * 4 * 8].view(nblk * 4, 8).cpu().double()
d = t[:, 1:6] - t[:, 0:5]
names = ["issue Q loads + stage K tile + barrier", "convert Q (wait Q data)", "64 x mfma_f32_32x32x2 (+LDS reads)", "epilogue: round/div/mask + stores", "row-max reduce-scatter"]
print("waves", t.shape[0], " (s_memtime ticks @100MHz? -> cycles of the constant clock)")
for i, nm in enumerate(names):
    print(f"  {nm:45s} mean {d[:, i].mean():9.1f}  p50 {d[:, i].median():9.1f}  max {d[:, i].max():9.1f}")
span = (t[:, 5].max() - t[:, 0].min())
print("  wave total mean", float((t[:, 5] - t[:, 0]).mean()), " kernel span", float(span))
# start-time spread of blocks
s0 = t[:, 0] - t[:, 0].min()
print("  wave start offsets: p50", float(s0.median()), "p90", float(s0.quantile(0.9)), "max", float(s0.max()))

# ---- select_kernel phases (wave 0 of each head) ----
p2 = _kvc.make_params(_kvc.SNAPKV, q, k, v, W, 120, 7, "maxpool", "canonical")
ko = torch.empty(1, 32, 128, 128, dtype=torch.bfloat16, device=dev); vo = torch.empty_like(ko)
idx = torch.zeros(1, 32, 120, dtype=torch.int64, device=dev)
nb = _kvc.lib().kvc_workspace_bytes(ctypes.byref(p2)); ws2 = _kvc.workspace(dev, nb)
for it in range(3):
    rc = _kvc.lib().kvc_compress(ctypes.byref(p2), _kvc._ptr(q), _kvc._ptr(k), _kvc._ptr(v), _kvc._ptr(ko), _kvc._ptr(vo), _kvc._ptr(idx), None, _kvc._ptr(ws2), nb, st)
    assert rc == 0, _kvc.lib().kvc_last_error()
    torch.cuda.synchronize()
ts = idx[0, :, :6].cpu().double()
dd = ts[:, 1:] - ts[:, :-1]
for i, nm in enumerate(["load keys", "threshold bit search (16 rounds)", "membership scans + list", "order (rank sort)", "idx store + fused gather"]):
    print(f"  select: {nm:36s} mean {dd[:, i].mean():9.1f} max {dd[:, i].max():9.1f}")
print("  select total", float((ts[:, 5] - ts[:, 0]).mean()))
