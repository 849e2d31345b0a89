import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import golden_util as G
from kvcache_factory_amd import _kvc
from oracle import kvc_oracle as O
dev = torch.device('cuda:0')
m = G.MANIFEST['C2_snapkv_8k_fp16']
qd, kd, vd = G.inputs(m, device=dev, expanded=False)
q, k = qd.cpu(), kd.cpu()
sc_o, lg_o, pr_o, ws_o = O.scores(q, k, m['W'], m['kernel'], m['pooling'], want_intermediates=True, dot_mode=O.DOT_CHAIN, sum_mode=O.SUM_KVC)
sc_g, lg_g, rmax, rsum = _kvc.scores(_kvc.SNAPKV, qd, kd, m['W'], m['kernel'], m['pooling'], want_intermediates=True)
lg_g = lg_g[0].cpu().permute(0, 2, 1).contiguous()
dl = (G.bits(lg_g) != G.bits(lg_o))
print('logit mismatches', int(dl.sum()), 'of', dl.numel())
ds = (G.bits(sc_g[0].cpu()) != G.bits(sc_o))
print('score mismatches', int(ds.sum()), 'of', ds.numel())
nz = ds.nonzero()[:10]
for h, j in nz.tolist():
    print(' h', h, 'j', j, 'gpu', float(sc_g[0, h, j]), hex(int(G.bits(sc_g[0].cpu())[h, j]) & 0xffff), 'oracle', float(sc_o[h, j]), hex(int(G.bits(sc_o)[h, j]) & 0xffff), 'wsum oracle', float(ws_o[h, j]))
# no-pooling scores to see raw window sums
sc_g2 = _kvc.scores(_kvc.SNAPKV, qd, kd, m['W'], 1, 'maxpool')
d2 = (G.bits(sc_g2[0].cpu()) != G.bits(ws_o))
print('wsum mismatches', int(d2.sum()))
for h, j in d2.nonzero()[:10].tolist():
    p = pr_o[h, :, j].float()
    print(' h', h, 'j', j, 'gpu', hex(int(G.bits(sc_g2[0].cpu())[h, j]) & 0xffff), 'oracle', hex(int(G.bits(ws_o)[h, j]) & 0xffff), 'probs', [hex(int(x) & 0xffff) for x in G.bits(pr_o[h, :, j])], 'sum f32', float(p.sum()))
# rowmax / rowsum
print('rowmax', rmax[0, 0].tolist()); print('rowsum', rsum[0, 0].tolist())
