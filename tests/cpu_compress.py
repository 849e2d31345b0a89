"""TEST-ONLY stand-in for kvcache_factory_amd._kvc.compress backed by the CPU oracle, so the host-side plumbing
(cluster classes, patched attention forward, cache layer) can be exercised without a GPU.  Never shipped."""
import torch

from kvcache_factory_amd import _kvc
from oracle import kvc_oracle as O


def oracle_compress(method, q, k, v, window, n_keep, kernel_size=5, pooling="avgpool", tie_mode="torch_cpu",
                    n_q_heads=None, return_indices=False, return_scores=False):
    assert k.shape[0] == 1
    k, v = k.contiguous(), v.contiguous()
    if method == _kvc.STREAMINGLLM:
        hq = n_q_heads if n_q_heads is not None else k.shape[1]
        ko, vo, idx = O.streaming(k, v, window, n_keep, hq)
        sc = None
    else:
        tm = O.TIES_TORCH if tie_mode in ("torch_cpu", 0) else O.TIES_CANON
        ko, vo, idx, sc = O.compress(q, k, v, window, n_keep, kernel_size, pooling if method != _kvc.H2O else "avgpool",
                                     full_rows=method == _kvc.H2O, dot_mode=O.DOT_CHAIN,
                                     sum_mode=O.SUM_TORCH16, tie_mode=tm)
    out = [ko, vo]
    if return_indices:
        out.append(idx[None])
    if return_scores:
        out.append(sc[None] if sc is not None else None)
    return tuple(out)


def oracle_compress_batch(method, qs, ks, vs, window, keeps, kernel_size=5, pooling="avgpool", tie_mode="torch_cpu",
                          q_rows="all", n_q_heads=None, return_indices=False, dot_mode=None, outs=None, reuse=False):
    """Stand-in for _kvc.compress_batch: the items one after the other through oracle_compress."""
    ko, vo, ix = [], [], []
    for i, (k, v) in enumerate(zip(ks, vs)):
        q = qs[i] if qs is not None else None
        if q is not None and q_rows == "window":                     # rebuild a full-length q: only the last W rows score
            assert q.shape[2] == window
            full = torch.zeros(q.shape[0], q.shape[1], k.shape[2], q.shape[3], dtype=q.dtype)
            full[:, :, -window:] = q
            q = full
        r = oracle_compress(method, q, k, v, window, keeps[i], kernel_size, pooling, tie_mode, n_q_heads, True)
        if outs is not None:                                          # caller's buffers: fill the first rows, hand out views
            kb, vb = outs[i]
            rows = r[0].shape[2]
            kb[:, :, :rows].copy_(r[0]); vb[:, :, :rows].copy_(r[1])
            r = (kb[:, :, :rows], vb[:, :, :rows], r[2])
        ko.append(r[0]); vo.append(r[1]); ix.append(r[2])
    return (ko, vo, ix) if return_indices else (ko, vo)


def oracle_l2norm_compress(k, v, rows, n_q_heads=None, return_indices=False, return_norms=False, out=None):
    """Stand-in for _kvc.l2norm_compress (host-logic tests only)."""
    assert k.shape[0] == 1 and out is None
    hq = n_q_heads if n_q_heads is not None else k.shape[1]
    ko, vo, idx, norms = O.l2norm(k.contiguous(), v.contiguous(), rows, hq)
    res = [ko, vo]
    if return_indices:
        res.append(idx[None])
    if return_norms:
        res.append(norms[None])
    return tuple(res)


def oracle_compress_merge(method, q, k, v, window, n_keep, kernel_size=5, pooling="avgpool", tie_mode="torch_cpu", n_q_heads=None):
    """Stand-in for _kvc.compress_merge (host-logic tests only)."""
    assert k.shape[0] == 1
    k, v = k.contiguous(), v.contiguous()
    if method == _kvc.STREAMINGLLM:
        hq = n_q_heads if n_q_heads is not None else k.shape[1]
        idx = torch.arange(n_keep, dtype=torch.int64).expand(hq, n_keep).contiguous()
    else:
        sc = O.scores(q, k, window, kernel_size, pooling if method != _kvc.H2O else "avgpool", full_rows=method == _kvc.H2O,
                      dot_mode=O.DOT_CHAIN, sum_mode=O.SUM_TORCH16)
        idx, _ = O.topk(sc, n_keep, O.TIES_TORCH if tie_mode in ("torch_cpu", 0) else O.TIES_CANON)
    ko, vo, _ = O.merge_pivot(k, v, idx, window)
    return ko, vo


def oracle_think_prune(kc, q, recent_size, ratio, return_scores=False):
    """Stand-in for _kvc.think_prune (host-logic tests only)."""
    pruned, recent, keep, sc = O.think_prune(kc, q.contiguous(), recent_size, ratio)
    return (pruned, recent, keep, sc[None]) if return_scores else (pruned, recent, keep)


def oracle_decode_step(q, k_new, v_new, k_prefix, v_prefix, prefix_rows, k_tail, v_tail, tail_rows, scaling, k_pruned=None, chan=None,
                       pruned_rows=0):
    """Stand-in for _kvc.decode_step (host-logic tests only): in-place tail append, then fp32 softmax attention over
    [pruned rows (kept channels) | prefix rows | tail rows], token i of the step seeing tail rows < tail_rows + i + 1."""
    b, hq, t, D = q.shape
    g = hq // k_new.shape[1]
    k_tail[:, :, tail_rows:tail_rows + t] = k_new
    v_tail[:, :, tail_rows:tail_rows + t] = v_new
    outs = []
    for i in range(t):
        T = tail_rows + i + 1
        kt, vt = k_tail[:, :, :T].repeat_interleave(g, 1).float(), v_tail[:, :, :T].repeat_interleave(g, 1).float()
        qi = q[:, :, i:i + 1].float()
        parts = []
        if pruned_rows:
            qa = torch.gather(qi[:, :, 0], -1, chan.long().clamp(min=0)) * (chan >= 0)
            parts.append(torch.einsum("bhd,bhrd->bhr", qa, k_pruned.float())[:, :, None])
        parts.append(qi @ k_prefix[:, :, :prefix_rows].float().transpose(2, 3))
        parts.append(qi @ kt.transpose(2, 3))
        w = torch.softmax(torch.cat(parts, -1) * scaling, -1)
        outs.append(w @ torch.cat([v_prefix[:, :, :pruned_rows + prefix_rows].float(), vt], 2))
    return torch.cat(outs, 2).transpose(1, 2).to(q.dtype)


def oracle_cam_plan(q, k, window, start_budget):
    """Stand-ins for _kvc.cam_plan / cam_merge / select / gather (host-logic tests of CAMKVCluster only)."""
    sc, prob = O.cam_plan(q.contiguous(), k.contiguous(), window, start_budget, dot_mode=O.DOT_CHAIN, sum_mode=O.SUM_TORCH16)
    return sc[None], prob[None]


def oracle_cam_merge(v, merge_mask, window, start_budget):
    v.copy_(O.cam_merge(v, merge_mask[0], window, start_budget))
    return v


def oracle_select(scores_t, n_keep, tie_mode="torch_cpu"):
    idx, _ = O.topk(scores_t[0].contiguous(), n_keep, O.TIES_TORCH if tie_mode in ("torch_cpu", 0) else O.TIES_CANON)
    return idx[None]


def oracle_gather(src, idx, window, n_q_heads):
    return O.gather(src.contiguous(), idx[0], window, n_q_heads)
