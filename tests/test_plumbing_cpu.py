"""Host-side plumbing (A0, A12-A15) without a GPU: cluster classes, init_* factories, replace_llama/replace_mistral,
the patched attention forward and the true-length cache layer.  The HIP call is replaced by the CPU oracle through
tests/cpu_compress.py (test-only injection; the product itself has no CPU path).

Includes BASELINE.json config C1: StreamingLLM on a Llama-3-8B-shaped 1-layer model, seq_len=1000, budget=64,
window = cap-4 = 60 (run_longbench.py:220-223), sdpa on CPU."""
import pytest
import torch

import golden_util as G
from cpu_compress import (oracle_cam_merge, oracle_cam_plan, oracle_compress, oracle_compress_batch, oracle_compress_merge, oracle_decode_step,
                          oracle_gather, oracle_l2norm_compress, oracle_select, oracle_think_prune)
from kvcache_factory_amd import _kvc, monkeypatch as mp, pyramidkv_utils as pu


@pytest.fixture()
def cpu_backend(monkeypatch, oracle):
    monkeypatch.setattr(_kvc, "compress", oracle_compress)
    monkeypatch.setattr(_kvc, "compress_batch", oracle_compress_batch)
    monkeypatch.setattr(_kvc, "l2norm_compress", oracle_l2norm_compress)
    monkeypatch.setattr(_kvc, "compress_merge", oracle_compress_merge)
    monkeypatch.setattr(_kvc, "think_prune", oracle_think_prune)
    monkeypatch.setattr(_kvc, "decode_step", oracle_decode_step)
    for name, fn in (("cam_plan", oracle_cam_plan), ("cam_merge", oracle_cam_merge), ("select", oracle_select), ("gather", oracle_gather)):
        monkeypatch.setattr(_kvc, name, fn)
    monkeypatch.setattr(pu, "BATCH_LAYERS", False)       # these tests watch one update_kv per layer
    yield


def _llama(layers=1, dtype=torch.float32):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=512, num_hidden_layers=layers, num_attention_heads=32,
                      num_key_value_heads=8, head_dim=128, vocab_size=512, max_position_embeddings=8192,
                      attn_implementation="sdpa")
    torch.manual_seed(0)
    return LlamaForCausalLM(cfg).to(dtype).eval()


def _set_knobs(model, **kw):
    for layer in model.model.layers:                     # run_longbench.py:253-261
        for k, v in kw.items():
            setattr(layer.self_attn.config, k, v)


def test_clusters_passthrough_returns_same_objects(cpu_backend):
    """q_len < cap: the very same tensor objects come back (pyramidkv_utils.py:314-315, :218, :541, :603)."""
    q, k, v = (torch.randn(1, 4, 30, 64) for _ in range(3))
    for c in (pu.SnapKVCluster(8, 40), pu.H2OKVCluster(8, 40), pu.StreamingLLMKVCluster(8, 40),
              pu.PyramidKVCluster(num_hidden_layers=4, window_size=8, max_capacity_prompt=40, layer_idx=1)):
        ko, vo = c.update_kv(k, q, v, None, 1)
        assert ko is k and vo is v


def test_cluster_errors_match_reference(cpu_backend):
    with pytest.raises(AssertionError):
        pu.SnapKVCluster(window_size=64, max_capacity_prompt=64)             # :289
    q, k, v = (torch.randn(1, 2, 64, 64) for _ in range(3))
    with pytest.raises(AssertionError):
        pu.SnapKVCluster(8, 40).update_kv(k[:, :, :50], q, v, None, 1)       # :309
    with pytest.raises(ValueError, match="Pooling method not supported"):
        pu.SnapKVCluster(8, 40, pooling="median").update_kv(k, q, v, None, 1)   # :333
    with pytest.raises(ValueError, match="Merge method not supported"):
        pu.SnapKVCluster(8, 40, merge="bogus").update_kv(k, q, v, None, 1)      # :164
    with pytest.raises(RuntimeError, match="head_dim 128"):                  # merge_kv hard-codes 128 (:149); these tensors have 64
        pu.SnapKVCluster(8, 40, merge="pivot").update_kv(k, q, v, None, 1)
    # H2O ignores pooling entirely (:555-561)
    ko, vo = pu.H2OKVCluster(8, 40, pooling="median").update_kv(k, q, v, None, 1)
    assert ko.shape == (1, 2, 40, 64)


@pytest.mark.parametrize("name", ["snap_fp32_maxpool_W8_L257_D64", "pyr_bf16_layer15_L1024", "h2o_fp16_W8_L257",
                                  "stream_bf16_L1000", "edge_qlen_eq_cap", "edge_pyr_middle_branch"])
def test_cluster_update_kv_matches_reference_fixture(cpu_backend, name):
    """Cluster API end to end (expanded K/V exactly as the reference's caller passes them, and GQA-native)."""
    m = G.MANIFEST[name]
    q, k, v = G.inputs(m)
    kw = dict(window_size=m["W"], max_capacity_prompt=m["cap"], kernel_size=m["kernel"], pooling=m["pooling"])
    cls = {"snapkv": pu.SnapKVCluster, "h2o": pu.H2OKVCluster, "streamingllm": pu.StreamingLLMKVCluster,
           "pyramidkv": lambda **a: pu.PyramidKVCluster(num_hidden_layers=m["n_layers"], layer_idx=m["layer_idx"], **a)}[m["method"]]
    g = m["Hq"] // m["Hkv"]
    for kk, vv in ((k, v), (k[:, ::g].contiguous(), v[:, ::g].contiguous())):
        ko, vo = cls(**kw).update_kv(kk, q, vv, None, g)
        assert list(ko.shape) == m["out_shape"]
        assert G.sha(ko) == m["k_out_sha256"] and G.sha(vo) == m["v_out_sha256"]


def test_init_factories_rebuild_cluster_every_forward(cpu_backend):
    """init_* default-fill once and rebuild kv_cluster on every call (pyramidkv_utils.py:880-1031)."""
    class Cfg:
        pass

    class Attn:
        def __init__(self):
            self.config, self.layer_idx = Cfg(), 3
    a = Attn()
    pu.init_snapkv(a)
    assert (a.config.window_size, a.config.max_capacity_prompt, a.config.kernel_size, a.config.pooling) == (32, 4096, 5, "avgpool")
    first = a.kv_cluster
    a.config.max_capacity_prompt = 128
    pu.init_snapkv(a)
    assert a.kv_cluster is not first and a.kv_cluster.max_capacity_prompt == 128
    b = Attn()
    pu.init_pyramidkv(b, num_hidden_layers=32)
    assert b.config.max_capacity_prompt == 2048 and b.kv_cluster.layer_idx == 3 and b.kv_cluster.num_hidden_layers == 32
    c = Attn(); pu.init_H2O(c); assert isinstance(c.kv_cluster, pu.H2OKVCluster) and c.config.max_capacity_prompt == 2048
    d = Attn(); pu.init_StreamingLLM(d); assert isinstance(d.kv_cluster, pu.StreamingLLMKVCluster)


def test_replace_llama_and_mistral_rebind_and_restore():
    import transformers
    LA = transformers.models.llama.modeling_llama.LlamaAttention
    MA = transformers.models.mistral.modeling_mistral.MistralAttention
    orig_l, orig_m = LA.forward, MA.forward
    try:
        for method in ("pyramidkv", "snapkv", "h2o", "streamingllm", "l2norm", "cam", "adakv", "headkv"):
            mp.replace_llama(method); mp.replace_mistral(method)
            assert LA.forward.kvc_method == method and MA.forward.kvc_method == method
        mp.replace_llama("no-such-method")               # unknown strings patch nothing (monkeypatch.py:19-87)
        assert LA.forward.kvc_method == "headkv"
        with pytest.raises(NotImplementedError):
            mp.replace_llama("minference")
    finally:
        mp.replace_llama("fullkv"); mp.replace_mistral("fullkv")
    assert LA.forward is orig_l and MA.forward is orig_m


def _generate(model, ids, n):
    with torch.no_grad():
        return model.generate(ids, max_new_tokens=n, do_sample=False, use_cache=True, return_dict_in_generate=True)


def test_c1_streamingllm_llama3_8b_shape_one_layer(cpu_backend):
    """BASELINE config C1 (plumbing): cache holds tokens 0-3 + the last 60 for all 32 heads, the prefill step
    attends over the uncompressed K/V (first new token identical to full-KV), decode continues at the true length."""
    model = _llama(1)
    ids = torch.randint(0, 512, (1, 1000), generator=torch.Generator().manual_seed(1))
    full = _generate(model, ids, 3)
    full_k = full.past_key_values.layers[0].keys            # [1, 8, 1003, 128]
    try:
        mp.replace_llama("streamingllm")
        _set_knobs(model, window_size=60, max_capacity_prompt=64, kernel_size=7, pooling="maxpool")
        seen = {}
        orig = pu.StreamingLLMKVCluster.update_kv

        def spy(self, k, q, v, am, g):
            seen["k"], seen["q_strides"] = k, q.stride()
            return orig(self, k, q, v, am, g)
        pu.StreamingLLMKVCluster.update_kv = spy
        out = _generate(model, ids, 3)
    finally:
        pu.StreamingLLMKVCluster.update_kv = orig
        mp.replace_llama("fullkv")
    layer = out.past_key_values.layers[0]
    assert layer.get_seq_length() == 1002 and layer.keys.shape == (1, 32, 64 + 2, 128)
    assert seen["k"].shape == (1, 8, 1000, 128)                       # GQA-native hand-off
    assert seen["q_strides"][1:] == (128, 4096, 1)                    # q arrives in [L, H, D] memory order
    expect = torch.cat([full_k[:, :, :4], full_k[:, :, 940:1000]], dim=2).repeat_interleave(4, dim=1)
    assert torch.equal(layer.keys[:, :, :64], expect)
    assert out.sequences[0, 1000] == full.sequences[0, 1000]           # prefill logits unaffected by eviction
    assert out.sequences.shape == full.sequences.shape


@pytest.mark.parametrize("method,cls", [("snapkv", "SnapKVCluster"), ("pyramidkv", "PyramidKVCluster"), ("h2o", "H2OKVCluster")])
def test_scored_methods_through_the_model(cpu_backend, method, cls):
    """The tensors captured at update_kv, fed to the oracle directly, give exactly what the cache then holds; decode
    positions continue from the true prompt length."""
    model = _llama(2)
    L, cap, W = 160, 48, 8
    ids = torch.randint(0, 512, (1, L), generator=torch.Generator().manual_seed(2))
    calls = []
    klass = getattr(pu, cls)
    orig = klass.update_kv

    def spy(self, k, q, v, am, g):
        r = orig(self, k, q, v, am, g)
        calls.append((self, k.clone(), q.clone(), v.clone(), r[0].clone()))
        return r
    pos_seen = []
    def rope_hook(mod, args, kwargs, out):
        pos_seen.append((kwargs.get("position_ids") if "position_ids" in kwargs else args[1]).clone())
    hook = model.model.rotary_emb.register_forward_hook(rope_hook, with_kwargs=True)
    try:
        mp.replace_llama(method)
        _set_knobs(model, window_size=W, max_capacity_prompt=cap, kernel_size=7, pooling="maxpool")
        klass.update_kv = spy
        out = _generate(model, ids, 2)
    finally:
        klass.update_kv = orig
        hook.remove()
        mp.replace_llama("fullkv")
    assert len(calls) == 2                                              # one compression per layer, prefill only
    for li, (cl, k, q, v, kc) in enumerate(calls):
        n_keep = cl._budget(L)
        ko, vo, idx, sc = oracle_compress(cl._method, q, k, v, W, n_keep, 7, "maxpool", return_indices=True, return_scores=True)
        assert torch.equal(kc, ko)
        layer = out.past_key_values.layers[li]
        assert torch.equal(layer.keys[:, :, :n_keep + W], ko) and layer.get_seq_length() == L + 1
        assert layer.keys.shape[2] == n_keep + W + 1
    assert [int(p[0, 0]) for p in pos_seen] == [0, L]                   # prefill at 0.., decode at the TRUE length


@pytest.mark.parametrize("method", ["snapkv", "pyramidkv", "streamingllm", "h2o"])
def test_layer_batching_gives_the_per_layer_cache(cpu_backend, monkeypatch, method):
    """pyramidkv_utils.BATCH_LAYERS: the patched forwards park every layer's update_kv in a PrefillBatch and the last
    layer runs them as ONE compress_batch call (per-layer budgets for PyramidKV) — cache bytes, lengths and generated
    tokens equal those of one update_kv per layer.  H2O is never parked (it scores with every query row)."""
    model = _llama(3)
    L, cap, W = 160, 48, 8
    ids = torch.randint(0, 512, (1, L), generator=torch.Generator().manual_seed(3))
    batch_calls = []

    def spy_batch(method_, qs, ks, vs, window, keeps, *a, **kw):
        batch_calls.append((len(ks), list(keeps), kw.get("q_rows"), None if qs is None else tuple(qs[0].shape),
                            None if kw.get("outs") is None else [o[0].data_ptr() for o in kw["outs"]]))
        return oracle_compress_batch(method_, qs, ks, vs, window, keeps, *a, **kw)
    monkeypatch.setattr(_kvc, "compress_batch", spy_batch)
    outs = {}
    try:
        mp.replace_llama(method)
        _set_knobs(model, window_size=W, max_capacity_prompt=cap, kernel_size=7, pooling="maxpool")
        for flag in (False, True):
            monkeypatch.setattr(pu, "BATCH_LAYERS", flag)
            outs[flag] = _generate(model, ids, 3)
    finally:
        mp.replace_llama("fullkv")
    if method == "h2o":
        assert batch_calls == []
    else:
        assert len(batch_calls) == 1 and batch_calls[0][0] == 3                 # one call, all three layers
        if method == "pyramidkv":
            assert len(set(batch_calls[0][1])) == 3                             # per-layer budgets travel with it
        if method != "streamingllm":
            assert batch_calls[0][2] == "window" and batch_calls[0][3] == (1, 32, W, 128)
    assert torch.equal(outs[True].sequences, outs[False].sequences)
    for la, lb in zip(outs[True].past_key_values.layers, outs[False].past_key_values.layers):
        assert torch.equal(la.keys, lb.keys) and torch.equal(la.values, lb.values)
        assert la.get_seq_length() == lb.get_seq_length() == L + 2
    assert len(outs[True].past_key_values._kvc_pending) == 0
    if method != "h2o":            # N1: the batched launch wrote K' / V' straight into the cache layers' own buffers
        ptrs = batch_calls[0][4]
        assert ptrs is not None and ptrs == [la._kbuf.data_ptr() for la in outs[True].past_key_values.layers]
        # ... each sized for its OWN budget (PyramidKV's layers differ), not for the largest of the prompt
        assert [la._kbuf.shape[2] for la in outs[True].past_key_values.layers] == [kk + W + pu.SPARE_ROWS for kk in batch_calls[0][1]]


def test_layer_batching_settles_an_unfinished_prefill(cpu_backend, monkeypatch):
    """A prefill that stops before the last layer (only layer 0 is run here) leaves its entry parked; the next forward
    of that layer flushes it first and then takes the decode branch at the true length."""
    from transformers import DynamicCache
    monkeypatch.setattr(pu, "BATCH_LAYERS", True)
    model = _llama(2)
    attn = model.model.layers[0].self_attn
    try:
        mp.replace_llama("snapkv")
        _set_knobs(model, window_size=8, max_capacity_prompt=48, kernel_size=7, pooling="maxpool")
        cache = DynamicCache()
        g = torch.Generator().manual_seed(4)
        hs = torch.randn(1, 100, 4096, generator=g)
        pos = model.model.rotary_emb(hs, torch.arange(100)[None])
        with torch.no_grad():
            attn(hs, position_embeddings=pos, attention_mask=None, past_key_values=cache)
            layer = cache.layers[0]
            assert len(cache._kvc_pending) == 1 and layer.get_seq_length() == 0
            pos1 = model.model.rotary_emb(hs[:, :1], torch.tensor([[100]]))
            attn(hs[:, :1], position_embeddings=pos1, attention_mask=None, past_key_values=cache)
        assert len(cache._kvc_pending) == 0 and layer.get_seq_length() == 101 and layer.keys.shape == (1, 32, 49, 128)
    finally:
        mp.replace_llama("fullkv")


def test_decode_appends_in_place(cpu_backend):
    """N1 (decode side): after the prefill the cache layer owns a buffer with spare rows; generated tokens are written
    in place (stable storage, no torch.cat per token) and the filled part is handed out as a view — the same bytes a
    cat-grown cache would hold; the buffer doubles when the spare rows run out."""
    from kvcache_factory_amd.cache import CompressedDynamicLayer
    g = torch.Generator().manual_seed(9)
    k0, v0 = torch.randn(1, 4, 10, 8, generator=g), torch.randn(1, 4, 10, 8, generator=g)
    layer = CompressedDynamicLayer()
    layer.RESERVE = 3
    layer.prefill(k0, v0, 100)
    ref_k, ref_v = k0, v0
    ptr = layer._kbuf.data_ptr()
    for step in range(8):
        kn, vn = torch.randn(1, 4, 1, 8, generator=g), torch.randn(1, 4, 1, 8, generator=g)
        ko, vo = layer.update(kn, vn)
        ref_k, ref_v = torch.cat([ref_k, kn], 2), torch.cat([ref_v, vn], 2)
        assert torch.equal(ko, ref_k) and torch.equal(vo, ref_v) and layer.get_seq_length() == 101 + step
        assert layer.get_mask_sizes(1) == (11 + step + 1, 0)
        if step < 3:
            assert layer._kbuf.data_ptr() == ptr                      # spare rows: same storage
    assert layer._kbuf.data_ptr() != ptr and layer._kbuf.shape[2] >= 18  # grew by doubling


def test_batch_reordering_works_on_the_layers_own_buffers(cpu_backend):
    """Beam search / num_return_sequences / contrastive search drive DynamicLayer.reorder_cache, batch_repeat_interleave and
    batch_select_indices, which stock transformers implements by REASSIGNING keys / values.  Those are computed properties
    here: the operations are restated on the buffers (prefix, spare rows, split tail), anything that would assign a real tensor to
    keys / values is refused instead of silently lost, reset() really empties the layer, the ragged cache (no batch dimension,
    as in the reference) refuses, and crop / offload raise."""
    from kvcache_factory_amd.cache import CompressedDynamicLayer, RaggedDynamicLayer, ThinkDynamicLayer
    g = torch.Generator().manual_seed(3)
    k0, v0 = torch.randn(3, 4, 10, 8, generator=g), torch.randn(3, 4, 10, 8, generator=g)
    layer = CompressedDynamicLayer()
    layer.prefill(k0, v0, 100)
    kn, vn = torch.randn(3, 4, 1, 8, generator=g), torch.randn(3, 4, 1, 8, generator=g)
    layer.update(kn, vn)
    ref_k, ref_v = torch.cat([k0, kn], 2), torch.cat([v0, vn], 2)
    beam = torch.tensor([2, 0, 0])
    layer.reorder_cache(beam)
    assert torch.equal(layer.keys, ref_k[beam]) and torch.equal(layer.values, ref_v[beam]) and layer.get_seq_length() == 101
    kn2, vn2 = torch.randn(3, 4, 1, 8, generator=g), torch.randn(3, 4, 1, 8, generator=g)
    ko, vo = layer.update(kn2, vn2)                                    # decoding goes on against the reordered histories
    assert torch.equal(ko, torch.cat([ref_k[beam], kn2], 2)) and torch.equal(vo, torch.cat([ref_v[beam], vn2], 2))
    layer.batch_repeat_interleave(2)
    assert layer.keys.shape[0] == 6 and torch.equal(layer.keys[1], ko[0]) and torch.equal(layer.keys[2], ko[1])
    layer.batch_select_indices(torch.tensor([0, 5]))
    assert layer.keys.shape[0] == 2 and torch.equal(layer.keys[1], ko[2])
    with pytest.raises(AttributeError):
        layer.keys = k0                                                # would have been a silent no-op
    with pytest.raises(NotImplementedError):
        layer.crop(-1)
    with pytest.raises(NotImplementedError):
        layer.offload()
    layer.reset()
    assert layer.get_seq_length() == 0 and layer.keys is None and layer.stored_length() == 0 and not layer.can_split()
    # ThinK's two-part cache reorders every part
    t = ThinkDynamicLayer()
    msk = torch.zeros(3, 4, 8, dtype=torch.bool)
    msk[..., [0, 2, 3, 5, 7]] = True
    t.prefill_think(torch.randn(3, 4, 6, 5, generator=g), k0, msk, torch.cat([v0[:, :, :6], v0], 2), 50)
    assert t._chan[0, 0].tolist()[:5] == [0, 2, 3, 5, 7] and t._chan[0, 0, 5:].eq(-1).all() and t._kp.shape[-1] == t._chan.shape[-1]
    kp, m, kk = t.key_pruned.clone(), t.mask.clone(), t.keys.clone()
    t.reorder_cache(beam)
    assert torch.equal(t.key_pruned, kp[beam]) and torch.equal(t.mask, m[beam]) and torch.equal(t.keys, kk[beam])
    t.reset()
    assert t.get_seq_length() == 0 and t.keys is None
    r = RaggedDynamicLayer()
    r.prefill(dict(k_flat=torch.zeros(5, 8), v_flat=torch.zeros(5, 8), seg_off=torch.tensor([0]), seg_len=torch.tensor([5]), lens=[5], slack=0), 9)
    for op, arg in ((r.reorder_cache, beam), (r.batch_repeat_interleave, 2), (r.batch_select_indices, beam)):
        with pytest.raises(NotImplementedError):
            op(arg)
    r.reset()
    assert r.get_seq_length() == 0 and r.keys is None


def test_prefill_batch_groups_by_layout(cpu_backend, monkeypatch):
    """PrefillBatch.flush issues one compress_batch per group of entries that share a layout (shape, strides, dtype, window,
    pooling ...); sinks receive their own results in the order the entries were added; H2O and pass-through are refused."""
    calls = []

    def spy(method_, qs, ks, vs, window, keeps, *a, **kw):
        calls.append((len(ks), tuple(ks[0].shape), list(keeps)))
        return oracle_compress_batch(method_, qs, ks, vs, window, keeps, *a, **kw)
    monkeypatch.setattr(_kvc, "compress_batch", spy)
    g = torch.Generator().manual_seed(11)
    mk = lambda L: (torch.randn(1, 2, L, 64, generator=g), torch.randn(1, 4, L, 64, generator=g), torch.randn(1, 2, L, 64, generator=g))   # noqa: E731
    pb, got = pu.PrefillBatch(), {}
    cl = pu.SnapKVCluster(window_size=4, max_capacity_prompt=20, kernel_size=3, pooling="maxpool")
    for name, L in (("a", 64), ("b", 96), ("c", 64)):
        k, q, v = mk(L)
        assert pb.add(cl, k, q, v, lambda kc, vc, name=name: got.__setitem__(name, (kc, vc)), tag=name)
        ko, vo = oracle_compress(_kvc.SNAPKV, q, k, v, 4, 16, 3, "maxpool")
        got[name + "_want"] = (ko, vo)
    k, q, v = mk(10)
    assert not pb.add(cl, k, q, v, None)                                         # q_len < cap: pass-through is the caller's
    assert not pb.add(pu.H2OKVCluster(window_size=4, max_capacity_prompt=20), *[t for t in (mk(64)[0], mk(64)[1], mk(64)[2])], None)
    assert len(pb) == 3 and pb.holds("b")
    pb.flush()
    assert len(pb) == 0 and sorted(c[0] for c in calls) == [1, 2]               # L = 64 twice in one call, L = 96 alone
    for name in "abc":
        assert torch.equal(got[name][0], got[name + "_want"][0]) and torch.equal(got[name][1], got[name + "_want"][1])


@pytest.mark.parametrize("group", [0, 1, 2])
def test_layer_batching_flushes_in_groups(cpu_backend, monkeypatch, group):
    """pyramidkv_utils.GROUP_LAYERS bounds what stays parked: every group-th layer (and the last) flushes — 5 layers with
    group 2 give calls of 2, 2 and 1 layers — and the cache is the per-layer one either way."""
    model = _llama(5)
    L, cap, W = 120, 40, 8
    ids = torch.randint(0, 512, (1, L), generator=torch.Generator().manual_seed(7))
    sizes = []

    def spy_batch(method_, qs, ks, vs, window, keeps, *a, **kw):
        sizes.append(len(ks))
        return oracle_compress_batch(method_, qs, ks, vs, window, keeps, *a, **kw)
    monkeypatch.setattr(_kvc, "compress_batch", spy_batch)
    monkeypatch.setattr(pu, "GROUP_LAYERS", group)
    outs = {}
    try:
        mp.replace_llama("pyramidkv")
        _set_knobs(model, window_size=W, max_capacity_prompt=cap, kernel_size=7, pooling="maxpool")
        for flag in (False, True):
            monkeypatch.setattr(pu, "BATCH_LAYERS", flag)
            outs[flag] = _generate(model, ids, 2)
    finally:
        mp.replace_llama("fullkv")
    assert sizes == {0: [5], 1: [1] * 5, 2: [2, 2, 1]}[group]
    assert torch.equal(outs[True].sequences, outs[False].sequences)
    for la, lb in zip(outs[True].past_key_values.layers, outs[False].past_key_values.layers):
        assert torch.equal(la.keys, lb.keys) and torch.equal(la.values, lb.values)


@pytest.mark.parametrize("impl", ["eager", "sdpa"])
@pytest.mark.parametrize("method", ["pyramidkv", "snapkv"])
def test_generate_with_eager_and_sdpa_attention(cpu_backend, monkeypatch, method, impl):
    """Decode over layers that store DIFFERENT numbers of tokens (PyramidKV): transformers builds one mask per forward from
    the first layer's lengths and its eager attention adds it uncut — the patched forward trims it to each layer's keys.
    Eager and sdpa generate the same tokens; batched and per-layer compression too."""
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=512, intermediate_size=256, num_hidden_layers=4, num_attention_heads=8,
                      num_key_value_heads=2, head_dim=64, vocab_size=512, max_position_embeddings=4096,
                      attn_implementation=impl)
    torch.manual_seed(0)
    model = LlamaForCausalLM(cfg).eval()
    L, cap, W = 200, 40, 8
    ids = torch.randint(0, 512, (1, L), generator=torch.Generator().manual_seed(8))
    outs = {}
    try:
        mp.replace_llama(method)
        _set_knobs(model, window_size=W, max_capacity_prompt=cap, kernel_size=7, pooling="maxpool")
        for flag in (False, True):
            monkeypatch.setattr(pu, "BATCH_LAYERS", flag)
            outs[flag] = _generate(model, ids, 4)
    finally:
        mp.replace_llama("fullkv")
    assert torch.equal(outs[True].sequences, outs[False].sequences) and outs[True].sequences.shape[1] == L + 4
    stored = [la.keys.shape[2] for la in outs[True].past_key_values.layers]
    if method == "pyramidkv":
        assert len(set(stored)) > 1                                     # the layers really differ
    test_generate_with_eager_and_sdpa_attention.tokens = getattr(test_generate_with_eager_and_sdpa_attention, "tokens", {})
    test_generate_with_eager_and_sdpa_attention.tokens[(method, impl)] = outs[True].sequences
    other = test_generate_with_eager_and_sdpa_attention.tokens.get((method, "eager" if impl == "sdpa" else "sdpa"))
    if other is not None:
        assert torch.equal(other, outs[True].sequences)


def test_decode_mask_is_trimmed_per_layer(cpu_backend):
    """A materialised 4-D mask wider than a layer's stored keys (built from another layer's length) is cut to its last
    columns in the decode branch; one that is too narrow is an error, not a silent broadcast."""
    from transformers import DynamicCache
    model = _llama(2)
    attn = model.model.layers[1].self_attn
    try:
        mp.replace_llama("snapkv")
        _set_knobs(model, window_size=8, max_capacity_prompt=48, kernel_size=7, pooling="maxpool")
        cache = DynamicCache()
        hs = torch.randn(1, 100, 4096, generator=torch.Generator().manual_seed(4))
        pos = model.model.rotary_emb(hs, torch.arange(100)[None])
        with torch.no_grad():
            attn(hs, position_embeddings=pos, attention_mask=None, past_key_values=cache)
            pos1 = model.model.rotary_emb(hs[:, :1], torch.tensor([[100]]))
            ref, _ = attn(hs[:, :1], position_embeddings=pos1, attention_mask=None, past_key_values=cache)
            cache.layers[1]._stored -= 1; cache.layers[1].true_length -= 1          # undo the append, decode again
            wide = torch.zeros(1, 1, 1, 80)                                          # 80 > 49 stored keys
            out, _ = attn(hs[:, :1], position_embeddings=pos1, attention_mask=wide, past_key_values=cache)
            assert torch.allclose(out, ref, atol=1e-5)
            cache.layers[1]._stored -= 1; cache.layers[1].true_length -= 1
            with pytest.raises(RuntimeError, match="attention mask covers"):
                attn(hs[:, :1], position_embeddings=pos1, attention_mask=torch.zeros(1, 1, 1, 20), past_key_values=cache)
    finally:
        mp.replace_llama("fullkv")


def test_l2norm_cluster_and_factory(cpu_backend):
    """L2NormCluster (pyramidkv_utils.py:394-429): constructor / reset signature, the two pass-through branches returning the
    same objects (:413-416), the kept branch against the reference fixture, and init_l2norm's defaults (:954-968)."""
    m = G.MANIFEST["l2norm_bf16_L600"]
    q, k, v = G.inputs(m)
    g = m["Hq"] // m["Hkv"]
    for kk, vv in ((k, v), (k[:, ::g].contiguous(), v[:, ::g].contiguous())):      # expanded like the reference's caller, and GQA-native
        ko, vo = pu.L2NormCluster(max_capacity_prompt=m["cap"], layer_idx=5, skip_layers=[0, 1]).update_kv(kk, q, vv, None, g)
        assert list(ko.shape) == m["out_shape"]
        assert G.sha(ko) == m["k_out_sha256"] and G.sha(vo) == m["v_out_sha256"]
    ko, vo = pu.L2NormCluster(max_capacity_prompt=m["cap"], layer_idx=1, skip_layers=[0, 1]).update_kv(k, q, v, None, g)
    assert ko is k and vo is v                                                       # skipped layer
    ko, vo = pu.L2NormCluster(max_capacity_prompt=4096, layer_idx=5).update_kv(k, q, v, None, g)
    assert ko is k and vo is v                                                       # prompt shorter than the budget
    with pytest.raises(AssertionError):
        pu.L2NormCluster(64).update_kv(k[:, :, :50], q, v, None, g)                  # :407

    class Cfg:
        pass

    class Attn:
        def __init__(self):
            self.config, self.layer_idx = Cfg(), 7
    a = Attn()
    pu.init_l2norm(a)
    assert (a.config.max_capacity_prompt, a.config.skip_layers) == (4096, [0, 1])
    assert isinstance(a.kv_cluster, pu.L2NormCluster) and a.kv_cluster.layer_idx == 7
    first = a.kv_cluster
    a.config.max_capacity_prompt = 96
    pu.init_l2norm(a)
    assert a.kv_cluster is not first and a.kv_cluster.max_capacity_prompt == 96


def test_l2norm_through_the_model(cpu_backend):
    """replace_llama("l2norm") on a 3-layer Llama (CPU, oracle-backed compute): layers 0 and 1 are skipped and keep the whole
    prompt, layer 2 keeps max_capacity_prompt rows; decode appends to each and generation runs."""
    model = _llama(layers=3)
    ids = torch.randint(0, 512, (1, 96))
    mp.replace_llama("l2norm")
    try:
        model2 = _llama(layers=3)
        _set_knobs(model2, max_capacity_prompt=40, skip_layers=[0, 1])
        out = _generate(model2, ids, 3)
        cache = out.past_key_values
        assert [cache.layers[i].keys.shape[2] for i in range(3)] == [96 + 2, 96 + 2, 40 + 2]
        assert cache.get_seq_length() == 96 + 2
    finally:
        mp.replace_llama("fullkv")


@pytest.mark.parametrize("name", ["merge_snapkv_bf16_L600", "merge_snapkv_fp16_W32_L1024", "merge_pyramidkv_bf16_L1024", "merge_h2o_bf16_L300"])
def test_cluster_with_pivot_merge_matches_reference_fixture(cpu_backend, name):
    """Clusters built with merge="pivot" (merge_kv, pyramidkv_utils.py:119-170 behind :337-339): update_kv returns the merged K / V
    of the reference — key rows [window, selected], value rows [selected, window] — for the expanded and the GQA-native K/V."""
    m = G.MANIFEST[name]
    q, k, v = G.inputs(m)
    kw = dict(window_size=m["W"], max_capacity_prompt=m["cap"], kernel_size=m["kernel"], pooling=m["pooling"], merge="pivot")
    base = m["method"].split("_")[1]
    cls = {"snapkv": pu.SnapKVCluster, "h2o": pu.H2OKVCluster,
           "pyramidkv": lambda **a: pu.PyramidKVCluster(num_hidden_layers=m["n_layers"], layer_idx=m["layer_idx"], **a)}[base]
    g = m["Hq"] // m["Hkv"]
    for kk, vv in ((k, v), (k[:, ::g].contiguous(), v[:, ::g].contiguous())):
        ko, vo = cls(**kw).update_kv(kk, q, vv, None, g)
        assert list(ko.shape) == m["out_shape"]
        assert G.sha(ko) == m["k_out_sha256"] and G.sha(vo) == m["v_out_sha256"]


@pytest.mark.parametrize("name", ["think_bf16_L600", "think_fp16_D64_L300", "think_fp32_L600"])
def test_update_think_matches_reference_fixture(cpu_backend, name):
    """SnapKVCluster.update_think (pyramidkv_utils.py:349-392): the four return values of the reference — pruned keys, recent keys,
    channel mask, compressed values — and the two-tensor pass-through; init_think's defaults (:926-952)."""
    m = G.MANIFEST[name]
    a = G.arrays(name)
    q, k, v = G.inputs(m)
    g = m["Hq"] // m["Hkv"]
    cl = pu.SnapKVCluster(window_size=m["W"], max_capacity_prompt=m["cap"], kernel_size=m["kernel"], pooling=m["pooling"],
                          recent_size=m["recent"], ratio=m["ratio"])
    pruned, recent, mask, vo = cl.update_think(k[:, ::g].contiguous(), q, v[:, ::g].contiguous(), None, g)
    assert list(pruned.shape) == m["out_shape"] and mask.dtype == torch.bool
    assert G.sha(pruned) == m["pruned_sha256"] and G.sha(recent.contiguous()) == m["recent_sha256"] and G.sha(vo) == m["v_out_sha256"]
    assert torch.equal(mask[0], torch.from_numpy(a["keep"]).bool())
    short = pu.SnapKVCluster(window_size=8, max_capacity_prompt=4096).update_think(k, q, v, None, g)
    assert len(short) == 2 and short[0] is k and short[1] is v

    class Cfg:
        pass

    class Attn:
        def __init__(self):
            self.config, self.layer_idx = Cfg(), 0
    at = Attn()
    pu.init_think(at)
    assert (at.config.recent_size, at.config.ratio, at.config.max_capacity_prompt) == (32, 0.4, 4096)
    assert isinstance(at.kv_cluster, pu.SnapKVCluster) and at.kv_cluster.recent_size == 32


def test_pivot_merge_through_the_model(cpu_backend):
    """replace_llama("snapkv") with config.merge = "pivot" (run_longbench.py writes the knob onto every layer's config): prefill
    takes the direct path (merge_kv is not batched), the cache holds cap rows per layer, decode appends and generation runs."""
    ids = torch.randint(0, 512, (1, 96))
    mp.replace_llama("snapkv")
    try:
        model = _llama(layers=2, dtype=torch.bfloat16)                         # merge_kv is built for the 16-bit dtypes
        _set_knobs(model, window_size=8, max_capacity_prompt=40, kernel_size=7, pooling="maxpool", merge="pivot")
        out = _generate(model, ids, 3)
        cache = out.past_key_values
        assert [cache.layers[i].keys.shape[2] for i in range(2)] == [40 + 2, 40 + 2]
        assert cache.get_seq_length() == 96 + 2
    finally:
        mp.replace_llama("fullkv")


def test_think_through_the_model(cpu_backend):
    """replace_llama("think") (monkeypatch.py:80-83): prefill stores channel-pruned keys + recent keys + mask + values in a
    ThinkDynamicLayer, decode attends over both parts (llama_model_think.py:167-196).  With ratio = 0 nothing is pruned and
    the generated tokens must equal plain SnapKV's; with ratio = 0.4 the cache has the reference's shapes."""
    from kvcache_factory_amd.cache import ThinkDynamicLayer
    ids = torch.randint(0, 512, (1, 96))
    knobs = dict(window_size=8, max_capacity_prompt=40, kernel_size=7, pooling="maxpool")
    seqs = {}
    try:
        for method, extra in (("snapkv", {}), ("think", dict(recent_size=16, ratio=0.0)), ("think", dict(recent_size=16, ratio=0.4))):
            mp.replace_llama("fullkv"); mp.replace_llama(method)
            model = _llama(layers=2)
            _set_knobs(model, **knobs, **extra)
            out = _generate(model, ids, 4)
            seqs[(method, extra.get("ratio"))] = out.sequences
            if method == "think":
                layer = out.past_key_values.layers[0]
                assert isinstance(layer, ThinkDynamicLayer)
                k = int(128 * extra["ratio"])
                assert tuple(layer.key_pruned.shape) == (1, 32, 40 - 16, 128 - k) and tuple(layer.mask.shape) == (1, 32, 128)
                assert layer.keys.shape[2] == 16 + 3 and layer.values.shape[2] == 40 + 3 and layer.get_seq_length() == 96 + 3
                assert int(layer.mask.sum()) == 32 * (128 - k)
        assert torch.equal(seqs[("snapkv", None)], seqs[("think", 0.0)])
        with pytest.raises(NotImplementedError):
            mp.replace_mistral("think")
    finally:
        mp.replace_llama("fullkv")


@pytest.mark.parametrize("name", ["cam_bf16_L300", "cam_fp16_W32_D64_L257", "cam_bf16_passthrough"])
def test_cam_cluster_given_the_references_draws(cpu_backend, monkeypatch, name):
    """CAMKVCluster.update_kv (pyramidkv_utils.py:431-513), init_CAM (:970-988): same constructor / signature; with the draw replaced
    by the reference's recorded outcomes the returned K', V' and the IN-PLACE merged value_states are the reference's bytes; the
    pass-through returns the same objects; H_kv-head values are expanded instead of merged in place."""
    m, arr = G.MANIFEST[name], G.arrays(name)
    q, k, v = G.inputs(m)
    g = m["Hq"] // m["Hkv"]
    cl = pu.CAMKVCluster(start_budget_ratio=m["start_ratio"], window_size=m["W"], max_capacity_prompt=m["cap"], kernel_size=m["kernel"], pooling=m["pooling"])
    if m["passthrough"]:
        ko, vo = cl.update_kv(k, q, v, None, g)
        assert ko is k and vo is v
        return
    mask = torch.from_numpy(arr["mask"])[None].float()
    monkeypatch.setattr(pu.CAMKVCluster, "draw", staticmethod(lambda prob: mask))
    vv = v.clone()
    ko, vo = cl.update_kv(k, q, vv, None, g)                              # expanded values: merged in place, like the reference
    if m["dtype"] == "bf16":                                              # (fp16: a flipped score may move the selection, see the oracle test)
        assert G.sha(ko) == m["k_out_sha256"] and G.sha(vo) == m["v_out_sha256"]
    assert G.sha(vv) == m["merged_values_sha256"]
    v_kv = v[:, ::g].contiguous()
    ko2, vo2 = cl.update_kv(k[:, ::g].contiguous(), q, v_kv, None, g)     # GQA-native: same outputs, the caller's values untouched
    assert torch.equal(ko2, ko) and torch.equal(vo2, vo) and torch.equal(v_kv, v[:, ::g])

    class _Attn:
        class config:
            pass
    at = _Attn()
    at.config.merge = None
    pu.init_CAM(at)
    assert isinstance(at.kv_cluster, pu.CAMKVCluster) and at.config.max_capacity_prompt == 2048 and at.config.window_size == 32


def test_cam_through_the_model(cpu_backend):
    """replace_llama("cam") (monkeypatch.py:39-43): the prefill merges into the repeat_kv-expanded values and attends over them, the
    compressed cache holds cap rows per head and decoding goes on at the true length."""
    ids = torch.randint(0, 512, (1, 96))
    try:
        mp.replace_llama("cam")
        model = _llama(layers=2)
        _set_knobs(model, window_size=8, max_capacity_prompt=40, kernel_size=5, pooling="avgpool", merge=None)
        torch.manual_seed(0)
        out = _generate(model, ids, 3)
        layer = out.past_key_values.layers[0]
        assert out.sequences.shape[1] == 99 and layer.get_seq_length() == 98 and layer.keys.shape == (1, 32, 40 + 2, 128)
    finally:
        mp.replace_llama("fullkv")
