"""The C-ABI library without a GPU: it loads, exports every symbol include/kvc.h declares, validates parameters
with the reference's error behaviour, and its host-only arithmetic (PyramidKV schedule) matches the fixtures.
No compute entry point is called with real pointers here."""
import ctypes
import os
import re

import pytest

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "kvc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kvc_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(kvc):
    syms = _declared_symbols()
    assert set(syms) == set(kvc.EXPORTS), (syms, kvc.EXPORTS)
    lib = ctypes.CDLL(kvc.LIB_PATH)
    for s in syms:
        assert getattr(lib, s) is not None
    assert kvc.lib().kvc_version() == 5


def test_params_struct_matches_header(kvc):
    """ctypes mirror has the header's field order and size (14 x int32 + 10 x int64 = 136 bytes)."""
    text = open(os.path.join(ROOT, "include", "kvc.h")).read()
    body = text[text.index("typedef struct kvc_params {"):text.index("} kvc_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in re.findall(r"int(?:32|64)_t\s+([^;]+);", body):
        fields += [f.strip() for f in decl.split(",")]
    assert fields == [f[0] for f in kvc.Params._fields_]
    assert ctypes.sizeof(kvc.Params) == 14 * 4 + 10 * 8


def _p(kvc, **kw):
    p = kvc.Params()
    base = dict(method=kvc.SNAPKV, dtype=kvc.BF16, bsz=1, n_q_heads=32, n_kv_heads=8, q_len=8000, head_dim=128,
                window=8, k=120, kernel_size=7, pooling=kvc.POOL_MAX, tie_mode=kvc.TIES_CANONICAL,
                q_stride_b=8000 * 4096, q_stride_h=128, q_stride_l=4096,
                k_stride_b=8 * 8000 * 128, k_stride_h=8000 * 128, k_stride_l=128,
                v_stride_b=8 * 8000 * 128, v_stride_h=8000 * 128, v_stride_l=128)
    base.update(kw)
    for k, v in base.items():
        setattr(p, k, v)
    return p


def test_workspace_bytes_and_layout(kvc):
    L = kvc.lib()
    p = _p(kvc)
    n = L.kvc_workspace_bytes(ctypes.byref(p))
    # logits 32*8000*8*2 + tile maxima + chunk sums + scores + indices, each 256-B aligned
    # logits 32*8000*8*2, tile maxima, chunk sums, scores, indices — each 256-B aligned
    assert n >= 32 * 8000 * 8 * 2 + 32 * 7992 * 2 + 32 * 120 * 8 and n < 8 * 1024 * 1024
    keeps = (ctypes.c_int32 * 3)(120, 60, 17)
    nb = L.kvc_workspace_bytes_batch(ctypes.byref(p), 3, keeps)
    assert 3 * (32 * 8000 * 8 * 2) <= nb < 3 * n + 4096
    assert L.kvc_workspace_bytes_batch(ctypes.byref(p), 0, keeps) == 0
    many = (ctypes.c_int32 * 40)(*([120] * 40))          # > 32 items run as chunks in ONE workspace: sized for the largest chunk
    assert L.kvc_workspace_bytes_batch(ctypes.byref(p), 40, many) == L.kvc_workspace_bytes_batch(ctypes.byref(p), 32, many)
    offs = (ctypes.c_size_t * 3)()
    assert L.kvc_workspace_layout(ctypes.byref(p), ctypes.byref(offs)) == 0
    assert offs[0] == 0 and offs[1] % 256 == 0 and offs[2] > offs[1]
    assert L.kvc_workspace_bytes(ctypes.byref(_p(kvc, method=kvc.STREAMINGLLM))) == 0


@pytest.mark.parametrize("kw,code,msg", [
    (dict(dtype=7), -1, "dtype"),
    (dict(method=9), -1, "method"),
    (dict(n_kv_heads=5), -1, "multiple"),
    (dict(k=7993), -1, "k="),                           # topk k out of range in the reference
    (dict(q_len=8), -1, "q_len"),
    (dict(window=0), -1, "window"),
    (dict(pooling=5), -1, "Pooling method not supported"),   # pyramidkv_utils.py:333
    (dict(kernel_size=6), -2, "even kernel_size"),
    (dict(head_dim=96), -2, "head_dim"),
    (dict(window=128, k=100), -2, "window"),
    (dict(tie_mode=3), -1, "tie_mode"),
])
def test_invalid_params_are_rejected_before_any_gpu_work(kvc, kw, code, msg):
    L = kvc.lib()
    p = _p(kvc, **kw)
    assert L.kvc_workspace_bytes(ctypes.byref(p)) == 0
    rc = L.kvc_scores(ctypes.byref(p), None, None, None, None, 0, None)
    assert rc == code
    assert msg in L.kvc_last_error().decode()


def test_null_and_misaligned_pointers_are_rejected(kvc):
    L = kvc.lib()
    p = _p(kvc)
    assert L.kvc_compress(ctypes.byref(p), None, None, None, None, None, None, None, None, 0, None) == kvc.ERR_INVALID
    fake = ctypes.c_void_p(0x10000)
    odd = ctypes.c_void_p(0x10008)
    assert L.kvc_scores(ctypes.byref(p), odd, fake, fake, fake, 1 << 30, None) == kvc.ERR_ALIGNMENT
    assert L.kvc_scores(ctypes.byref(p), fake, fake, fake, None, 0, None) == kvc.ERR_WORKSPACE
    assert L.kvc_scores(ctypes.byref(p), fake, fake, fake, ctypes.c_void_p(0x20000), 16, None) == kvc.ERR_WORKSPACE
    assert "workspace too small" in L.kvc_last_error().decode()
    bad = _p(kvc, k_stride_l=130)
    assert L.kvc_scores(ctypes.byref(bad), fake, fake, fake, ctypes.c_void_p(0x20000), 1 << 30, None) == kvc.ERR_ALIGNMENT


@pytest.mark.parametrize("name", G.names(lambda m: m["method"] == "pyramidkv"))
def test_pyramid_k_matches_reference(kvc, name):
    m = G.MANIFEST[name]
    assert kvc.pyramid_k(m["cap"], m["W"], m["L"], m["layer_idx"], m["n_layers"]) == (-1 if m["passthrough"] else m["n_keep"])


def test_h2o_workspace_is_bounded(kvc):
    """Workspace of one H2O call at Llama-3-8B shapes: exact mode <= 1 GiB of logits (+ column partials) instead of the whole
    [32][L][L] matrix (4.1 GB at 8k, 65 GB at 32k); fast mode: two floats per query row and the scores."""
    import ctypes
    for L in (8000, 32000):
        p = kvc.Params()
        for kk, vv in dict(method=kvc.H2O, dtype=kvc.BF16, bsz=1, n_q_heads=32, n_kv_heads=8, q_len=L, head_dim=128, window=8, k=120,
                           kernel_size=7, pooling=kvc.POOL_NONE, tie_mode=kvc.TIES_TORCH_CPU, q_stride_b=L * 4096, q_stride_h=128,
                           q_stride_l=4096, k_stride_b=8 * L * 128, k_stride_h=L * 128, k_stride_l=128, v_stride_b=8 * L * 128,
                           v_stride_h=L * 128, v_stride_l=128).items():
            setattr(p, kk, vv)
        exact = kvc.lib().kvc_workspace_bytes(ctypes.byref(p))
        p.dot_mode = kvc.DOT_MFMA16
        fast = kvc.lib().kvc_workspace_bytes(ctypes.byref(p))
        assert 0 < exact <= (1 << 30) + 32 * (L // 256 + 2) * L * 4 + (1 << 22), exact
        assert 0 < fast <= 32 * L * 8 + 32 * L * 2 + 32 * 120 * 8 + (1 << 16), fast


def test_product_refuses_cpu_tensors(kvc):
    """No CPU fallback: CPU tensors raise instead of silently computing somewhere else."""
    import torch
    from kvcache_factory_amd import pyramidkv_utils as pu
    q, k, v = (torch.randn(1, 2, 64, 64) for _ in range(3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pu.SnapKVCluster(8, 40).update_kv(k, q, v, None, 1)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: importing the whole product package must not pull it in."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); "
            "import kvcache_factory_amd.pyramidkv_utils, kvcache_factory_amd.monkeypatch, kvcache_factory_amd.llama_model, "
            "kvcache_factory_amd.mistral_model, kvcache_factory_amd.synth; "
            "bad = [m for m in sys.modules if m.split('.')[0] == 'oracle']; assert not bad, bad" % ROOT)
    subprocess.check_call([sys.executable, "-c", code])


def test_graft_entry_build_runs(kvc):
    """__graft_entry__.build() — the driver's "does it build" check: make for the HIP library (gfx950 cross-compile) and
    the oracle, import of the package, version and exports."""
    import importlib
    ge = importlib.import_module("__graft_entry__")
    ge.build()
