#!/usr/bin/env python3
"""bench.py — KV tokens compressed/sec for the post-prefill scoring+eviction hot path on MI355X.

Metric (BASELINE.json): "KV tokens compressed/sec + selection-index exact-match, Llama-3-8B 8k->128".
One PROMPT = 32 `update_kv` calls (one per layer of Llama-3-8B, each on that layer's own synthetic Q/K/V, resident in
HBM before timing), run through the C-ABI exactly as the patched attention forward does.  One STEP = `--prompts`
consecutive prompts (default per config, chosen so that the K timed steps last about a second: the driver's
utilisation sampler and its wall clock can see them), cycling over `--sets` distinct resident prompts so the K working
set (0.5 GB per prompt at 8k) is never cache resident.  tokens/step = prompts * q_len * 32 layers.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c2_w16|c2_w32|c2_w64|c3|c4|c5] [--tie-mode torch_cpu|canonical]
                    [--mode batch|calls] [--in-flight 1..4]

N > 1 (torchrun, one rank per GPU): the path shards by independent prompts/layers with no exchange, so every rank
runs its own replica of the workload (weak scaling); time = max over ranks, value = N * tokens / time.
Prints ONE JSON line (rank 0).  DESIGN.md §5 has the byte accounting.
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from kvcache_factory_amd import _kvc, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured float4-copy rate
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak (MI355X_MICROARCH.md); the exact H2O kernel runs the f32-input MFMA (157 TF)
MFMA_F32_PEAK_TFLOPS = 157.3 # f32-input MFMA (v_mfma_f32_32x32x2_f32: 64 cycles per SIMD, 256 CUs x 4 SIMDs at 2.4 GHz): the exact dot mode's pipe
HQ, HKV, D, LAYERS = 32, 8, 128, 32

CONFIGS = {
    "c2": dict(method="snapkv", L=8000, cap=128, W=8, kernel=7, pooling="maxpool", dtype=torch.bfloat16, prompts=40,
               desc="SnapKV Llama-3-8B shapes (Hq=32,Hkv=8,D=128), seq_len=8000 -> max_capacity_prompt=128, bf16, W=8, maxpool7"),
    "c2_w32": dict(method="snapkv", L=8000, cap=128, W=32, kernel=7, pooling="maxpool", dtype=torch.bfloat16, prompts=16,
                   desc="SnapKV 8k->128, W=32 (needle-runner window)"),
    "c2_w16": dict(method="snapkv", L=8000, cap=128, W=16, kernel=7, pooling="maxpool", dtype=torch.bfloat16, prompts=24,
                   desc="SnapKV 8k->128, W=16 (two M-tiles per KV head; cap - W = 112 kept)"),
    "c2_w64": dict(method="snapkv", L=8000, cap=128, W=64, kernel=7, pooling="maxpool", dtype=torch.bfloat16, prompts=8,
                   desc="SnapKV 8k->128, W=64 (the library's window default; cap - W = 64 kept)"),
    "c4": dict(method="pyramidkv", L=8000, cap=128, W=8, kernel=7, pooling="maxpool", dtype=torch.bfloat16, prompts=32,
               desc="PyramidKV 8k, total budget 128x32 (k_l = 234..17)"),
    "c5": dict(method="pyramidkv", L=32000, cap=2048, W=8, kernel=7, pooling="maxpool", dtype=torch.bfloat16, prompts=6, sets=1,
               desc="PyramidKV Mistral-7B shapes, 32k -> 2048 (k_l = 3978..103)"),
    "c3": dict(method="h2o", L=8000, cap=128, W=8, kernel=7, pooling=None, dtype=torch.bfloat16, layers=4, prompts=1,
               desc="H2O heavy-hitter scoring, Llama-3-8B shapes, seq_len=8000 -> 128, bf16 (all 8000 query rows score; "
                    "4 layers per prompt here)"),
}
METHODS = {"snapkv": _kvc.SNAPKV, "pyramidkv": _kvc.PYRAMIDKV, "h2o": _kvc.H2O}


def algorithmic_bytes(L, W, k, es=2):
    """GQA-aware bytes one layer call must move (K scan once, V never scanned).
    Returns (K-scan kernel's share, whole path, selection stage: scores read + indices written)."""
    scan = HKV * L * D * es + HQ * W * D * es
    compact = 2 * HQ * k * D * es + 2 * HKV * W * D * es + 2 * HQ * (k + W) * D * es + 8 * HQ * k
    select = HQ * (L - W) * es + 8 * HQ * k
    return scan, scan + compact, select


def layer_budgets(cfg):
    if cfg["method"] == "pyramidkv":
        return [_kvc.pyramid_k(cfg["cap"], cfg["W"], cfg["L"], l, LAYERS) for l in range(LAYERS)]
    return [cfg["cap"] - cfg["W"]] * LAYERS


_STREAMS = {}


def side_streams(dev, n):
    key = (dev.index, n)
    if key not in _STREAMS:
        _STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _STREAMS[key]


def make_inputs(cfg, dev, seed0, expanded=False):
    """One resident prompt: LAYERS x (q [1,32,L,128] in [L,H,D] memory order, k, v [1,8,L,128])."""
    out = []
    for l in range(LAYERS):
        q, k, v = synth.make_qkv(HQ, HKV, cfg["L"], D, cfg["dtype"], seed0 + l, expanded=expanded, device=dev)
        if expanded:
            k, v = k.contiguous(), v.contiguous()
        out.append((q, k, v))
    return out


class Prompt:
    """One resident prompt and its compression, in one of the launch modes:
    batch — ONE kvc_compress_batch call (every kernel launched once for the 32 layers; what the patched forward's
            PrefillBatch does, group = all layers);  calls — 32 kvc_compress calls spread over `streams` HIP streams."""

    def __init__(self, cfg, dev, tie_mode, mode, qkv, ks, n_streams=1, stream=None, dot_mode=None):
        self.mode, self.dev = mode, dev
        method = METHODS[cfg["method"]]
        home = stream if stream is not None else torch.cuda.current_stream(dev)
        if mode == "batch":
            with torch.cuda.stream(home):                   # the plan's workspace belongs to the stream it runs on
                self.bp = _kvc.BatchPlan(method, qkv, cfg["W"], ks, cfg["kernel"], cfg["pooling"], tie_mode,
                                         want_indices=True, dot_mode=dot_mode)
            self.handle = ctypes.c_void_p(home.cuda_stream)
        else:
            self.plans = []
            streams = side_streams(dev, n_streams) if n_streams > 1 else [home]
            for l, (q, k, v) in enumerate(qkv):
                st = streams[l % len(streams)]
                with torch.cuda.stream(st):
                    p = _kvc.CompressPlan(method, q, k, v, cfg["W"], ks[l], cfg["kernel"], cfg["pooling"], tie_mode,
                                          want_indices=True, dot_mode=dot_mode)
                p.handle = ctypes.c_void_p(st.cuda_stream)
                self.plans.append(p)
            self.streams = streams

    def run(self):
        if self.mode == "batch":
            self.bp.run(self.handle)
        else:
            for p in self.plans:
                p.run(p.handle)


def reduce_max_time(dt, dist, device):
    """Every rank times its own replica; the job time is the slowest rank's (MAX all-reduce; RCCL on GPUs, gloo in the
    CPU test).  No data-path collective exists on this path."""
    if dist is None:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_value(world, steps, tokens_per_step, dt):
    """Whole-job throughput of `world` independent replicas (weak scaling): all ranks' tokens over the max time."""
    return world * steps * tokens_per_step / dt


def rank_seed(rank):
    """Replicas compress different prompts: layer seeds are offset per rank."""
    return 1000 * rank


def time_steps(prompts, per_step, steps, warmup, dev, dist, streams=None):
    """W warm-up steps, then exactly K timed steps bracketed by barrier + synchronize on both sides.  A step runs `per_step`
    prompts, cycling over the resident ones (each on the stream(s) it was built for; `streams` lists every side stream in
    use so that a step's end mark waits for all of them).  Returns (wall seconds of the K steps, per-step HIP-event ms)."""
    def step(i0):
        for j in range(per_step):
            prompts[(i0 + j) % len(prompts)].run()
    cur = torch.cuda.current_stream(dev)

    def mark():
        e = torch.cuda.Event(enable_timing=True)
        for s in streams or ():                             # a step ends when every stream is done with it
            cur.wait_stream(s)
        e.record(cur)
        for s in streams or ():
            s.wait_stream(cur)
        return e
    i = 0
    for _ in range(warmup):
        step(i)
        i += per_step
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    marks = [mark()]
    t0 = time.perf_counter()
    for _ in range(steps):
        step(i)
        i += per_step
        marks.append(mark())
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    per = [marks[j].elapsed_time(marks[j + 1]) for j in range(steps)]
    return reduce_max_time(dt, dist, dev), per


def kernel_breakdown(prompt, dev, tie_mode, reps=10):
    """Live HIP-event time of each kernel of ONE batched launch (32 layers), by running the same kvc_compress_batch with
    debug_stage_mask subsets on the launch stream: bit0 K scan, bits1-2 softmax+pool, bit5 selection (+ gather), bit6
    gather alone."""
    bp = prompt.bp
    stream = prompt.handle

    def timed(mask):
        pp = _kvc.Params.from_buffer_copy(bp.p)
        pp.debug_stage_mask = mask
        assert bp.call(params=pp, stream=stream) == 0, _kvc.lib().kvc_last_error()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            bp.call(params=pp, stream=stream)
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e-3 / reps
    bp.run(stream)
    out = {"logits_kernel": timed(1), "softmax_pool_kernel": timed(6)}
    if tie_mode == "torch_cpu":
        g = timed(64)
        out["select_exact_kernel"] = max(timed(32) - g, 0.0)
        out["gather_kernel"] = g
    else:
        out["select_kernel (selection + fused gather)"] = timed(32)
    return out


def pmc_traffic(kernel_prefix, tag="r03", cfgname=None):
    """HBM bytes per launch of one kernel from the committed PMC passes (profiles/<tag>_pmc_batch_*.csv; separate
    rocprofv3 --pmc runs of tools/prof_driver.py; FETCH_SIZE doubled per the gfx950 correction).  None if absent."""
    import csv

    def one(name):
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_batch_{cfgname + '_' if cfgname else ''}{name}.csv")
        with open(path) as fh:
            return [float(r["mean_per_dispatch"]) for r in csv.DictReader(fh)
                    if r["kernel"].startswith(kernel_prefix) and r["counter"] == name][0]
    try:
        return (2.0 * one("FETCH_SIZE") + one("WRITE_SIZE")) * 1024.0
    except Exception:
        return None


def cpu_baseline(cfg, budget_s=12.0, dev=None):
    """The reference's CPU path on this host's cores, on a bounded sample of the same workload: the stock torch-CPU op
    sequence of pyramidkv_utils.py:317-346 (what the reference executes; `value`) and the oracle port
    (oracle/kvc_oracle.cpp, OpenMP over heads) as a side figure; parity counters of the HIP path vs the oracle on the same
    layer (H2O: one query head per KV head on the oracle side)."""
    from oracle import kvc_oracle as O      # cpu_baseline leg only (the checker, timed as the baseline)
    import math
    import torch.nn.functional as F
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))          # the GPU box gives one GPU's share of the host: 16 cores
    L, W = cfg["L"], cfg["W"]
    h2o = cfg["method"] == "h2o"
    q, k, v = synth.make_qkv(HQ, HKV, L, D, cfg["dtype"], 0)
    n_keep = layer_budgets(cfg)[0]
    out = {"unit": "tokens/s", "cores": threads, "kind": "port"}
    torch.set_num_threads(threads)

    def mask_block(aw):
        mask = torch.full((W, W), torch.finfo(aw.dtype).min)
        mc = torch.arange(W)
        mask.masked_fill_(mc < (mc + 1).view(W, 1), 0)
        aw[:, :, -W:, -W:] += mask[None, None]
    if not h2o:
        kx, vx = k.repeat_interleave(HQ // HKV, 1), v.repeat_interleave(HQ // HKV, 1)

        def ref_ops():
            aw = torch.matmul(q[..., -W:, :], kx.transpose(2, 3)) / math.sqrt(D)
            mask_block(aw)
            aw = F.softmax(aw, dim=-1, dtype=torch.float32).to(q.dtype)
            s = aw[:, :, -W:, :-W].sum(dim=-2)
            c = (F.max_pool1d if cfg["pooling"] == "maxpool" else F.avg_pool1d)(s, kernel_size=cfg["kernel"], padding=cfg["kernel"] // 2, stride=1)
            idx = c.topk(n_keep, dim=-1).indices.unsqueeze(-1).expand(-1, -1, -1, D)
            return (torch.cat([kx[:, :, :-W].gather(2, idx), kx[:, :, -W:]], 2), torch.cat([vx[:, :, :-W].gather(2, idx), vx[:, :, -W:]], 2))
        for _ in range(10):
            ref_ops()
        ts = []
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s or len(ts) < 5:
            t1 = time.perf_counter()
            ref_ops()
            ts.append(time.perf_counter() - t1)
        out["value"] = L / statistics.median(ts)
        out["sample"] = (f"{len(ts)} layer calls of the bench config (L={L}, k={n_keep}, all {HQ} heads, K/V repeat_kv-expanded as the "
                         f"reference passes them), stock torch-CPU op sequence of pyramidkv_utils.py:317-346, torch.set_num_threads({threads}), "
                         f"10 warm-ups, median")
        O.compress(q, k, v, W, n_keep, cfg["kernel"], cfg["pooling"], n_threads=threads)     # warm-up
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 4.0:
            O.compress(q, k, v, W, n_keep, cfg["kernel"], cfg["pooling"], n_threads=threads)
            n += 1
        out["oracle_port_tokens_per_s"] = n * L / (time.perf_counter() - t0)
    else:
        # H2O: the reference's op sequence needs ~12 GB and ~70 s per layer for 32 heads; sample = 2 query heads of one KV head
        qs, ks_ = q[:, :2], k[:, :1].repeat_interleave(2, 1)
        t0 = time.perf_counter()
        aw = torch.matmul(qs, ks_.transpose(2, 3)) / math.sqrt(D)
        mask_block(aw)
        aw = F.softmax(aw, dim=-1, dtype=torch.float32).to(q.dtype)
        s = aw[:, :, :, :-W].sum(dim=-2)
        s.topk(n_keep, dim=-1)
        dt = time.perf_counter() - t0
        out["value"] = L / (dt * HQ / 2)
        out["sample"] = (f"2 of {HQ} query heads of one layer call (L={L}), stock torch-CPU op sequence of pyramidkv_utils.py:544-561, "
                         f"{threads} threads, scaled to {HQ} heads")
    try:
        if dev is not None:
            qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
            par = {}
            pool = cfg["pooling"] if not h2o else None
            if h2o:
                g = HQ // HKV
                heads = [g * i + (i % g) for i in range(HKV)]
                qo = q[:, heads].contiguous()
                sc_h2o = O.scores(qo, k, W, cfg["kernel"], "avgpool", full_rows=True, dot_mode=O.DOT_CHAIN, sum_mode=O.SUM_TORCH16,
                                  n_threads=threads)                  # (0.07 Tflop on the host cores: once, for both tie modes)
            for tie, otie in (("canonical", O.TIES_CANON), ("torch_cpu", O.TIES_TORCH)):
                got = _kvc.compress(METHODS[cfg["method"]], qd, kd, vd, W, n_keep, cfg["kernel"], pool, tie,
                                    return_indices=True, return_scores=True)
                gi, gs = got[2][0].cpu(), got[3][0].cpu()
                if h2o:
                    sc = sc_h2o
                    idx, _ = O.topk(sc, n_keep, otie, n_threads=threads)
                    ko = O.gather(k, idx, W, HKV)
                    gi, gs, gk = gi[heads], gs[heads], got[0][:, heads].cpu()
                    kv_equal = bool(torch.equal(gk, ko))
                else:
                    ko, vo, idx, sc = O.compress(q, k, v, W, n_keep, cfg["kernel"], cfg["pooling"], dot_mode=O.DOT_CHAIN,
                                                 sum_mode=O.SUM_TORCH16, tie_mode=otie, n_threads=threads)
                    kv_equal = bool(torch.equal(got[0].cpu(), ko) and torch.equal(got[1].cpu(), vo))
                par[tie] = {"score_bits_differing": int((gs.view(torch.int16) != sc.view(torch.int16)).sum()),
                            "scores_compared": int(sc.numel()),
                            "heads_with_identical_indices": int((gi == idx).all(dim=1).sum()), "heads": int(idx.shape[0]),
                            "k_out_v_out_bytes_equal": kv_equal}
            out["parity_vs_oracle_same_layer"] = par
    except Exception as e:  # pragma: no cover
        out["parity_error"] = repr(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--tie-mode", default="torch_cpu", choices=["canonical", "torch_cpu"],
                    help="torch_cpu (default, the clusters' default): indices bit-identical to the reference's CPU run, "
                         "ties included; canonical: value desc / index asc, faster, differs only inside tie groups")
    ap.add_argument("--mode", default="batch", choices=["batch", "calls"],
                    help="batch: one kvc_compress_batch call per prompt (all 32 layers per kernel launch); "
                         "calls: 32 kvc_compress calls per prompt spread over --streams streams")
    ap.add_argument("--in-flight", type=int, default=1, choices=[1, 2, 3, 4],
                    help="prompts in flight: 2 alternates consecutive prompts between two HIP streams (a serving stack's overlap: "
                         "the latency-bound exact top-k of one prompt runs beside the K scan of the next)")
    ap.add_argument("--prompts", type=int, default=0, help="prompts per step (0: the config's default, sized for ~20 ms steps)")
    ap.add_argument("--sets", type=int, default=0, help="distinct resident prompts the step cycles over (0: the config's default, 2; "
                                                          "1 at 32k, where one prompt's K is 2 GB and generating it takes a while)")
    ap.add_argument("--streams", type=int, default=16, help="calls mode: HIP streams the 32 independent layer calls are spread over")
    ap.add_argument("--dot-mode", default="exact", choices=["exact", "mfma16"],
                    help="exact: f32-MFMA fmaf chain (bit-identical to the oracle); mfma16: packed bf16 MFMA scan (tolerance mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    cfg = CONFIGS[a.config]
    _kvc.DOT_MODE = a.dot_mode
    global LAYERS
    LAYERS = cfg.get("layers", LAYERS)
    h2o = cfg["method"] == "h2o"
    if h2o:
        a.mode, a.in_flight, a.streams = "calls", 1, 1       # compute-bound single calls
    per_step = a.prompts or cfg["prompts"]
    ks = layer_budgets(cfg)
    n_sets = max(a.sets or cfg.get("sets", 2), a.in_flight)
    inputs = [make_inputs(cfg, dev, rank_seed(rank) + 100 * s) for s in range(n_sets)]
    fl = side_streams(dev, a.in_flight) if a.in_flight > 1 else None
    prompts = [Prompt(cfg, dev, a.tie_mode, a.mode, inputs[s], ks, a.streams if a.mode == "calls" else 1,
                      stream=fl[s % len(fl)] if fl else None) for s in range(n_sets)]
    used = list(fl or ()) + (prompts[0].streams if a.mode == "calls" and a.streams > 1 else [])
    for p in prompts:
        p.run()                                            # first call outside the timed region (one-time attribute setup)
    torch.cuda.synchronize(dev)

    dt, per = time_steps(prompts, per_step, a.steps, a.warmup, dev, dist, used)
    tokens_per_step = per_step * cfg["L"] * LAYERS
    value = job_value(world, a.steps, tokens_per_step, dt)
    med_ms = statistics.median(per)
    t_layer = dt / a.steps / per_step / LAYERS

    out = {
        "metric": "KV tokens compressed/sec", "value": value, "unit": "tokens/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": cfg["desc"], "name": a.config, "layers_per_prompt": LAYERS, "prompts_per_step": per_step,
                   "resident_prompt_sets": n_sets, "q_len": cfg["L"], "budget": cfg["cap"],
                   "kv_layout": "gqa_native [1,8,L,128] as the patched attention forward hands K/V over",
                   "tie_mode": a.tie_mode, "dot_mode": a.dot_mode, "mode": a.mode, "prompts_in_flight": a.in_flight,
                   "streams": a.streams if a.mode == "calls" else a.in_flight,
                   "launch": ("one kvc_compress_batch call per prompt: every kernel launched once for the 32 layers" if a.mode == "batch"
                              else "32 host-enqueued kvc_compress calls per prompt"),
                   "multi_gpu": "replicas, no collective"},
        "median_ms_per_step": med_ms, "value_from_median_step": world * tokens_per_step / (med_ms * 1e-3),
        "per_layer_us": t_layer * 1e6, "us_per_prompt": dt / a.steps / per_step * 1e6,
    }
    if rank == 0:
        kmean = sum(ks) / len(ks)
        scan_b, path_b, sel_b = algorithmic_bytes(cfg["L"], cfg["W"], kmean)
        out["path_algorithmic_bytes_per_layer"] = path_b
        out["path_achieved_GBs"] = path_b / t_layer / 1e9
        out["path_frac"] = path_b / t_layer / 1e9 / HBM_PEAK_GBS
        if h2o:
            flops = 2.0 * HQ * cfg["L"] * cfg["L"] * D       # one QK^T per layer
            exact = a.dot_mode == "exact"

            def pmc_c3(prefixes):
                vals = [pmc_traffic(p_, tag="r03", cfgname="c3") for p_ in prefixes]
                return None if any(v is None for v in vals) else sum(vals)
            out["roofline"] = {"bound": "mfma", "kernel": "h2o_fused_kernel (logits, torch-order softmax sums and column sums in one kernel)"
                               if exact else "h2o_fast_stats_kernel + h2o_fast_colsum_kernel",
                               "achieved": flops / t_layer / 1e12, "peak": MFMA_F32_PEAK_TFLOPS if exact else MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": flops / t_layer / 1e12 / (MFMA_F32_PEAK_TFLOPS if exact else MFMA_PEAK_TFLOPS),
                               "frac_of_bf16_mfma_peak": flops / t_layer / 1e12 / MFMA_PEAK_TFLOPS,
                               "traffic": pmc_c3(["kvc::h2o_fused_kernel", "kvc::h2o_kperm_kernel", "kvc::h2o_colcomb_kernel"]) if exact else None,
                               "traffic_source": "profiles/r03_pmc_batch_c3_{FETCH,WRITE}_SIZE.csv: permute + fused + combine kernels of one layer "
                                                 "(FETCH_SIZE x2 gfx950 correction); algorithmic inputs Q + K = 82 MB",
                               "launch_us": t_layer * 1e6, "algorithmic_flops_per_layer": flops,
                               "note": "exact mode: scored against the f32-input MFMA it computes on (MI355X_MICROARCH.md: Peak FP32 (matrix) "
                                       "157.3 TFLOP/s), which shares its SIMD's issue time with VALU work (DESIGN.md section 4); "
                                       "frac_of_bf16_mfma_peak is the same rate against the 2.5 PFLOP/s bf16 peak round 2 was scored on"}
            if exact:
                out["roofline"].update({"mfma_f32_TFLOPs": flops / t_layer / 1e12, "mfma_f32_peak_TFLOPs": MFMA_F32_PEAK_TFLOPS,
                                        "mfma_f32_frac": flops / t_layer / 1e12 / MFMA_F32_PEAK_TFLOPS})
        elif a.mode == "batch":
            kt = kernel_breakdown(prompts[0], dev, a.tie_mode)
            tot = sum(kt.values())
            out["kernels_us_per_32_layer_launch"] = {k_: round(v_ * 1e6, 1) for k_, v_ in kt.items()}
            out["kernel_time_share"] = {k_: round(v_ / tot, 3) for k_, v_ in kt.items()}
            dom = max(kt, key=kt.get)
            t_scan = kt["logits_kernel"]
            kscan = {"kscan_kernel": "logits_kernel (K scan + window QK^T)", "kscan_launch_us": t_scan * 1e6,
                     "kscan_algorithmic_bytes_per_launch": scan_b * LAYERS, "kscan_achieved_GBs": scan_b * LAYERS / t_scan / 1e9,
                     "kscan_frac": scan_b * LAYERS / t_scan / 1e9 / HBM_PEAK_GBS,
                     "kscan_traffic": pmc_traffic("kvc::logits_kernel"),
                     "kscan_traffic_source": "profiles/r03_pmc_batch_{FETCH,WRITE}_SIZE.csv (separate rocprofv3 --pmc passes of "
                                             "tools/prof_driver.py; FETCH_SIZE x2 gfx950 correction; includes the logits it writes)"}
            if a.dot_mode == "exact":
                # the exact dot mode computes the d-ascending fp32 fmaf chain on the f32-input MFMA, which runs at the fp32 VALU
                # rate: with 32 FLOP per K byte that pipe, not HBM, bounds the K scan — both fractions reported
                fl = 2.0 * HQ * cfg["W"] * cfg["L"] * D * LAYERS
                kscan.update({"kscan_mfma_f32_TFLOPs": fl / t_scan / 1e12, "kscan_mfma_f32_peak_TFLOPs": MFMA_F32_PEAK_TFLOPS,
                              "kscan_mfma_f32_frac": fl / t_scan / 1e12 / MFMA_F32_PEAK_TFLOPS,
                              "kscan_binding_limit": "f32-input MFMA (exact fmaf-chain dot products); HBM in dot_mode mfma16"})
            if dom == "logits_kernel" and a.dot_mode == "exact":
                # "bound: hbm" is wrong for the exact mode — the K scan is bound by the f32-input MFMA (the guide's
                # "Peak FP32 (matrix)" 157.3 TF/s) that computes its fmaf chains; the HBM view stays in the kscan_* fields
                out["roofline"] = {"bound": "mfma", "kernel": kscan["kscan_kernel"], "achieved": kscan["kscan_mfma_f32_TFLOPs"],
                                   "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": kscan["kscan_mfma_f32_frac"],
                                   "traffic": kscan["kscan_traffic"], "launch_us": t_scan * 1e6,
                                   "algorithmic_flops_per_launch": 2.0 * HQ * cfg["W"] * cfg["L"] * D * LAYERS,
                                   "algorithmic_bytes_per_launch": scan_b * LAYERS,
                                   "units_per_launch": f"{cfg['L'] * LAYERS} tokens x {scan_b / cfg['L']:.0f} B/token x 32 flop/B",
                                   "peak_note": "f32-input MFMA (MI355X_MICROARCH.md: Peak FP32 (matrix) 157.3 TFLOP/s): the exact dot mode's "
                                                "d-ascending fmaf chains; as an HBM scan the same launch is kscan_frac of 8 TB/s",
                                   "time_share": kt[dom] / tot}
            elif dom == "logits_kernel":
                out["roofline"] = {"bound": "hbm", "kernel": kscan["kscan_kernel"], "achieved": kscan["kscan_achieved_GBs"],
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kscan["kscan_frac"], "traffic": kscan["kscan_traffic"],
                                   "launch_us": t_scan * 1e6, "algorithmic_bytes_per_launch": scan_b * LAYERS,
                                   "units_per_launch": f"{cfg['L'] * LAYERS} tokens x {scan_b / cfg['L']:.0f} B/token",
                                   "time_share": kt[dom] / tot}
            else:
                t_dom = kt[dom]
                out["roofline"] = {"bound": "latency" if dom.startswith("select_exact") else "hbm", "kernel": dom, "achieved": sel_b * LAYERS / t_dom / 1e9, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": sel_b * LAYERS / t_dom / 1e9 / HBM_PEAK_GBS,
                                   "traffic": pmc_traffic("kvc::select"), "launch_us": t_dom * 1e6,
                                   "algorithmic_bytes_per_launch": sel_b * LAYERS, "time_share": t_dom / tot,
                                   "note": "latency-bound: one wave per head walks libstdc++'s heap / partition moves (a single dependent "
                                           "chain per head, DESIGN.md §3); its bytes are the pooled scores it reads and the indices it writes"}
            out["roofline"].update(kscan)
        else:
            out["roofline"] = {"bound": "hbm", "kernel": "whole per-call path (latency-bound launches)", "achieved": path_b / t_layer / 1e9,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": path_b / t_layer / 1e9 / HBM_PEAK_GBS, "traffic": None}
        if not a.no_extras and world == 1 and not h2o:
            extra = {}
            half = max(5, a.steps // 5)

            def quick(ps, streams=None, per=max(1, per_step // 4)):
                d, _ = time_steps(ps, per, half, 2, dev, None, streams)
                return half * per * cfg["L"] * LAYERS / d
            other = "torch_cpu" if a.tie_mode == "canonical" else "canonical"
            try:
                extra[f"tokens_per_s_tie_mode_{other}"] = quick([Prompt(cfg, dev, other, "batch", inputs[s], ks) for s in range(n_sets)])
            except Exception as e:
                extra[f"tie_mode_{other}_error"] = str(e)
            if a.in_flight == 1 and a.mode == "batch":
                for n_fl, word in ((2, "two"), (3, "three")):
                    try:    # several prompts in flight: own stream, outputs and workspace each; outputs compared with the single-stream run
                        sn = side_streams(dev, n_fl)
                        xin = [inputs[s % n_sets] for s in range(max(n_sets, n_fl))]
                        pf = [Prompt(cfg, dev, a.tie_mode, "batch", xin[s], ks, stream=sn[s % n_fl]) for s in range(len(xin))]
                        extra[f"tokens_per_s_{word}_prompts_in_flight"] = quick(pf, sn)
                        torch.cuda.synchronize(dev)
                        same = all(torch.equal(x, y) for i, pa in enumerate(pf) for x, y in
                                   zip(pa.bp.k_out + pa.bp.v_out + pa.bp.idx,
                                       prompts[i % n_sets].bp.k_out + prompts[i % n_sets].bp.v_out + prompts[i % n_sets].bp.idx))
                        extra[f"{word}_prompts_in_flight_outputs_identical_to_single_stream"] = bool(same)
                        del pf
                    except Exception as e:
                        extra[f"{word}_in_flight_error"] = str(e)
            if a.mode == "batch" and a.dot_mode == "exact":
                try:
                    extra["tokens_per_s_dot_mode_mfma16_tolerance_mode"] = quick(
                        [Prompt(cfg, dev, a.tie_mode, "batch", inputs[s], ks, dot_mode="mfma16") for s in range(n_sets)])
                except Exception as e:
                    extra["dot_mode_mfma16_error"] = str(e)
            try:
                pc = [Prompt(cfg, dev, a.tie_mode, "calls", inputs[0], ks, a.streams)]
                extra[f"tokens_per_s_32_kvc_compress_calls_{a.streams}_streams"] = quick(pc, pc[0].streams, per=2)
                pc1 = [Prompt(cfg, dev, a.tie_mode, "calls", inputs[0], ks, 1)]
                extra["tokens_per_s_32_kvc_compress_calls_single_stream"] = quick(pc1, per=1)
                del pc, pc1
            except Exception as e:
                extra["calls_error"] = str(e)
            try:        # the literal reference-shaped call: repeat_kv-expanded K/V [1,32,L,128]
                xin = make_inputs(cfg, dev, rank_seed(rank), expanded=True)
                extra["tokens_per_s_batch_expanded_kv_as_reference_passes"] = quick([Prompt(cfg, dev, a.tie_mode, "batch", xin, ks)])
                del xin
            except Exception as e:
                extra["expanded_kv_error"] = str(e)
            try:        # what the patched forward pays per prompt on the host side as well: PrefillBatch.add x 32 (window-Q copies,
                #         layout keys) + grouped flushes (plan building, output allocation into the cache layers, library calls)
                from kvcache_factory_amd import pyramidkv_utils as pu
                from kvcache_factory_amd.cache import CompressedDynamicLayer
                cl = {"snapkv": pu.SnapKVCluster, "pyramidkv": pu.PyramidKVCluster}[cfg["method"]]
                kw = dict(window_size=cfg["W"], max_capacity_prompt=cfg["cap"], kernel_size=cfg["kernel"], pooling=cfg["pooling"])
                clusters = [cl(num_hidden_layers=LAYERS, layer_idx=l, **kw) if cfg["method"] == "pyramidkv" else cl(**kw) for l in range(LAYERS)]
                old = pu.TIE_MODE
                pu.TIE_MODE = a.tie_mode

                def host_prompt(qkv):
                    pb = pu.PrefillBatch()
                    layers = [CompressedDynamicLayer() for _ in range(LAYERS)]
                    for l, (q, k, v) in enumerate(qkv):
                        pb.add(clusters[l], k, q, v, lambda kc, vc, layer=layers[l]: layer.prefill(kc, vc, cfg["L"]), tag=layers[l],
                               alloc=layers[l].reserve)
                        if len(pb) >= pu.GROUP_LAYERS > 0:
                            pb.flush(overlap=True)
                    if len(pb):
                        pb.flush(overlap=True)
                    pb.settle()
                    return layers
                for _ in range(3):
                    host_prompt(inputs[0])
                torch.cuda.synchronize(dev)
                n_h = max(10, per_step // 2)
                t0 = time.perf_counter()
                for i in range(n_h):
                    host_prompt(inputs[i % n_sets])
                torch.cuda.synchronize(dev)
                extra["tokens_per_s_through_PrefillBatch_host_path"] = n_h * cfg["L"] * LAYERS / (time.perf_counter() - t0)
                extra["PrefillBatch_group_layers"] = pu.GROUP_LAYERS
                pu.TIE_MODE = old
            except Exception as e:
                extra["host_path_error"] = repr(e)
            try:        # SURVEY 8f N4 on one layer of this configuration (HIP events around 20 calls each; us per layer call)
                q0, k0, v0 = inputs[0][0]
                W0, keep0 = cfg["W"], ks[0]

                def per_call_us(fn, reps=20):
                    for _ in range(3):
                        fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        fn()
                    e1.record()
                    torch.cuda.synchronize(dev)
                    return e0.elapsed_time(e1) * 1e3 / reps
                sc0 = _kvc.scores(METHODS[cfg["method"]], q0, k0, W0, cfg["kernel"], cfg["pooling"])
                idx0 = _kvc.select(sc0, keep0, a.tie_mode)
                kc0 = _kvc.compress(METHODS[cfg["method"]], q0, k0, v0, W0, keep0, cfg["kernel"], cfg["pooling"], a.tie_mode)[0]
                extra["n4_us_per_layer_call"] = {
                    "l2norm_compress_rows_%d" % cfg["cap"]: per_call_us(lambda: _kvc.l2norm_compress(k0, v0, cfg["cap"], n_q_heads=HQ)),
                    "merge_pivot_after_scores_and_select": per_call_us(lambda: _kvc.merge_pivot(METHODS[cfg["method"]], k0, v0, idx0, W0)),
                    "think_prune_recent32_ratio0.4": per_call_us(lambda: _kvc.think_prune(kc0, q0, min(32, kc0.shape[2]), 0.4)),
                    "note": "Python binding included (output allocation, ctypes call); kernel times in profiles/r02_kernel_stats_n4_*.csv"}
            except Exception as e:
                extra["n4_error"] = repr(e)
            out["extras"] = extra
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, dev=dev)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
