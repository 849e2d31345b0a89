#!/usr/bin/env python3
"""bench.py — KV tokens compressed/sec for the post-prefill scoring+eviction hot path on MI355X.

Metric (BASELINE.json): "KV tokens compressed/sec + selection-index exact-match, Llama-3-8B 8k->128".
One "step" = the compression of ONE 8k-token prompt = 32 `update_kv` calls (one per layer of Llama-3-8B, each on
that layer's own synthetic Q/K/V, already resident in HBM), enqueued back to back through the C-ABI
(kvc_compress) exactly as the patched attention forward does.  tokens/step = q_len * 32 layers.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c2_w32|c4|c5] [--tie-mode canonical|torch_cpu]

N > 1 (torchrun, one rank per GPU): the path shards by independent prompts/layers with no exchange, so every rank
runs its own replica of the workload (weak scaling); time = max over ranks, value = N * tokens / time.
Prints ONE JSON line (rank 0).  See DESIGN.md §Measurement for the byte accounting.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from kvcache_factory_amd import _kvc, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured float4-copy rate

CONFIGS = {
    # name: (method, q_len, cap, window, kernel, pooling, dtype, per-layer-k?)
    "c2": dict(method="snapkv", L=8000, cap=128, W=8, kernel=7, pooling="maxpool", dtype=torch.bfloat16,
               desc="SnapKV Llama-3-8B shapes (Hq=32,Hkv=8,D=128), seq_len=8000 -> max_capacity_prompt=128, bf16, W=8, maxpool7"),
    "c2_w32": dict(method="snapkv", L=8000, cap=128, W=32, kernel=7, pooling="maxpool", dtype=torch.bfloat16,
                   desc="SnapKV 8k->128, W=32 (needle-runner window)"),
    "c4": dict(method="pyramidkv", L=8000, cap=128, W=8, kernel=7, pooling="maxpool", dtype=torch.bfloat16,
               desc="PyramidKV 8k, total budget 128x32 (k_l = 234..17)"),
    "c5": dict(method="pyramidkv", L=32000, cap=2048, W=8, kernel=7, pooling="maxpool", dtype=torch.bfloat16,
               desc="PyramidKV Mistral-7B shapes, 32k -> 2048 (k_l = 3978..103)"),
}
CONFIGS["c3"] = dict(method="h2o", L=8000, cap=128, W=8, kernel=7, pooling=None, dtype=torch.bfloat16, layers=4,
                     desc="H2O heavy-hitter scoring, Llama-3-8B shapes, seq_len=8000 -> 128, bf16 (all 8000 query rows score; "
                          "4 layers per step: each layer materialises a 4.1 GB logit matrix)")
HQ, HKV, D, LAYERS = 32, 8, 128, 32
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak (MI355X_MICROARCH.md); the H2O kernel runs the exact f32-input MFMA (157 TF peak)
METHODS = {"snapkv": _kvc.SNAPKV, "pyramidkv": _kvc.PYRAMIDKV, "h2o": _kvc.H2O}


def algorithmic_bytes(L, W, k, es=2):
    """GQA-aware bytes one layer call must move (K scan once, V never scanned)."""
    scan = HKV * L * D * es + HQ * W * D * es                       # K scan + Q window  (the K-scan kernel's share)
    compact = 2 * HQ * k * D * es + 2 * HKV * W * D * es + 2 * HQ * (k + W) * D * es + 8 * HQ * k
    return scan, scan + compact


def layer_budgets(cfg):
    if cfg["method"] == "pyramidkv":
        return [_kvc.pyramid_k(cfg["cap"], cfg["W"], cfg["L"], l, LAYERS) for l in range(LAYERS)]
    return [cfg["cap"] - cfg["W"]] * LAYERS


_STREAMS = {}


def side_streams(dev, n):
    """n HIP streams per device.  The 32 layer compressions of a prompt are independent, so the
    drop-in calls are spread round-robin over a few streams exactly as a serving stack would overlap them."""
    key = (dev.index, n)
    if key not in _STREAMS:
        _STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _STREAMS[key]


def build_plans(cfg, dev, tie_mode, expanded, n_streams=1, seed0=0):
    plans = []
    ks = layer_budgets(cfg)
    streams = side_streams(dev, n_streams)
    for l in range(LAYERS):
        q, k, v = synth.make_qkv(HQ, HKV, cfg["L"], D, cfg["dtype"], seed0 + l, expanded=expanded, device=dev)
        if expanded:
            k, v = k.contiguous(), v.contiguous()
        with torch.cuda.stream(streams[l % n_streams]):          # the plan's workspace belongs to its stream
            plan = _kvc.CompressPlan(METHODS[cfg["method"]], q, k, v, cfg["W"], ks[l], cfg["kernel"], cfg["pooling"],
                                     tie_mode, want_indices=True)
        plan.stream_handle = ctypes.c_void_p(streams[l % n_streams].cuda_stream)
        plans.append(plan)
    torch.cuda.synchronize(dev)
    return plans, ks


class BatchStep:
    """The same 32 layer compressions through kvc_compress_batch: one library call, each kernel launched once."""

    def __init__(self, cfg, dev, tie_mode, plans, ks):
        self.bp = _kvc.BatchPlan(METHODS[cfg["method"]], [(p.q, p.k, p.v) for p in plans], cfg["W"], ks, cfg["kernel"],
                                 cfg["pooling"], tie_mode, want_indices=True)
        self.stream_handle = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def run(self, stream=None):
        self.bp.run(stream if stream is not None else self.stream_handle)


def run_step(plans, stream=None):
    for p in plans:
        p.run(stream if stream is not None else p.stream_handle)


def capture_step(plans, dev):
    """One step (32 layer calls over their streams) as a HIP graph: fork from the capture stream, run, join.
    Replaying it costs one host call per step instead of ~130 kernel launches."""
    handles = {p.stream_handle.value for p in plans}
    tstreams = [s for sl in _STREAMS.values() for s in sl if s.cuda_stream in handles and s.device == dev]
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream(device=dev)
    with torch.cuda.graph(g, stream=cap):
        fork = torch.cuda.Event()
        fork.record(cap)
        for s in tstreams:
            s.wait_event(fork)
        run_step(plans)
        for s in tstreams:
            e = torch.cuda.Event()
            e.record(s)
            cap.wait_event(e)
    return g


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


def time_steps(plans, steps, warmup, dev, dist, graph=None):
    if graph is not None:
        for _ in range(warmup):
            graph.replay()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            graph.replay()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        return reduce_max_time(time.perf_counter() - t0, dist, dev)
    for _ in range(warmup):
        run_step(plans)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        run_step(plans)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    return reduce_max_time(time.perf_counter() - t0, dist, dev)


def time_scan_kernel(plans, dev, reps=20):
    """Average duration of ONE launch of the dominant kernel (the K-scan / window-logits kernel), measured live with
    HIP events on the launch stream: kvc_scores with debug_stage_mask=1 enqueues only that kernel."""
    lib = _kvc.lib()
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    calls = []
    for p in plans:
        pp = _kvc.Params.from_buffer_copy(p.p)
        pp.debug_stage_mask = 1
        sc = torch.empty(1, HQ, p.k.shape[2] - p.p.window, dtype=p.k.dtype, device=dev)
        calls.append((pp, p, sc))

    def sweep():
        for pp, p, sc in calls:
            rc = lib.kvc_scores(ctypes.byref(pp), _kvc._ptr(p.q), _kvc._ptr(p.k), _kvc._ptr(sc), _kvc._ptr(p.ws), p.nbytes, stream)
            assert rc == 0, lib.kvc_last_error()
    sweep()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sweep()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) * 1e-3 / (reps * len(calls))


def time_scan_kernel_batch(bstep, dev, reps=10):
    """Same for the batched call: ONE logits_kernel launch covers all 32 layers (debug_stage_mask=1 on kvc_compress_batch)."""
    lib, bp = _kvc.lib(), bstep.bp
    pp = _kvc.Params.from_buffer_copy(bp.p)
    pp.debug_stage_mask = 1
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def launch():
        rc = lib.kvc_compress_batch(ctypes.byref(pp), bp.n, bp._keep, bp._q, bp._k, bp._v, bp._ko, bp._vo, bp._ix, None,
                                    _kvc._ptr(bp.ws), bp.nbytes, stream)
        assert rc == 0, lib.kvc_last_error()
    launch()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) * 1e-3 / reps


def cpu_baseline(cfg, budget_s=12.0, dev=None):
    """The CPU oracle (a port of the reference's algorithm, oracle/kvc_oracle.cpp) timed on this host's cores on a
    bounded sample of the same workload: whole layer calls of the bench config, repeated for ~budget_s seconds."""
    from oracle import kvc_oracle as O      # cpu_baseline leg only (the checker, timed as the baseline)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))          # the GPU box gives one GPU's share of the host: 16 cores
    L = cfg["L"]
    q, k, v = synth.make_qkv(HQ, HKV, L, D, cfg["dtype"], 0)
    n_keep = layer_budgets(cfg)[0]
    O.compress(q, k, v, cfg["W"], n_keep, cfg["kernel"], cfg["pooling"], n_threads=threads)     # warm-up
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        O.compress(q, k, v, cfg["W"], n_keep, cfg["kernel"], cfg["pooling"], n_threads=threads)
        n += 1
    dt = time.perf_counter() - t0
    out = {"value": n * L / dt, "unit": "tokens/s", "cores": threads, "kind": "port",
           "sample": f"{n} layer calls of the bench config (L={L}, k={n_keep}, all {HQ} heads), oracle/kvc_oracle.cpp, OpenMP over heads"}
    # the stock torch-CPU op sequence the reference executes (pyramidkv_utils.py:317-346), same inputs, same host
    try:
        import math
        import torch.nn.functional as F
        torch.set_num_threads(threads)
        kx, vx = k.repeat_interleave(HQ // HKV, 1), v.repeat_interleave(HQ // HKV, 1)
        W = cfg["W"]

        def ref_ops():
            aw = torch.matmul(q[..., -W:, :], kx.transpose(2, 3)) / math.sqrt(D)
            mask = torch.full((W, W), torch.finfo(aw.dtype).min)
            mc = torch.arange(W)
            mask.masked_fill_(mc < (mc + 1).view(W, 1), 0)
            aw[:, :, -W:, -W:] += mask[None, None]
            aw = F.softmax(aw, dim=-1, dtype=torch.float32).to(q.dtype)
            s = aw[:, :, -W:, :-W].sum(dim=-2)
            c = (F.max_pool1d if cfg["pooling"] == "maxpool" else F.avg_pool1d)(s, kernel_size=cfg["kernel"], padding=cfg["kernel"] // 2, stride=1)
            idx = c.topk(n_keep, dim=-1).indices.unsqueeze(-1).expand(-1, -1, -1, D)
            return (torch.cat([kx[:, :, :-W].gather(2, idx), kx[:, :, -W:]], 2), torch.cat([vx[:, :, :-W].gather(2, idx), vx[:, :, -W:]], 2))
        for _ in range(3):
            ref_ops()
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < min(budget_s, 6.0):
            ref_ops()
            n += 1
        out["torch_cpu_ops_tokens_per_s"] = n * L / (time.perf_counter() - t0)
        out["torch_cpu_ops_threads"] = threads
    except Exception as e:  # pragma: no cover
        out["torch_cpu_ops_error"] = repr(e)
    # parity counters on the same layer: the HIP path against the oracle's product arithmetic, both tie modes
    try:
        if dev is not None and cfg["method"] != "h2o":
            qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
            par = {}
            for tie, otie in (("canonical", O.TIES_CANON), ("torch_cpu", O.TIES_TORCH)):
                ko, vo, idx, sc = O.compress(q, k, v, cfg["W"], n_keep, cfg["kernel"], cfg["pooling"], dot_mode=O.DOT_CHAIN,
                                             sum_mode=O.SUM_TORCH16, tie_mode=otie, n_threads=threads)
                g = _kvc.compress(METHODS[cfg["method"]], qd, kd, vd, cfg["W"], n_keep, cfg["kernel"], cfg["pooling"], tie,
                                  return_indices=True, return_scores=True)
                gi, gs = g[2][0].cpu(), g[3][0].cpu()
                par[tie] = {"score_bits_differing": int((gs.view(torch.int16) != sc.view(torch.int16)).sum()),
                            "scores_compared": int(sc.numel()),
                            "heads_with_identical_indices": int((gi == idx).all(dim=1).sum()), "heads": int(idx.shape[0]),
                            "k_out_v_out_bytes_equal": bool(torch.equal(g[0].cpu(), ko) and torch.equal(g[1].cpu(), vo))}
            out["parity_vs_oracle_same_layer"] = par
    except Exception as e:  # pragma: no cover
        out["parity_error"] = repr(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--tie-mode", default="torch_cpu", choices=["canonical", "torch_cpu"],
                    help="torch_cpu (default, the clusters' default): indices bit-identical to the reference's CPU run, "
                         "ties included; canonical: value desc / index asc, faster, differs only inside tie groups")
    ap.add_argument("--mode", default="batch", choices=["batch", "calls"],
                    help="batch: one kvc_compress_batch call per step (all 32 layers per kernel launch); "
                         "calls: 32 kvc_compress calls per step spread over --streams streams")
    ap.add_argument("--streams", type=int, default=16, help="HIP streams the 32 independent layer calls are spread over")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every kvc_compress call from the host instead of replaying a HIP graph of the step")
    ap.add_argument("--dot-mode", default="exact", choices=["exact", "mfma16"],
                    help="exact: f32-MFMA fmaf chain (bit-identical to the oracle); mfma16: packed bf16 MFMA scan (tolerance mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (expanded K/V, exact ties)")
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
    _kvc.DOT_MODE = a.dot_mode                 # default dot_mode of every plan built below
    global LAYERS
    LAYERS = cfg.get("layers", LAYERS)
    if cfg["method"] == "h2o":
        a.mode, a.no_extras = "calls", True            # one 4.1 GB logit matrix at a time: sequential calls, one stream
        a.streams = 1

    plans, ks = build_plans(cfg, dev, a.tie_mode, expanded=False, n_streams=a.streams, seed0=rank_seed(rank))
    run_step(plans)                      # first call outside any capture (one-time LDS attribute setup)
    torch.cuda.synchronize(dev)
    graph, launch_mode = None, "host-enqueued kvc_compress calls"
    call_plans = plans
    if a.mode == "batch":
        plans = [BatchStep(cfg, dev, a.tie_mode, call_plans, ks)]
        run_step(plans)
        torch.cuda.synchronize(dev)
        launch_mode = "one kvc_compress_batch call per step: every kernel launched once for the 32 layers"
    elif not a.no_graph:
        try:
            graph = capture_step(plans, dev)
            launch_mode = "HIP graph of the step (32 kvc_compress calls captured once, replayed per step)"
        except Exception as e:  # pragma: no cover
            launch_mode = f"host-enqueued (graph capture failed: {e})"
            graph = None
    dt = time_steps(plans, a.steps, a.warmup, dev, dist, graph)
    tokens_per_step = cfg["L"] * LAYERS
    value = job_value(world, a.steps, tokens_per_step, dt)

    out = {
        "metric": "KV tokens compressed/sec", "value": value, "unit": "tokens/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": cfg["desc"], "name": a.config, "layers_per_step": LAYERS, "q_len": cfg["L"], "budget": cfg["cap"],
                   "kv_layout": "gqa_native [1,8,L,128] as the patched attention forward hands K/V over",
                   "tie_mode": a.tie_mode, "dot_mode": a.dot_mode, "mode": a.mode, "streams": a.streams if a.mode == "calls" else 1, "launch": launch_mode, "per_layer_us": dt / a.steps / LAYERS * 1e6,
                   "multi_gpu": "replicas, no collective"},
    }
    if rank == 0:
        # ---- roofline of the dominant kernel (K scan), live HIP-event timing ----
        es = 2
        scan_b, path_b = algorithmic_bytes(cfg["L"], cfg["W"], sum(ks) / len(ks), es)
        if cfg["method"] == "h2o":
            t_scan = None
        elif a.mode == "batch":
            t_scan = time_scan_kernel_batch(plans[0], dev)        # one launch = 32 layers
            scan_b *= LAYERS
        else:
            t_scan = time_scan_kernel(call_plans, dev)
        if cfg["method"] == "h2o":
            # F = 2*Hq*L*L*D flop per layer (one QK^T); achieved over the whole per-layer time
            flops = 2.0 * HQ * cfg["L"] * cfg["L"] * D
            t_layer = dt / a.steps / LAYERS
            out["roofline"] = {"bound": "mfma", "kernel": "h2o_logits_kernel (+ row/column softmax sums)", "achieved": flops / t_layer / 1e12,
                               "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / t_layer / 1e12 / MFMA_PEAK_TFLOPS, "traffic": None,
                               "launch_us": t_layer * 1e6, "algorithmic_flops_per_layer": flops,
                               "note": "exact-arithmetic version: f32-input MFMA (157 TF peak), logits materialised once"}
            t_scan = 1.0
        out.setdefault("roofline", {"bound": "hbm", "kernel": "logits_kernel (K scan + window QK^T)", "achieved": scan_b / t_scan / 1e9,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": scan_b / t_scan / 1e9 / HBM_PEAK_GBS, "traffic": None,
                           "launch_us": t_scan * 1e6, "algorithmic_bytes_per_launch": scan_b,
                           "units_per_launch": f"{cfg['L'] * (LAYERS if a.mode == 'batch' else 1)} tokens x 2056 B/token",
                           "path_achieved_GBs": path_b / (dt / a.steps / LAYERS) / 1e9,
                           "path_frac": path_b / (dt / a.steps / LAYERS) / 1e9 / HBM_PEAK_GBS,
                           "path_algorithmic_bytes_per_layer": path_b})
        try:      # HBM bytes of the K-scan kernel from the committed PMC pass (profiles/, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE;
            #   gfx950 correction: FETCH_SIZE counts half of a wide coalesced read).  KB per dispatch -> bytes per launch.
            import csv
            def _pmc(name):
                with open(os.path.join(ROOT, "profiles", f"r01_pmc_{'batch_' if a.mode == 'batch' else ''}{name}.csv")) as fh:
                    return [float(r["mean_per_dispatch"]) for r in csv.DictReader(fh)
                            if r["kernel"].startswith("kvc::logits_kernel") and r["counter"] == name][0]
            if a.config == "c2":
                out["roofline"]["traffic"] = (2.0 * _pmc("FETCH_SIZE") + _pmc("WRITE_SIZE")) * 1024.0
                out["roofline"]["traffic_source"] = ("profiles/r01_pmc_[batch_]{FETCH,WRITE}_SIZE.csv: separate rocprofv3 --pmc passes of the same "
                                                     "launch; FETCH_SIZE doubled per the gfx950 correction; includes the logits the kernel writes")
        except Exception:
            pass
        if not a.no_extras and world == 1:
            extra = {}
            other = "torch_cpu" if a.tie_mode == "canonical" else "canonical"
            half = max(2, a.steps // 2)
            try:
                if a.mode == "batch":
                    p2 = [BatchStep(cfg, dev, other, call_plans, ks)]
                else:
                    p2, _ = build_plans(cfg, dev, other, expanded=False, n_streams=a.streams)
                d2 = time_steps(p2, half, 1, dev, None)
                extra[f"tokens_per_s_tie_mode_{other}"] = half * tokens_per_step / d2
                del p2
            except Exception as e:
                extra[f"tie_mode_{other}_error"] = str(e)
            if a.mode == "batch":
                # two prompts in flight: consecutive steps alternate between two HIP streams (own outputs and workspace
                # each), so the latency-bound select of one prompt overlaps the K scan / softmax of the next
                try:
                    s2 = torch.cuda.Stream(device=dev)
                    with torch.cuda.stream(s2):
                        alt = BatchStep(cfg, dev, a.tie_mode, call_plans, ks)
                    torch.cuda.synchronize(dev)
                    pair = [plans[0], alt]
                    for i in range(2):
                        pair[i % 2].run()
                    torch.cuda.synchronize(dev)
                    n2 = 2 * half
                    t0 = time.perf_counter()
                    for i in range(n2):
                        pair[i % 2].run()
                    torch.cuda.synchronize(dev)
                    extra["tokens_per_s_two_prompts_in_flight_2_streams"] = n2 * tokens_per_step / (time.perf_counter() - t0)
                    del alt, pair
                except Exception as e:
                    extra["two_streams_error"] = str(e)
            if a.mode == "batch" and a.dot_mode == "exact":
                try:
                    _kvc.DOT_MODE = "mfma16"
                    pf = [BatchStep(cfg, dev, a.tie_mode, call_plans, ks)]
                    df = time_steps(pf, half, 1, dev, None)
                    extra["tokens_per_s_dot_mode_mfma16_tolerance_mode"] = half * tokens_per_step / df
                    del pf
                except Exception as e:
                    extra["dot_mode_mfma16_error"] = str(e)
                finally:
                    _kvc.DOT_MODE = a.dot_mode
            try:
                d0 = time_steps(call_plans, half, 1, dev, None)
                extra[f"tokens_per_s_32_kvc_compress_calls_{a.streams}_streams_host_enqueued"] = half * tokens_per_step / d0
            except Exception as e:
                extra["calls_error"] = str(e)
            try:
                p1, _ = build_plans(cfg, dev, a.tie_mode, expanded=False, n_streams=1)
                d1 = time_steps(p1, max(2, a.steps // 2), 1, dev, None)
                extra["tokens_per_s_single_stream_host_enqueued"] = max(2, a.steps // 2) * tokens_per_step / d1

                del p1
            except Exception as e:
                extra["single_stream_error"] = str(e)
            try:
                p3, _ = build_plans(cfg, dev, a.tie_mode, expanded=True, n_streams=a.streams)
                d3 = time_steps(p3, max(2, a.steps // 2), 1, dev, None)
                extra["tokens_per_s_expanded_kv_as_reference_passes"] = max(2, a.steps // 2) * tokens_per_step / d3
                del p3
            except Exception as e:
                extra["expanded_kv_error"] = str(e)
            out["extras"] = extra
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, dev=dev)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
