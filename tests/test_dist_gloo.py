"""The N > 1 path of bench.py on CPU: two gloo ranks (127.0.0.1), replicas with different prompts, the job time is the
MAX over ranks and the value the sum of both ranks' tokens over it.  There is no data-path collective to test — the
path shards by independent prompts/layers — so this covers exactly what bench.py does with N GPUs."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        my_dt = 0.5 + 0.25 * rank                       # rank 1 is the slow replica
        dist.barrier()
        dt = bench.reduce_max_time(my_dt, dist, torch.device("cpu"))
        value = bench.job_value(world, 10, 8000 * 32, dt)
        # replicas work on different synthetic prompts
        from kvcache_factory_amd import synth
        x = synth.normal((4,), bench.rank_seed(rank))
        gathered = [torch.zeros(4) for _ in range(world)]
        dist.all_gather(gathered, x)
        q.put((rank, dt, value, [g.tolist() for g in gathered]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_replicas_aggregate_like_bench():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, dt0, v0, g0), (r1, dt1, v1, g1) = out
    assert dt0 == dt1 == 0.75                                   # max over ranks
    assert v0 == v1 == 2 * 10 * 8000 * 32 / 0.75                # both replicas' tokens over the job time
    assert g0 == g1 and g0[0] != g0[1]                          # different prompts per rank
