"""Multi-process (gloo, world_size 2, CPU) test of the only inter-GPU exchange of the path: the
scalar all-reduce of {min cost, max |dcost|, #active, #converged} over independent shards."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ilqr_amd.dist import shard_range, allreduce_status, local_stats, to_status, StatusExchange


def test_shard_ranges_partition_the_batch():
    for total, world in ((8192, 8), (10, 3), (7, 8), (1024, 8)):
        cover = []
        for r in range(world):
            lo, hi = shard_range(total, world, r)
            cover += list(range(lo, hi))
        assert cover == list(range(total))
        sizes = [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)
    total = 37
    cost = rng.uniform(1, 100, total)
    prev = cost + rng.uniform(0, 1, total)
    status = rng.integers(0, 4, total)
    lo, hi = shard_range(total, world, rank)
    s = local_stats(cost[lo:hi], prev[lo:hi], status[lo:hi])
    x = StatusExchange()                 # the all-gather form used by bench.py (CPU tensors here)
    for _ in range(3):                   # several launches: exercises the double buffering
        x.launch(lambda t: t.copy_(s))
    gx = x.result()
    allreduce_status(s)
    g = to_status(s)
    assert (gx.min_cost, gx.max_dcost, gx.n_active, gx.n_converged) == (g.min_cost, g.max_dcost, g.n_active, g.n_converged)
    q.put((rank, g.min_cost, g.max_dcost, g.n_active, g.n_converged,
           float(cost.min()), float(np.abs(cost - prev).max()), int((status == 0).sum()), int((status == 1).sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_status_allreduce_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert abs(r[1] - r[5]) < 1e-12 and abs(r[2] - r[6]) < 1e-12 and r[3] == r[7] and r[4] == r[8]


def test_single_process_allreduce_is_identity():
    s = torch.tensor([3.0, 0.5, 10.0, 2.0], dtype=torch.float64)
    assert torch.equal(allreduce_status(s.clone()), s)
