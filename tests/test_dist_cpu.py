"""The N > 1 path of bench.py on CPU: world_size 2, gloo, 127.0.0.1.  Checks that pairs are dealt
to ranks exactly once, that the barrier + MAX-over-ranks timing works, and that the job rate is
the whole-job aggregate."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mgm_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.pairs_of_rank(16, world, rank)          # cfg5: 16 independent pairs
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    dist.barrier()
    elapsed = 0.010 * (rank + 1)                           # rank 1 is the slow one
    tmax = shard.max_over_ranks(elapsed, dist)
    rate = shard.job_rate([len(g) for g in gathered], tmax)
    q.put((rank, gathered, tmax, rate))
    dist.destroy_process_group()


def test_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in ps]
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for rank, gathered, tmax, rate in res:
        flat = sorted(i for g in gathered for i in g)
        assert flat == list(range(16))                     # every pair exactly once
        assert abs(tmax - 0.020) < 1e-12                   # the slowest rank's time on every rank
        assert abs(rate - 16 / 0.020) < 1e-6               # whole-job aggregate, not per-GPU


def test_single_rank_shortcuts():
    assert shard.pairs_of_rank(5, 1, 0) == [0, 1, 2, 3, 4]
    assert shard.max_over_ranks(1.5) == 1.5
    assert shard.job_rate([3], 1.5) == 2.0
