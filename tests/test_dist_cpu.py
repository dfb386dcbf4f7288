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


# ---- direction sharding: the ordered exchange of Lr slabs (mgm_amd/dist.py) over gloo ----------
def _dir_worker(rank, world, port, q, NDIR, ny=13):
    import numpy as np
    from mgm_amd import dist as mdist
    from mgm_amd import synth
    from oracle.oracle import Oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle(threads=1)
    nx, L = 21, 10                                      # 13 rows over 2 ranks: uneven slabs
    C = synth.raw_volume(nx, ny, L, seed=9, inf_frac=0.05)
    S, out, outc, lr = orc.mgm(C, -2, 8.0, 32.0, NDIR, 3, 0, 1, None, dump_lr=True)
    first, count = mdist.passes_of_rank(NDIR, world, rank)
    mine = [torch.from_numpy(np.ascontiguousarray(lr[p])) for p in range(first, first + count)]  # "my" passes only
    recv = mdist.exchange_lr(mine, NDIR, ny, dist, like=torch.empty((0, nx, L), dtype=torch.float32))
    r0, nr = mdist.row_slabs(ny, world)[rank]
    Srows = mdist.ordered_sum_numpy([recv[p].numpy() for p in range(NDIR)], C[r0:r0 + nr], 1)
    a, b = Srows.view(np.uint32), S[r0:r0 + nr].view(np.uint32)
    nan = np.isnan(Srows) & np.isnan(S[r0:r0 + nr])
    q.put((rank, int(np.sum((a != b) & ~nan)), nr))
    dist.destroy_process_group()


def test_direction_sharding_exchange_is_bit_exact():
    for NDIR in (8, 3, 1):                              # 3 passes over 2 ranks: uneven blocks; 1 pass: rank 1 runs none (world > NDIR)
        world, port = 2, _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ps = [ctx.Process(target=_dir_worker, args=(r, world, port, q, NDIR)) for r in range(world)]
        [p.start() for p in ps]
        res = [q.get(timeout=180) for _ in ps]
        [p.join(60) for p in ps]
        assert all(p.exitcode == 0 for p in ps)
        assert sorted(r[2] for r in res) == [6, 7] and all(r[1] == 0 for r in res), res


def _spawn_dir(world, NDIR, ny):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_dir_worker, args=(r, world, port, q, NDIR, ny)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=300) for _ in ps]
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    return res


def test_direction_sharding_exchange_world4_and_world8():
    """the ordered exchange had only ever run with TWO ranks.  Here the same code (mgm_amd/dist.py: grouped
    point-to-point sends, the schedule the RCCL leg posts) with 4 and 8 gloo ranks: 8 / 4 / 2 passes (fewer passes than
    ranks: ranks without a pass still receive), odd row counts, and fewer rows than ranks (ranks without rows still
    send).  Every rank's rows of the ordered sum S must equal the oracle's mgm() bit for bit (mgm_core.cc:582-587 is a sum
    in PASS order -- not thread-finish order, 798-805, and not an all-reduce)."""
    from mgm_amd import dist as mdist
    for world, NDIR, ny in ((4, 8, 13), (4, 2, 9), (8, 8, 13), (8, 4, 5)):
        res = _spawn_dir(world, NDIR, ny)
        rows = sorted((r[0], r[2]) for r in res)
        assert [n for _, n in rows] == [n for _, n in mdist.row_slabs(ny, world)], (world, NDIR, ny, rows)
        assert sum(n for _, n in rows) == ny and all(r[1] == 0 for r in res), (world, NDIR, ny, res)


def test_partitions():
    from mgm_amd import dist as mdist
    #  NDIR in {2, 4, 8} x n in {2, 4, 8} x odd row counts, rows fewer than ranks included: blocks of passes and
    # slabs of rows are contiguous, cover everything exactly once, differ in size by at most one -- and the C library's own
    # plan (mgm_multi_plan, what mgm_multi_aggregate and the CLI's MGM_DEVICES use) is the same partition
    import ctypes as C
    import mgm_amd
    L = mgm_amd.load_library()
    for NDIR in (2, 4, 8):
        for world in (2, 4, 8):
            for ny in (1, 3, 7, 13, 1079, 1081, 4095):
                s = mdist.row_slabs(ny, world)
                assert len(s) == world and s[0][0] == 0 and sum(n for _, n in s) == ny
                assert all(s[i][0] + s[i][1] == s[i + 1][0] for i in range(world - 1))
                assert max(n for _, n in s) - min(n for _, n in s) <= 1
                blocks = [mdist.passes_of_rank(NDIR, world, r) for r in range(world)]
                assert [p for f, n in blocks for p in range(f, f + n)] == list(range(NDIR))
                assert max(n for _, n in blocks) - min(n for _, n in blocks) <= 1
                assert mdist.n_rounds(NDIR, world) == max(n for _, n in blocks)
                arr = [(C.c_int * world)() for _ in range(4)]
                assert L.mgm_multi_plan(world, NDIR, ny, *arr) == 0
                assert [(arr[0][r], arr[1][r]) for r in range(world)] == blocks
                assert [(arr[2][r], arr[3][r]) for r in range(world)] == s
    for ny, world in ((1080, 8), (13, 2), (5, 8)):
        s = mdist.row_slabs(ny, world)
        assert sum(n for _, n in s) == ny and all(s[i][0] + s[i][1] == s[i + 1][0] for i in range(world - 1))
    for NDIR in (8, 4, 3, 1):
        for world in (1, 2, 4, 8):
            blocks = [mdist.passes_of_rank(NDIR, world, r) for r in range(world)]
            flat = [p for f, n in blocks for p in range(f, f + n)]
            assert flat == list(range(NDIR))
            assert all(mdist.owner_of_pass(p, NDIR, world) == [r for r, (f, n) in enumerate(blocks) if f <= p < f + n][0]
                       for p in range(NDIR))


# ---- bench.py's own launcher: `python bench.py --gpus 2` must start its two ranks itself -----------------------
def test_bench_launches_its_own_ranks():
    """The driver's command line, unchanged, on CPU: bench.py re-executes itself under torch.distributed.run with
    two ranks (gloo here, RCCL on the GPU box), times K steps between barriers, and rank 0 prints the one JSON line
    with the world size that was really initialised.  The context is a stub that computes nothing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "run_bench_stub.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--workload", "cfg5",
                        "--batch", "2", "--repeats", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["pairs_per_step"] == 2 and d["data"].startswith("stub")
    # whole-job aggregate: 2 ranks x 4 steps x 2 volumes over the slowest rank's time
    assert abs(d["value"] - 16 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    assert len(d["repeat_values"]) == 1 and d["roofline"]["per_kernel"]["k_pass2"]["GBps"] > 0
    assert d["roofline"]["per_kernel"]["k_pass2"]["format_bytes"] > 0 and "cfg5_replicas" not in d  # (explicit workload: no extra legs)


def _run_bench(args, extra_env=None, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **(extra_env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "run_bench_stub.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None), lines


def test_driver_command_line_runs_the_extra_legs():
    """`python bench.py --gpus 2 --steps K --warmup W` and nothing else -- the driver's command: after the cfg3 headline the
    SAME json line carries the cfg5 replicas and the direction-sharded leg (here: agreement + ordered slab exchange over
    gloo on the stub's host tensors), and it is still ONE line."""
    r, d, lines = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["workload"].startswith("cfg3")
    c5 = d["cfg5_replicas"]
    assert c5["n_gpus"] == 2 and c5["pairs_per_step_per_gpu"] == 16 and c5["value"] > 0 and c5["scaling"] == "weak"
    dr = d["directions"]
    assert dr["ranks"] == 2 and dr["scaling"] == "strong" and dr["transport"] == "rccl"
    assert dr["rccl"]["rccl_ranks"] == 2 and dr["rccl"]["differs_from_single"] == 0 and dr["value"] == dr["rccl"]["value"]
    # the self-diagnosis of a first node run (round 4): the link probe in the exchange's own pattern, the rounds of the
    # exchange, DESIGN.md section 6's model evaluated with the measured link rate, and the pair split by run
    lp = dr["rccl_link_probe"]
    assert len(lp["gbps"]) == 2 and len(lp["gbps"][0]) == 1 and lp["min"] > 0 and lp["max"] >= lp["min"]
    assert dr["rccl"]["exchange_rounds"] == 4  # 8 passes over 2 ranks
    m = dr["model"]["prediction"]
    assert m["world"] == 2 and m["passes_per_rank"] == 4 and m["exchange_ms"] > 0 and m["total_ms"] > m["exchange_ms"]
    assert d["cfg4_pairs2"]["ranks"] == 2 and d["cfg4_pairs2"]["value"] > 0
    assert "extras_note" not in d


def test_extras_watchdog_prints_the_headline_when_a_leg_hangs():
    """A leg that never returns (the stub sleeps in the cfg5 leg when told to) must not take the headline with it: the
    watchdog prints the one line with what had been measured and ends every rank."""
    r, d, lines = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "0", "--extras-timeout", "6"],
                             {"MGM_STUB_HANG_AT": "cfg5"}, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and d["value"] > 0 and d["config"]["workload"].startswith("cfg3")
    assert "watchdog" in d["extras_note"] and "directions" not in d


def test_single_rank_default_line_has_both_legs():
    r, d, lines = _run_bench(["--steps", "2", "--warmup", "1", "--repeats", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and d["n_gpus"] == 1
    assert d["cfg5_replicas"]["n_gpus"] == 1 and d["directions"]["ranks"] == 1 and d["directions"]["transport"] == "single"
    # the driver's line carries the one-pair legs and the round-5 legs (range images, TSGM_ITER, the fall-back kernels)
    v = d["variants"]
    assert "error" not in v, v.get("error")
    for k in ("cfg3 x1", "cfg3 x2", "cfg2 x1", "cfg2 x2", "cfg1s x2", "cfg3r x1", "cfg3i2 x1", "cfg3w3 x1", "cfg3L1536 x1", "cfg3nan x1"):
        assert v[k]["value"] > 0 and "roofline_frac" in v[k], k
    assert "frac_range_proportional" in v["cfg3r x1"]
    r = d["roofline"]
    assert 0 < r["moved_over_algorithmic"] <= 1.0 and r["frac_of_achievable"] > 0 and isinstance(r["saturated"], bool)


def test_exchange_rounds_and_timeout():
    """SlabExchange posts round by round (the overlap schedule) and gives the same slabs; a peer that never posts makes
    the receiver raise ExchangeTimeout instead of blocking forever."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rounds_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=180) for _ in ps]
    [p.join(60) for p in ps]
    assert sorted(res) == [(0, 0, "timeout"), (1, 0, "absent")], res


def _rounds_worker(rank, world, port, q):
    from datetime import timedelta
    from mgm_amd import dist as mdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=120))
    NDIR, ny, nx, L = 5, 11, 7, 6  # 5 passes over 2 ranks: 3 rounds, rank 1 has no pass in the last one
    vols = torch.rand((NDIR, ny, nx, L), generator=torch.Generator().manual_seed(5))
    first, count = mdist.passes_of_rank(NDIR, world, rank)
    ex = mdist.SlabExchange(NDIR, ny, nx, L, dist, None, torch.float32, "cpu", 60.0)
    assert ex.rounds == 3
    for k in range(ex.rounds):
        assert mdist.agree(True, dist)  # (what the overlapped schedule does between launches)
        ex.post(k, vols[first + k] if k < count else None)
    recv = ex.finish()
    r0, nr = mdist.row_slabs(ny, world)[rank]
    bad = int((recv != vols[:, r0:r0 + nr]).sum())
    assert mdist.agree(rank == 0, dist) is False  # one rank reporting a failure is every rank's failure
    # rank 1 goes away without posting: rank 0's exchange must time out, not hang
    what = "absent"
    if rank == 0:
        ex2 = mdist.SlabExchange(NDIR, ny, nx, L, dist, None, torch.float32, "cpu", 3.0)
        try:
            for k in range(ex2.rounds):
                ex2.post(k, vols[first + k] if k < count else None)
            ex2.finish()
            what = "completed?"
        except mdist.ExchangeTimeout:
            what = "timeout"
    q.put((rank, bad, what))
    q.close()
    q.join_thread()  # (os._exit below does not flush the queue's feeder thread)
    if rank == 1:
        import time
        time.sleep(8)  # stay alive (connected) while rank 0 waits in vain
    os._exit(0)  # (a process group with a dead exchange is not torn down in an orderly way)


def test_pairs2_mode_and_the_sharding_model():
    """`--mode pairs2` (one pair per two GPUs, one mgm() run each) prints its one line on two stub ranks; DESIGN.md section 6's
    model reproduces the table's figures from its inputs (8 GPUs, 1.61 GB per link at 153 GB/s = 10.5 ms)."""
    from mgm_amd import dist as mdist
    r, d, lines = _run_bench(["--gpus", "2", "--mode", "pairs2", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["workload"].startswith("cfg4")
    m = mdist.sharding_model(8, 8, 12.885, {1: 7.8}, 17.3, 1.25, 153.0)
    assert m["passes_per_rank"] == 1 and abs(m["gb_per_link"] - 1.6106) < 1e-3 and abs(m["exchange_ms"] - 10.53) < 0.05
    assert abs(m["total_ms"] - (1.25 + 7.8 + m["exchange_ms"] + 17.3 / 8)) < 1e-9
    m2 = mdist.sharding_model(2, 8, 12.885, {4: 14.0}, 17.3, 1.25, 153.0)
    assert m2["passes_per_rank"] == 4 and abs(m2["exchange_ms"] - 168.4) < 0.5


def test_driver_command_line_on_four_ranks():
    """The driver's SCALE command at N = 4 (stub ranks over gloo): one valid line, the direction-sharded leg ran on all four
    ranks (two rounds of the exchange), the pair split ran, and the line says what scales."""
    r, d, lines = _run_bench(["--gpus", "4", "--steps", "2", "--warmup", "1", "--repeats", "0"], timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and d["n_gpus"] == 4 and d["scaling"] == "weak"
    dr = d["directions"]
    assert dr["ranks"] == 4 and dr["transport"] == "rccl" and dr["rccl"]["exchange_rounds"] == 2 and dr["rccl"]["differs_from_single"] == 0
    assert dr["differs_from_single"] == {"rccl": 0} and dr["fallback_chain"] == ["rccl: ok"]
    assert len(dr["rccl_link_probe"]["gbps"]) == 4 and dr["model"]["prediction"]["world"] == 4
    assert d["cfg4_pairs2"]["ranks"] == 4 and "replicas" in d["multi_gpu_note"]


def test_transport_fallbacks_keep_the_line_valid():
    """rccl fails -> the peer path's figure is the leg's; rccl AND peer fail -> "replicas": no figure, the chain says why, the
    headline and every other key of the line are still there (the first node run must not lose its line to a
    transport error)."""
    r, d, lines = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "0"], {"MGM_STUB_FAIL_AT": "rccl"})
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-2000:]
    dr = d["directions"]
    assert dr["transport"] == "peer" and dr["ranks"] == 2 and dr["value"] == dr["peer"]["value"] and "error" in dr["rccl"]
    assert dr["fallback_chain"][0].startswith("rccl: failed") and dr["fallback_chain"][1] == "peer: ok" and "error" in dr["rccl_link_probe"]
    r, d, lines = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "0"], {"MGM_STUB_FAIL_AT": "rccl,peer"})
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-2000:]
    dr = d["directions"]
    assert dr["transport"] == "replicas" and dr["value"] is None and dr["ranks"] == 2 and d["value"] > 0
    assert dr["fallback_chain"][-1].startswith("replicas:") and dr["differs_from_single"] == {}


def test_a_rank_killed_mid_exchange_leaves_a_valid_line():
    """the first node run will be the RCCL exchange's first execution with n > 1 -- a rank that DIES between the
    agreement and its first transfer (SIGKILL: no Python teardown, its sockets just close) must not take the line with it.  The
    survivors' exchange fails or times out, torch.distributed.run terminates them (SIGTERM), and rank 0 still prints the ONE json
    line with the headline and everything measured before the leg; the exit code says a rank failed."""
    r, d, lines = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "0", "--extras-timeout", "120"],
                             {"MGM_STUB_DIE_AT": "exchange"}, timeout=600)
    assert len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert d["value"] > 0 and d["n_gpus"] == 2 and d["config"]["workload"].startswith("cfg3") and d["cfg5_replicas"]["value"] > 0
    dr = d.get("directions", {})
    assert "value" not in dr or dr.get("transport") in (None, "replicas", "peer"), dr  # (no figure from the leg that lost a rank)
    assert "extras_note" in d or "error" in dr or "error" in dr.get("rccl", {}), d.keys()
    assert r.returncode != 0  # the launcher reports the dead rank
