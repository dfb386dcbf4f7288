#!/usr/bin/env python
"""bench.py -- disparity-volumes/s of the MGM hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3] [--batch B] [--mode pairs|directions]

One "step" = one full pass of the hot path over one batch of synthetic stereo pairs whose images are
already resident in HBM: census transform + W*H*L cost volume (K1, K2), all-direction MGM
aggregation (K3), ordered sum + over-count fix + WTA + V-fit (K4-K6).  Outputs stay on the device.

N > 1: one process per GPU.  `python bench.py --gpus N` started WITHOUT a torch.distributed.run environment
launches itself under it (N ranks on 127.0.0.1); started by `python -m torch.distributed.run ... bench.py --gpus N`
it is one of the ranks.  In the default `pairs` mode every rank processes its own independent pairs -- the
reference has no cross-pair coupling, so there is no data-path collective -- and the job rate is all volumes of
all ranks over the slowest rank's time (weak scaling).  `--mode directions` shards the passes of ONE volume over
the ranks (strong scaling, cfg4-size volumes).  torch is used for the process group, the barrier and the
max-reduction only.

Workload matrix (BASELINE.json configs; cfg1 is the reference's CPU-runnable case and a parity test, not a bench line):
    --workload cfg3   (default; the configuration the metric is quoted on)   1920x1080x256, census 5x5, -O 8, TSGM 3, FH
    --workload cfg3h  the same with Hirschmueller potentials
    --workload cfg2   1920x1080x128, census 3x3, -O 4, TSGM 2
    --workload cfg4 [--mode directions]   4096x4096x192, census 5x5, -O 8, TSGM 3 (one volume per step)
    --workload cfg5   1024x1024x128, census 3x3, -O 4, TSGM 2, 16 pairs per step and GPU (throughput mode)

Rank 0 prints ONE JSON line, the last thing on stdout; fields `roofline`, `cpu_baseline` and `parity` are described
in DESIGN.md.  PARITY GATE: the disparity and cost maps of pair 0 computed by the LAST timed step are compared, bit for
bit, with the CPU oracle run on the same seeded pair in the same process; a mismatch fails the run (exit code 3).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs: shape, labels, census window, NDIR, TSGM, FH, P1, P2
WORKLOADS = {
    "cfg2": dict(nx=1920, ny=1080, dmin=-127, dmax=0, win=3, NDIR=4, MGM=2, FH=0, P1=8.0, P2=32.0,
                 desc="1920x1080 synthetic pair, 128 disparities, CENSUS 3x3, -O 4 TSGM=2"),
    "cfg3": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=1, P1=2.0, P2=20000.0,
                 desc="1920x1080 synthetic pair, 256 disparities, CENSUS 5x5, -O 8 TSGM=3, FH truncated-linear V"),
    "cfg3h": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=0, P1=8.0, P2=32.0,
                  desc="1920x1080 synthetic pair, 256 disparities, CENSUS 5x5, -O 8 TSGM=3, Hirschmueller V"),
    "cfg4": dict(nx=4096, ny=4096, dmin=-96, dmax=95, win=5, NDIR=8, MGM=3, FH=0, P1=8.0, P2=32.0,
                 desc="4096x4096 satellite-style synthetic pair, 192 disparities, CENSUS 5x5, -O 8 TSGM=3"),
    "cfg5": dict(nx=1024, ny=1024, dmin=-127, dmax=0, win=3, NDIR=4, MGM=2, FH=0, P1=8.0, P2=32.0,
                 desc="1024x1024 synthetic pair, 128 disparities, CENSUS 3x3, -O 4 TSGM=2 (throughput mode)"),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PARITY_MAX_CELLS = 1.2e9  # the in-run oracle takes whole volumes up to this size (cfg4: see tests/test_gpu_fullsize.py)


def pair_of(w, seed_offset=0):
    """The synthetic pair every leg of the run uses for a given index (GPU steps, parity oracle, CPU baseline)."""
    from mgm_amd import synth
    return synth.stereo_pair(w["nx"], w["ny"], w["dmin"] * 3 // 4, max(0, w["dmax"] * 3 // 4), seed=synth.SEED + seed_offset)


def kernel_source_hash():
    """Identity of the kernels a committed PMC summary was measured on (profiles/*_traffic.json)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mgm_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run_step(ctx, dus, dvs, w, outs, outcs, cvs=None):
    """One step = one batch: the cost volume of every pair, ONE pass launch over the batch, WTA+vfit per volume."""
    cvs = cvs or [None] * len(dus)
    cvs = [ctx.costvolume_dev(du, dv, w["dmin"], w["dmax"], "none", "census", float("inf"), w["win"], into=cv)
           for du, dv, cv in zip(dus, dvs, cvs)]
    ctx.aggregate_batch_dev(cvs, w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, None, "vfit", outs, outcs, want_S=False)
    return cvs


def oracle_whole_volume(w, threads):
    """Pair 0 through the CPU oracle (oracle/mgm_oracle.c), whole volume, `threads` OpenMP threads (the reference
    parallelises each diagonal of a pass the same way, mgm_core.cc:505-579).  Returns (disp, cost, seconds)."""
    from oracle.oracle import Oracle
    orc = Oracle(threads=threads)
    u, v, _ = pair_of(w)
    t0 = time.perf_counter()
    C = orc.costvolume(u, v, w["dmin"], w["dmax"], "none", "census", np.inf, w["win"])
    S, o, c = orc.mgm(C, w["dmin"], w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1)
    ro, rc = orc.refine(S, w["dmin"], "vfit", o, c)
    return ro, rc, time.perf_counter() - t0


def cpu_baseline(w, whole, seconds_target=15.0):
    """The CPU oracle (plain-C port of the reference path) timed on the GPU box's host cores: one thread on a row band of
    the same workload sized to about `seconds_target` seconds, and -- `whole` = (seconds, threads), measured by the
    parity leg -- one whole volume on every core the process may use.  The better of the two is the headline figure."""
    from mgm_amd import synth
    from oracle.oracle import Oracle
    orc = Oracle(threads=1)
    nx, L = w["nx"], w["dmax"] - w["dmin"] + 1

    def band(rows):
        u, v, _ = synth.stereo_pair(nx, rows, w["dmin"] * 3 // 4, max(0, w["dmax"] * 3 // 4))
        t0 = time.perf_counter()
        C = orc.costvolume(u, v, w["dmin"], w["dmax"], "none", "census", np.inf, w["win"])
        S, o, c = orc.mgm(C, w["dmin"], w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1)
        orc.refine(S, w["dmin"], "vfit", o, c)
        return time.perf_counter() - t0

    per_row = band(8) / 8  # calibrate on a thin band, then size the sample
    rows = int(max(16, min(w["ny"], seconds_target / per_row)))
    dt = band(rows)
    vol_per_s = (rows / w["ny"]) / dt  # linear in the number of rows
    res = {"value": vol_per_s, "unit": "disparity-volumes/s", "cores": 1, "kind": "port",
           "sample": "%dx%dx%d band (%d of %d rows) of the same workload, %.1f s of CPU time, extrapolated "
                     "linearly in rows; oracle/mgm_oracle.c, 1 thread" % (nx, rows, L, rows, w["ny"], dt),
           "mcell_updates_per_s": nx * rows * L * w["NDIR"] / dt / 1e6}
    if whole is not None:
        dto, T = whole
        res["openmp"] = {"value": 1.0 / dto, "cores": T, "sample": "one whole %dx%dx%d volume, %.1f s wall" % (nx, w["ny"], L, dto)}
        if 1.0 / dto > res["value"]:
            res.update({"value": 1.0 / dto, "cores": T, "mcell_updates_per_s": nx * w["ny"] * L * w["NDIR"] / dto / 1e6,
                        "sample": res["openmp"]["sample"] + " on %d OpenMP threads; 1 thread: %.4f volumes/s (%s)" % (T, vol_per_s, res["sample"])})
    res["host_cpus"] = os.cpu_count()
    res["cpu_model"] = cpu_model()
    return res


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a rendezvous environment: become the launcher of N ranks of this same file."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    js = [l for l in lines if l.startswith("{") and l.rstrip().endswith("}")]
    for l in lines:
        if not js or l is not js[-1]:
            print(l, file=sys.stderr)
    if js:
        print(js[-1], flush=True)  # the ONE json line, last thing on stdout
    sys.exit(p.returncode if p.returncode or js else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity gate (profiling runs)")
    ap.add_argument("--repeats", type=int, default=None,
                    help="further timed blocks of K steps after the reported one (their rates go to `repeat_values`); "
                         "default: as many as make the GPU phase last about 10 s, at most 8")
    ap.add_argument("--batch", type=int, default=None, choices=list(range(1, 17)),
                    help="pairs per step and GPU: their volumes share ONE launch of the pass kernel (pairs mode).  Default: "
                         "12 (204 GB of Lr volumes at 1920x1080x256 x 8 directions), 16 for workloads of at most 128 labels "
                         "(two / four of those volumes share every wave), 1 for cfg4")
    ap.add_argument("--mode", default="pairs", choices=["pairs", "directions"],
                    help="N>1: 'pairs' = independent pairs, one per GPU (weak scaling, default); 'directions' = ONE "
                         "volume per step, its passes sharded over the GPUs with an ordered RCCL exchange (strong)")
    args = ap.parse_args()
    w = WORKLOADS[args.workload]
    stub = os.environ.get("MGM_BENCH_STUB") == "1"  # tests/test_dist_cpu.py: gloo ranks on CPU, a context that computes nothing

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args)
    import torch
    dist = None
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d runs inside a torch.distributed.run job of %d ranks" % (args.gpus, world))
    if not stub:
        if not torch.cuda.is_available():
            sys.exit("bench.py needs an MI355X: there is no CPU path in mgm_amd")
        torch.cuda.set_device(local)
    if world > 1 or args.mode == "directions":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    n_ranks = dist.get_world_size() if dist is not None else 1  # the world RCCL actually initialised

    if stub:
        from tests.bench_stub import StubContext
        ctx = StubContext(local)
    else:
        import mgm_amd
        ctx = mgm_amd.Context(local)
    from mgm_amd import shard
    nx, ny, L = w["nx"], w["ny"], w["dmax"] - w["dmin"] + 1
    # pairs mode: every rank gets its own pairs (different seeds): independent units, no exchange.
    # directions mode: every rank holds the SAME pair and builds the full cost volume itself.
    if args.batch is None:
        args.batch = 1 if args.workload == "cfg4" else (16 if L <= 128 else 12)
    B = args.batch if args.mode == "pairs" else 1
    dus, dvs, outs, outcs = [], [], [], []
    for b in range(B):
        u, v, _ = pair_of(w, (rank * B + b) if args.mode == "pairs" else 0)
        dus.append(ctx.upload_image(u))
        dvs.append(ctx.upload_image(v))
        outs.append(ctx.new_image(nx, ny))
        outcs.append(ctx.new_image(nx, ny))
    du, dv = dus[0], dvs[0]

    def sync_all():
        if not stub:
            torch.cuda.synchronize()
        ctx.synchronize()
        if world > 1:
            dist.barrier()
            if not stub:
                torch.cuda.synchronize()

    last = {}
    if args.mode == "directions":
        from mgm_amd import dist as mdist

        def step(cv):
            cv = ctx.costvolume_dev(du, dv, w["dmin"], w["dmax"], "none", "census", float("inf"), w["win"], into=cv)
            last["o"], last["c"] = mdist.aggregate_direction_sharded(ctx, cv, w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, "vfit", dist)
            return cv
    else:
        def step(cvs):
            return run_step(ctx, dus, dvs, w, outs, outcs, cvs)

    def timed_block():
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(cv)  # pairs mode: enqueue only, nothing synchronises inside the timed region
        sync_all()
        return shard.max_over_ranks(time.perf_counter() - t0, dist if world > 1 else None, device="cpu" if stub else "cuda")

    cv = None  # the W*H*L volumes are allocated once and refilled every step
    for _ in range(max(1, args.warmup)):
        cv = step(cv)
    sync_all()
    ctx.timing(True)
    ctx.timing_reset()
    dt = timed_block()  # the reported measurement: exactly K steps between two barrier + synchronize brackets
    kern = {}
    for name, ms in ctx.timings():
        kern.setdefault(name, []).append(ms)
    ctx.timing(False)
    # further blocks of K steps: the same measurement again (stability), and a GPU phase long enough for a coarse
    # utilisation sampler to see
    reps = args.repeats if args.repeats is not None else int(min(8, max(0, np.ceil(10.0 / max(dt, 1e-3)) - 1)))
    rep_dt = [timed_block() for _ in range(reps)]

    # ---- parity gate: pair 0 of the last timed step against the CPU oracle --------------------------------------
    cells = float(nx) * ny * L
    parity, whole = None, None
    if rank == 0 and not args.no_parity and not stub:
        if cells > PARITY_MAX_CELLS:
            parity = {"status": "skipped", "why": "%.1f G cells: beyond the in-run oracle (covered by tests/test_gpu_fullsize.py)" % (cells / 1e9)}
        else:
            if args.mode == "directions":
                got_o, got_c = last["o"].cpu().numpy(), last["c"].cpu().numpy()
            else:
                got_o, got_c = outs[0].download()[0], outcs[0].download()[0]
            from oracle.oracle import usable_cpus
            T = min(32, usable_cpus())
            ref_o, ref_c, secs = oracle_whole_volume(w, T)
            whole = (secs, T)

            def nd(a, b):
                a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
                return int(np.sum((a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))))
            bad = nd(ref_o, got_o.reshape(ref_o.shape)) + nd(ref_c, got_c.reshape(ref_c.shape))
            parity = {"status": "bit-exact" if bad == 0 else "FAILED", "differing_words": bad,
                      "what": "refined disparity and cost maps (2 x %dx%d float32) of pair 0 from the last timed step vs oracle/mgm_oracle.c "
                              "on the same pair (%d threads, %.1f s)" % (nx, ny, T, secs)}
    for x in (cv if isinstance(cv, list) else [cv]):
        x.free()

    if rank == 0:
        vols_per_block = args.steps * (n_ranks * B if args.mode == "pairs" else 1)
        if args.mode == "pairs":
            value = shard.job_rate([args.steps * B] * n_ranks, dt)  # whole-job aggregate: every rank did K batches of B volumes
        else:
            value = args.steps / dt                                # K volumes, each computed by all ranks together
        avg = {k: float(np.mean(vs)) for k, vs in kern.items()}
        per_step = {k: float(np.sum(vs)) / args.steps for k, vs in kern.items()}
        pass_name = "k_pass2" if "k_pass2" in avg else "k_pass"
        # in directions mode rank 0 ran NDIR/world passes and summed ny/world rows per launch
        frac_pass = (mdist.passes_of_rank(w["NDIR"], n_ranks, 0)[1] / w["NDIR"]) if args.mode == "directions" else 1.0
        frac_rows = (mdist.row_slabs(ny, n_ranks)[0][1] / ny) if args.mode == "directions" else 1.0
        # Aggregation stage = K3 (pass kernel) + K4-K6 (k_wta): the two launches together do what the
        # reference's aggregation loop does; SURVEY.md 8(d): 12 B per cell per direction.
        # With a batch of B volumes per step the pass kernel is launched once (over all of them) and k_wta B times.
        agg_ms = avg[pass_name] + B * avg["k_wta"]
        alg_bytes = 12.0 * w["NDIR"] * cells * (float(B) if args.mode == "pairs" else (2.0 / 3.0) * frac_pass + (1.0 / 3.0) * frac_rows)
        achieved = alg_bytes / (agg_ms * 1e-3) / 1e9
        # HBM bytes from PMC counters: a committed summary counts only if it was measured on THESE kernels
        traffic, traffic_src = None, None
        for prof in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")), reverse=True):
            tj = json.load(open(os.path.join(ROOT, "profiles", prof)))
            if tj.get("workload") == args.workload and tj.get("pairs_per_step") == B and tj.get("kernel_source_sha") == kernel_source_hash():
                traffic, traffic_src = tj.get("aggregation_hbm_bytes_per_step"), "profiles/" + prof
                break
        roofline = {"bound": "hbm", "kernel": "%s (one launch per batch of %d volumes) + k_wta (one launch per volume): the %d-direction aggregation" % (pass_name, B, w["NDIR"]),
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes,
                    "avg_launch_ms": {k: avg[k] for k in sorted(avg)},
                    "per_kernel": {
                        pass_name: {"alg_bytes": 8.0 * w["NDIR"] * cells * B,
                                    "GBps": 8.0 * w["NDIR"] * cells * B / (avg[pass_name] * 1e-3) / 1e9},
                        "k_wta": {"alg_bytes": (4.0 * w["NDIR"] + 4.0) * cells + 8.0 * nx * ny,
                                  "GBps": ((4.0 * w["NDIR"] + 4.0) * cells) / (avg["k_wta"] * 1e-3) / 1e9},
                        # single-word census costs are written once, as one byte per label (the fp32 volume is never made)
                        "k_cost": {"alg_bytes": 1.0 * cells, "GBps": 1.0 * cells / (avg["k_cost"] * 1e-3) / 1e9}}}
        res = {"metric": "disparity-volumes/sec (W*H*L cost volume -> 8-dir MGM -> WTA+vfit)", "value": value,
               "unit": "disparity-volumes/s", "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if args.mode == "pairs" else "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not stub else "stub (no device work)",
               "config": {"workload": "%s: %s" % (args.workload, w["desc"]), "W": nx, "H": ny, "L": L, "pairs_per_step": B,
                          "NDIR": w["NDIR"], "TSGM": w["MGM"], "potential": "FH" if w["FH"] else "Hirschmueller",
                          "P1": w["P1"], "P2": w["P2"], "census_win": w["win"], "refine": "vfit",
                          "parallelism": ("independent pairs, %d per step and GPU, no data-path collective" % B if args.mode == "pairs" else
                                          "one volume, %d-way direction sharding, ordered RCCL slab exchange" % n_ranks)
                          if n_ranks > 1 or args.mode == "directions" else "1 GPU"},
               "roofline": roofline,
               "kernel_ms_per_step": per_step,
               "repeat_values": [vols_per_block / t for t in rep_dt],
               "parity": parity,
               "mcell_updates_per_s": vols_per_block * cells * w["NDIR"] / dt / 1e6}
        if n_ranks == 1 and not args.no_cpu_baseline and not stub:
            res["cpu_baseline"] = cpu_baseline(w, whole)
    ctx.close()
    if dist is not None and dist.is_initialized():
        dist.destroy_process_group()
    failed = False
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(res), flush=True)  # the ONE json line, last thing on stdout
        failed = parity is not None and parity["status"] == "FAILED"
        if failed:
            print("bench.py: PARITY GATE FAILED: %d words differ from the oracle" % parity["differing_words"], file=sys.stderr, flush=True)
    if dist is not None:
        os._exit(3 if failed else 0)  # librccl leaves a version banner in the C stdio buffer that would be flushed after the json
    sys.exit(3 if failed else 0)


if __name__ == "__main__":
    main()
