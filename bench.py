#!/usr/bin/env python
"""bench.py -- disparity-volumes/s of the MGM hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3]

One "step" = one full pass of the hot path over one synthetic stereo pair whose images are
already resident in HBM: census transform + W*H*L cost volume (K1, K2), all-direction MGM
aggregation (K3), ordered sum + over-count fix + WTA + V-fit (K4-K6).  Outputs stay on the
device.  With N > 1 (launched by torch.distributed.run, one process per GPU) every rank
processes its own independent pairs -- the reference has no cross-pair coupling, so there
is no data-path collective -- and the job rate is N*K volumes over the slowest rank's time
(weak scaling).  torch is used for the process group, the barrier and the max-reduction only.

Rank 0 prints ONE JSON line; see the fields `roofline` and `cpu_baseline` in DESIGN.md.
"""
import argparse
import json
import os
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


def run_step(ctx, dus, dvs, w, outs, outcs, cvs=None):
    """One step = one batch: the cost volume of every pair, ONE pass launch over the batch, WTA+vfit per volume."""
    cvs = cvs or [None] * len(dus)
    cvs = [ctx.costvolume_dev(du, dv, w["dmin"], w["dmax"], "none", "census", float("inf"), w["win"], into=cv)
           for du, dv, cv in zip(dus, dvs, cvs)]
    ctx.aggregate_batch_dev(cvs, w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, None, "vfit", outs, outcs, want_S=False)
    return cvs


def cpu_baseline(w, seconds_target=15.0):
    """The CPU oracle (plain-C port of the reference path, 1 thread) timed on a row-band of the same
    workload; the band height is chosen so that the run takes about `seconds_target` seconds."""
    from mgm_amd import synth
    from oracle.oracle import Oracle
    orc = Oracle(threads=1)
    nx, L = w["nx"], w["dmax"] - w["dmin"] + 1
    # calibrate on a thin band, then size the sample
    rows = 8
    u, v, _ = synth.stereo_pair(nx, rows, w["dmin"] * 3 // 4, 0)
    t0 = time.perf_counter()
    C = orc.costvolume(u, v, w["dmin"], w["dmax"], "none", "census", np.inf, w["win"])
    S, o, c = orc.mgm(C, w["dmin"], w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1)
    orc.refine(S, w["dmin"], "vfit", o, c)
    per_row = (time.perf_counter() - t0) / rows
    rows = int(max(16, min(w["ny"], seconds_target / per_row)))
    u, v, _ = synth.stereo_pair(nx, rows, w["dmin"] * 3 // 4, 0)
    t0 = time.perf_counter()
    C = orc.costvolume(u, v, w["dmin"], w["dmax"], "none", "census", np.inf, w["win"])
    S, o, c = orc.mgm(C, w["dmin"], w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1)
    orc.refine(S, w["dmin"], "vfit", o, c)
    dt = time.perf_counter() - t0
    vol_per_s = (rows / w["ny"]) / dt  # linear in the number of rows
    res = {"value": vol_per_s, "unit": "disparity-volumes/s", "cores": 1, "kind": "port",
           "sample": "%dx%dx%d band (%d of %d rows) of the same workload, %.1f s of CPU time, extrapolated "
                     "linearly in rows; oracle/mgm_oracle.c, 1 thread" % (nx, rows, L, rows, w["ny"], dt),
           "mcell_updates_per_s": nx * rows * L * w["NDIR"] / dt / 1e6}
    # The reference parallelises each diagonal of a pass with OpenMP (mgm_core.cc:505-579); the port does the same.
    # That only pays on full-length diagonals, so this leg runs ONE whole volume on up to 32 threads.
    from oracle.oracle import usable_cpus
    T = min(32, usable_cpus())
    if T > 1 and os.environ.get("MGM_BENCH_OMP", "1") != "0":
        orc.set_threads(T)
        u, v, _ = synth.stereo_pair(nx, w["ny"], w["dmin"] * 3 // 4, 0)
        t0 = time.perf_counter()
        C = orc.costvolume(u, v, w["dmin"], w["dmax"], "none", "census", np.inf, w["win"])
        S, o, c = orc.mgm(C, w["dmin"], w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1)
        orc.refine(S, w["dmin"], "vfit", o, c)
        dto = time.perf_counter() - t0
        res["openmp"] = {"value": 1.0 / dto, "cores": T, "sample": "one whole %dx%dx%d volume, %.1f s wall" % (nx, w["ny"], L, dto)}
        if 1.0 / dto > res["value"]:  # the headline CPU figure is the better of the two legs
            res.update({"value": 1.0 / dto, "cores": T, "mcell_updates_per_s": nx * w["ny"] * L * w["NDIR"] / dto / 1e6,
                        "sample": res["openmp"]["sample"] + " on %d OpenMP threads; 1 thread: %.4f volumes/s (%s)" % (T, vol_per_s, res["sample"])})
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=None, choices=list(range(1, 17)),
                    help="pairs per step and GPU: their volumes share ONE launch of the pass kernel (pairs mode).  Default: "
                         "12 (204 GB of Lr volumes at 1920x1080x256 x 8 directions), or 16 for workloads of at most 128 labels "
                         "(two / four of those volumes share every wave)")
    ap.add_argument("--mode", default="pairs", choices=["pairs", "directions"],
                    help="N>1: 'pairs' = independent pairs, one per GPU (weak scaling, default); 'directions' = ONE "
                         "volume per step, its passes sharded over the GPUs with an ordered RCCL exchange (strong)")
    args = ap.parse_args()
    w = WORKLOADS[args.workload]

    import torch
    dist = None
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" %
                 (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: there is no CPU path in mgm_amd")
    torch.cuda.set_device(local)
    if world > 1 or args.mode == "directions":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    import mgm_amd
    from mgm_amd import synth
    ctx = mgm_amd.Context(local)
    nx, ny, L = w["nx"], w["ny"], w["dmax"] - w["dmin"] + 1
    # pairs mode: every rank gets its own pair (different seed): independent units, no exchange.
    # directions mode: every rank holds the SAME pair and builds the full cost volume itself.
    if args.batch is None:
        args.batch = 16 if L <= 128 else 12
    B = args.batch if args.mode == "pairs" else 1
    dus, dvs, outs, outcs = [], [], [], []
    for b in range(B):
        u, v, _ = synth.stereo_pair(nx, ny, w["dmin"] * 3 // 4, max(0, w["dmax"] * 3 // 4),
                                    seed=synth.SEED + ((rank * B + b) if args.mode == "pairs" else 0))
        dus.append(ctx.upload_image(u))
        dvs.append(ctx.upload_image(v))
        outs.append(ctx.new_image(nx, ny))
        outcs.append(ctx.new_image(nx, ny))
    du, dv = dus[0], dvs[0]

    def sync_all():
        torch.cuda.synchronize()
        ctx.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.mode == "directions":
        from mgm_amd import dist as mdist

        def step(cv):
            cv = ctx.costvolume_dev(du, dv, w["dmin"], w["dmax"], "none", "census", float("inf"), w["win"], into=cv)
            mdist.aggregate_direction_sharded(ctx, cv, w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, "vfit", dist)
            return cv
    else:
        def step(cvs):
            return run_step(ctx, dus, dvs, w, outs, outcs, cvs)

    cv = None  # the W*H*L volume is allocated once and refilled every step
    for _ in range(max(1, args.warmup)):
        cv = step(cv)
    sync_all()
    ctx.timing(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(cv)  # pairs mode: enqueue only, nothing synchronises inside the timed region
    sync_all()
    dt = time.perf_counter() - t0
    from mgm_amd import shard
    dt = shard.max_over_ranks(dt, dist if world > 1 else None, device="cuda")
    kern = {}
    for name, ms in ctx.timings():
        kern.setdefault(name, []).append(ms)
    ctx.timing(False)
    for x in (cv if isinstance(cv, list) else [cv]):
        x.free()

    if rank == 0:
        cells = float(nx) * ny * L
        if args.mode == "pairs":
            value = shard.job_rate([args.steps * B] * world, dt)  # whole-job aggregate: every rank did K batches of B volumes
        else:
            value = args.steps / dt                            # K volumes, each computed by all ranks together
        nvol_done = args.steps * (world * B if args.mode == "pairs" else 1)
        avg = {k: float(np.mean(vs)) for k, vs in kern.items()}
        per_step = {k: float(np.sum(vs)) / args.steps for k, vs in kern.items()}
        pass_name = "k_pass2" if "k_pass2" in avg else "k_pass"
        # in directions mode rank 0 ran NDIR/world passes and summed ny/world rows per launch
        frac_pass = (mdist.passes_of_rank(w["NDIR"], world, 0)[1] / w["NDIR"]) if args.mode == "directions" else 1.0
        frac_rows = (mdist.row_slabs(ny, world)[0][1] / ny) if args.mode == "directions" else 1.0
        # Aggregation stage = K3 (pass kernel) + K4-K6 (k_wta): the two launches together do what the
        # reference's aggregation loop does; SURVEY.md 8(d): 12 B per cell per direction.
        # With a batch of B volumes per step the pass kernel is launched once (over all of them) and k_wta B times.
        agg_ms = avg[pass_name] + B * avg["k_wta"]
        alg_bytes = 12.0 * w["NDIR"] * cells * (float(B) if args.mode == "pairs" else (2.0 / 3.0) * frac_pass + (1.0 / 3.0) * frac_rows)
        achieved = alg_bytes / (agg_ms * 1e-3) / 1e9
        traffic = None
        prof = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(prof):
            tj = json.load(open(prof))
            if tj.get("workload") == args.workload and tj.get("pairs_per_step") == B:
                traffic = tj.get("aggregation_hbm_bytes_per_step")
        roofline = {"bound": "hbm", "kernel": "%s (one launch per batch of %d volumes) + k_wta (one launch per volume): the 8-direction aggregation" % (pass_name, B),
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                    "avg_launch_ms": {k: avg[k] for k in sorted(avg)},
                    "per_kernel": {
                        pass_name: {"alg_bytes": 8.0 * w["NDIR"] * cells * B,
                                    "GBps": 8.0 * w["NDIR"] * cells * B / (avg[pass_name] * 1e-3) / 1e9},
                        "k_wta": {"alg_bytes": (4.0 * w["NDIR"] + 4.0) * cells + 8.0 * nx * ny,
                                  "GBps": ((4.0 * w["NDIR"] + 4.0) * cells) / (avg["k_wta"] * 1e-3) / 1e9},
                        "k_cost": {"alg_bytes": 4.0 * cells, "GBps": 4.0 * cells / (avg["k_cost"] * 1e-3) / 1e9}}}
        res = {"metric": "disparity-volumes/sec (W*H*L cost volume -> 8-dir MGM -> WTA+vfit)", "value": value,
               "unit": "disparity-volumes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if args.mode == "pairs" else "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%s: %s" % (args.workload, w["desc"]), "W": nx, "H": ny, "L": L, "pairs_per_step": B,
                          "NDIR": w["NDIR"], "TSGM": w["MGM"], "potential": "FH" if w["FH"] else "Hirschmueller",
                          "P1": w["P1"], "P2": w["P2"], "census_win": w["win"], "refine": "vfit",
                          "parallelism": ("independent pairs, one per GPU" if args.mode == "pairs" else
                                          "one volume, %d-way direction sharding, ordered RCCL slab exchange" % world)
                          if world > 1 or args.mode == "directions" else "1 GPU"},
               "roofline": roofline,
               "kernel_ms_per_step": per_step,
               "mcell_updates_per_s": nvol_done * cells * w["NDIR"] / dt / 1e6}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(w)
            res["cpu_baseline"]["host_cpus"] = os.cpu_count()
    ctx.close()
    if dist is not None and dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(res), flush=True)  # the ONE json line, last thing on stdout
    if dist is not None:
        os._exit(0)  # librccl leaves a version banner in the C stdio buffer that would be flushed after the json


if __name__ == "__main__":
    main()
