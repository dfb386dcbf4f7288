#!/usr/bin/env python
"""bench.py -- disparity-volumes/s of the MGM hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3] [--batch B] [--mode pairs|directions] [--extras auto|on|off]

One "step" = one full pass of the hot path over one batch of synthetic stereo pairs whose images are
already resident in HBM: census transform + W*H*L cost volume (K1, K2), all-direction MGM
aggregation (K3), ordered sum + over-count fix + WTA + V-fit (K4-K6).  Outputs stay on the device.

N > 1: one process per GPU.  `python bench.py --gpus N` started WITHOUT a torch.distributed.run environment
launches itself under it (N ranks on 127.0.0.1); started by `python -m torch.distributed.run ... bench.py --gpus N`
it is one of the ranks.  The HEADLINE (`value`) is always the `pairs` mode on the chosen workload: every rank
processes its own independent pairs -- the reference has no cross-pair coupling, so there is no data-path collective
-- and the job rate is all volumes of all ranks over the slowest rank's time (weak scaling).

With the driver's plain command line (no --workload / --batch / --mode) two more legs run after the headline and
land in the SAME json line (`--extras`):
    "cfg5_replicas"   BASELINE config 5: 16 independent 1024x1024x128 pairs per step and GPU, replicas only
    "directions"      BASELINE config 4: ONE 4096x4096x192 volume, its 8 passes sharded over the N GPUs with the ordered
                      row-slab exchange (mgm_amd/dist.py over RCCL: "rccl"; the same plan driven by ONE process through
                      mgm_multi_* with device-to-device peer copies: "peer"), next to the same volume on one GPU
                      ("single", rank 0) -- strong scaling, and a bit-for-bit comparison of the three results.
The extras are guarded: a leg that fails records its error, every exchange is time-boxed (mgm_amd/dist.py), and a
watchdog prints the line with whatever has been measured if the extras overrun `--extras-timeout` (a hung peer cannot
take the headline with it).  `--mode directions` runs the direction-sharded leg alone and makes it the headline.

Workload matrix (BASELINE.json configs; cfg1 is the reference's CPU-runnable case and a parity test, not a bench line):
    --workload cfg3   (default; the configuration the metric is quoted on)   1920x1080x256, census 5x5, -O 8, TSGM 3, FH
    --workload cfg3h  the same with Hirschmueller potentials
    --workload cfg2   1920x1080x128, census 3x3, -O 4, TSGM 2
    --workload cfg4   4096x4096x192, census 5x5, -O 8, TSGM 3 (one volume per step)
    --workload cfg5   1024x1024x128, census 3x3, -O 4, TSGM 2, 16 pairs per step and GPU (throughput mode)

Rank 0 prints ONE JSON line, the last thing on stdout; fields `roofline`, `cpu_baseline` and `parity` are described
in DESIGN.md.  PARITY GATE: the disparity and cost maps of ONE pair of the batch -- its index drawn from the step count --
computed by the LAST timed step are compared, bit for bit, with the CPU oracle run on the same seeded pair in the same
process (and with the reference's own maps where oracle/_ref travelled); a mismatch fails the run (exit code 3).
torch is used for the process group, the barrier and the max-reduction only.
"""
import argparse
import hashlib
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time
from datetime import timedelta

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
    # ---- the variants north_star names beside the headline combination (round 4): per-edge weights, AD, NCC, a label
    # count that runs padded, and one beyond the 512 labels of the second pass-kernel build.  Optional keys: cost
    # (default census), nch (1), aP2 / aThresh (the reference's -aP2 / -aThresh: image-driven weights w_pq), trunc.
    # main() scales P1 and P2 by the channel count (mgm.cc:356-357): the values here are the scaled ones.
    "cfg3w": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=1, P1=2.0, P2=20000.0, aP2=4.0, aThresh=12.0,
                  desc="cfg3 with per-edge weights: -aP2 4 -aThresh 12 (update_costW_trunclinear with w_pq)"),
    "cfg3hw": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=0, P1=8.0, P2=32.0, aP2=4.0, aThresh=12.0,
                   desc="cfg3h with per-edge weights: -aP2 4 -aThresh 12 (update_costW with w_pq)"),
    "cfg3ad": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=1, P1=6.0, P2=60000.0, cost="ad", nch=3,
                   desc="1920x1080 synthetic RGB pair, 256 disparities, -t ad (costs up to 765), -O 8 TSGM=3, FH"),
    "cfg3ad1": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=1, P1=2.0, P2=20000.0, cost="ad", nch=1,
                    desc="1920x1080 synthetic grey pair, 256 disparities, -t ad (costs 0..255), -O 8 TSGM=3, FH"),
    "cfg3ncc": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=1, P1=2.0, P2=20000.0, cost="ncc",
                    desc="1920x1080 synthetic pair, 256 disparities, -t ncc (CENSUS_NCC_WIN=5), -O 8 TSGM=3, FH"),
    "cfg3c7": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=7, NDIR=8, MGM=3, FH=1, P1=2.0, P2=20000.0,
                   desc="1920x1080 synthetic pair, 256 disparities, CENSUS 7x7 (48 descriptor bits: two words, costs in halves: fp32 volumes), -O 8 TSGM=3, FH"),
    "cfg3bt": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=1, P1=2.0, P2=20000.0, cost="btad", nch=1,
                   desc="1920x1080 synthetic grey pair, 256 disparities, -t btad (Birchfield-Tomasi, costs in halves: fp32 volumes), -O 8 TSGM=3, FH"),
    "cfg1s": dict(nx=700, ny=500, dmin=-120, dmax=30, win=3, NDIR=4, MGM=2, FH=0, P1=24.0, P2=96.0, cost="ad", nch=3,
                  desc="BASELINE config 1's shape on a synthetic RGB pair: 700x500, -r -120 -R 30 (151 labels, padded to 192), -t ad, -O 4 TSGM=2"),
    "cfg3L200": dict(nx=1920, ny=1080, dmin=-199, dmax=0, win=5, NDIR=8, MGM=3, FH=1, P1=2.0, P2=20000.0,
                     desc="1920x1080 synthetic pair, 200 disparities (a count off the kernels' widths: runs in 256 label slots), CENSUS 5x5, -O 8 TSGM=3, FH"),
    "cfg3L768": dict(nx=1920, ny=1080, dmin=-767, dmax=0, win=5, NDIR=8, MGM=3, FH=0, P1=8.0, P2=32.0,
                     desc="1920x1080 synthetic pair, 768 disparities (beyond the second build's 512), CENSUS 5x5, -O 8 TSGM=3"),
    # ---- round 5: what had no number at all -- ragged ranges and TSGM_ITER, and the FALL-BACK
    # kernels: weights that are not two-valued, more than 1024 labels, negative penalties, NaN costs, ragged + P2 = inf.
    # Optional keys: ragged (half width of the per-pixel window around the synthetic pair's true disparity: -m/-M images),
    # iter (TSGM_ITER), w8 ("three": three-valued planes, what a free-form w[8][H][W] of matlab/mgm_o.cc looks like to the
    # library), nan (one NaN cost in an uploaded volume: the operand-order-faithful kernel).
    "cfg3r": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=1, P1=2.0, P2=20000.0, ragged=24,
                  desc="cfg3 with range images: per-pixel windows of +-24 labels inside the 256-label hull (-m/-M; 49 of 256 labels exist per pixel)"),
    "cfg3hr": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=0, P1=8.0, P2=32.0, ragged=24,
                   desc="cfg3h with range images: per-pixel windows of +-24 labels inside the 256-label hull"),
    # round 6: what the widened range-proportional layout takes -- windows of 101 labels (128 slots per pixel), and BASELINE config 1's
    # cost (absolute differences of a colour pair: two-byte codes) and update function (TSGM = 2: update_cost2) with range images
    "cfg3r50": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=1, P1=2.0, P2=20000.0, ragged=50,
                    desc="cfg3 with range images: per-pixel windows of +-50 labels (101 of 256 labels exist per pixel; 128 slots per pixel)"),
    "cfg1sr": dict(nx=700, ny=500, dmin=-120, dmax=30, win=3, NDIR=4, MGM=2, FH=0, P1=24.0, P2=96.0, cost="ad", nch=3, ragged=20,
                   desc="BASELINE config 1's shape and cost with range images: 700x500 RGB, -t ad, -O 4 TSGM=2, windows of +-20 labels inside -120..30"),
    "cfg3i2": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=1, P1=2.0, P2=20000.0, iter=2,
                   desc="cfg3 with TSGM_ITER=2: a second winner search in ranges narrowed around the first solution (mgm.cc:377-388)"),
    "cfg3w3": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=1, P1=2.0, P2=20000.0, w8="three",
                   desc="cfg3 with free-form per-edge weights (three values: 1, 2.5, 4): the general weighted kernels, FH"),
    "cfg3hw3": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=0, P1=8.0, P2=32.0, w8="three",
                    desc="cfg3h with free-form per-edge weights (three values): the general weighted kernels, Hirschmueller"),
    "cfg3L1536": dict(nx=1920, ny=1080, dmin=-1535, dmax=0, win=5, NDIR=8, MGM=3, FH=0, P1=8.0, P2=32.0,
                      desc="1920x1080 synthetic pair, 1536 disparities (first build of the pass kernel), CENSUS 5x5, -O 8 TSGM=3"),
    "cfg3L2048": dict(nx=1920, ny=1080, dmin=-2047, dmax=0, win=5, NDIR=8, MGM=3, FH=0, P1=8.0, P2=32.0,
                      desc="1920x1080 synthetic pair, 2048 disparities (first build of the pass kernel), CENSUS 5x5, -O 8 TSGM=3"),
    "cfg3neg": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=0, P1=-2.0, P2=32.0,
                    desc="cfg3h with a negative penalty P1 = -2 (the tags of the queue kernels need E >= 0: first build)"),
    "cfg3nan": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=0, P1=8.0, P2=32.0, nan=1,
                    desc="cfg3h on an UPLOADED volume holding one NaN cost: the operand-order-faithful kernel (k_pass_exact)"),
    "cfg3rinf": dict(nx=1920, ny=1080, dmin=-255, dmax=0, win=5, NDIR=8, MGM=3, FH=0, P1=8.0, P2=float("inf"), ragged=24,
                     desc="cfg3hr with P2 = inf: all-INF slabs, INF - INF: the operand-order-faithful kernel (k_pass_exact)"),
}
# Set ONLY by a test driver that imports this module (tests/run_bench_stub.py) to drive the launcher, the rendezvous and the
# JSON contract on CPU ranks; nothing in the environment or on the command line can set it, and the line such a run prints
# says `"data": "stub (no device work)"`.
TEST_CONTEXT_FACTORY = None
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0  # what a float4 copy kernel reaches on this part (MI355X_MICROARCH.md: 6.29 TB/s): the practical ceiling
PARITY_MAX_CELLS = 1.2e9  # the in-run oracle takes whole volumes up to this size (cfg4: see tests/test_gpu_fullsize.py)


def labels_of(w):
    return w["dmax"] - w["dmin"] + 1


def pair_of(w, seed_offset=0):
    """The synthetic pair every leg of the run uses for a given index (GPU steps, parity oracle, CPU baseline)."""
    from mgm_amd import synth
    return synth.stereo_pair(w["nx"], w["ny"], w["dmin"] * 3 // 4, max(0, w["dmax"] * 3 // 4), seed=synth.SEED + seed_offset,
                             nch=w.get("nch", 1))


def cost_of(w):
    return w.get("cost", "census")


def trunc_of(w):
    return float(w.get("trunc", float("inf")))


def weighted(w):
    return float(w.get("aP2", 1.0)) != 1.0  # (mgm.cc:372: the weights are computed -- and used -- only when -aP2 != 1)


def strip_comments(src):
    """C / C++ source without its comments and with every run of white space as one blank (string and character
    literals kept as they are): what a compiler sees of it, give or take line numbers."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c in "\"'":  # a literal: copy up to its closing quote
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == "\\" else 1
            out.append(src[i:j + 1])
            i = j + 1
        elif src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def kernel_source_hash(csrc=None):
    """Identity of the kernels a committed PMC summary was measured on (profiles/*_traffic.json): the sources under
    mgm_amd/csrc without their comments and line breaks, so that rewording a comment does not orphan a measurement."""
    h = hashlib.sha256()
    d = csrc or os.path.join(ROOT, "mgm_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode() + b"\0" + strip_comments(open(os.path.join(d, f), encoding="utf-8").read()).encode() + b"\0")
    return h.hexdigest()[:16]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cost_bytes(w):
    """Bytes per label the costs of this workload travel as between K2, K3 and k_wta (mgm_api.hip, costvolume_fill): one for
    single-word census costs and grey-level absolute differences, two for absolute differences of colour pairs and squared
    differences (round 4), four (fp32) for everything else; weighted launches other than the two-valued compact ones read fp32."""
    L = labels_of(w)
    if os.environ.get("MGM_HIP_C8", "1") == "0":
        return 4.0
    if L in (64, 128, 192, 256, 384, 512, 768, 1024) or L < 1024:  # (other label counts run padded to the next of these)
        if cost_of(w) == "census" and w.get("nch", 1) * (w["win"] ** 2 - 1) > 32:
            return 4.0  # (a descriptor of several words: costs in halves or thirds of bit counts)
        if cost_of(w) == "census" or (cost_of(w) == "ad" and w.get("nch", 1) == 1):
            return 1.0
        if cost_of(w) in ("ad", "sd") and L <= 512 and not weighted(w):
            return 2.0
    return 4.0


# ---- the CPU legs (rank 0 only; never inside a timed GPU region) --------------------------------------------------------
def special_inputs(w, pair):
    """What pairs_leg feeds the device beside the pair, rebuilt on the host for the CPU legs (rank 0: seed offsets = pair):
    range images (ragged), free-form weight planes (w8), the uploaded volume with its NaN (nan)."""
    from mgm_amd import synth
    u, v, gt = pair_of(w, pair)
    x = {"u": u, "v": v, "dminI": None, "dmaxI": None, "w8": None, "vol": None}
    if w.get("ragged"):
        g = gt.astype(np.float32)
        x["dminI"], x["dmaxI"] = np.clip(g - w["ragged"], w["dmin"], w["dmax"]), np.clip(g + w["ragged"], w["dmin"], w["dmax"])
    if w.get("w8") == "three":
        rng = np.random.default_rng(77 + pair)
        x["w8"] = rng.choice(np.array([1.0, 2.5, 4.0], np.float32), size=(8, w["ny"], w["nx"]), p=[0.7, 0.2, 0.1])
    if w.get("nan"):
        vol = synth.raw_volume(w["nx"], w["ny"], labels_of(w), seed=5 + pair)
        vol[vol.shape[0] // 2, vol.shape[1] // 2, 7] = np.nan
        x["vol"] = vol
    return x


def iter_ranges(w, disp):
    """main()'s update_dmin_dmax + remove_nonfinite (mgm.cc:386-388) on flat range images, by the compiled reference's own
    function (oracle/_ref/libmgm_refpost.so); None where that library did not travel."""
    from oracle.oracle import RefPost
    if not RefPost.available():
        return None
    lo = np.full((w["ny"], w["nx"]), w["dmin"], np.float32)
    hi = np.full((w["ny"], w["nx"]), w["dmax"], np.float32)
    return RefPost().update_ranges(disp, lo, hi, 3, 2)


def oracle_whole_volume(w, threads, pair=0):
    """Pair `pair` through the CPU oracle (oracle/mgm_oracle.c), whole volume, `threads` OpenMP threads (the reference
    parallelises each diagonal of a pass the same way, mgm_core.cc:505-579).  Returns (disp, cost, seconds).  Round 6: also the
    workloads with range images (orc_mgm_ranged, pinned on the reference's mgm() with range images), free-form weights, an
    uploaded volume holding a NaN, and TSGM_ITER = 2 (the second call of mgm() with narrowed S ranges)."""
    from oracle.oracle import Oracle, int_ranges
    orc = Oracle(threads=threads)
    x = special_inputs(w, pair)
    u, v = x["u"], x["v"]
    t0 = time.perf_counter()
    if x["dminI"] is not None:
        lo, hi = int_ranges(x["dminI"], x["dmaxI"])
        C = orc.costvolume_ranged(u, v, lo, hi, w["dmin"], w["dmax"], "none", cost_of(w), trunc_of(w), w["win"])
        S, o, c = orc.mgm_ranged(C, w["dmin"], lo, hi, w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, x["w8"])
        ro, rc = orc.refine_ranged(S, w["dmin"], lo, hi, "vfit", o, c)
        return ro, rc, time.perf_counter() - t0
    C = x["vol"] if x["vol"] is not None else orc.costvolume(u, v, w["dmin"], w["dmax"], "none", cost_of(w), trunc_of(w), w["win"])
    w8 = orc.weights(u, w["aP2"], w["aThresh"]) if weighted(w) else x["w8"]
    S, o, c = orc.mgm(C, w["dmin"], w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, w8)
    ro, rc = orc.refine(S, w["dmin"], "vfit", o, c)
    for _ in range(1, int(w.get("iter", 1))):  # mgm.cc:377-388: CC keeps its ranges, mgm() is called with the narrowed images
        r = iter_ranges(w, ro)
        if r is None:
            raise RuntimeError("TSGM_ITER > 1 needs oracle/_ref/libmgm_refpost.so (update_dmin_dmax)")
        slo, shi = int_ranges(*r)
        ny, nx = slo.shape
        flo, fhi = np.full((ny, nx), w["dmin"], np.int32), np.full((ny, nx), w["dmax"], np.int32)
        shmin, shmax = int(slo.min()), int(shi.max())
        S, o, c = orc.mgm_ranged(C, w["dmin"], flo, fhi, w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, w8, (slo, shi, shmin, shmax))
        ro, rc = orc.refine_ranged(S, shmin, slo, shi, "vfit", o, c)
    return ro, rc, time.perf_counter() - t0


def reference_whole_volume(w, threads, want=None, pair=0):
    """The REAL reference (oracle/_ref/libmgm_ref.so: gfacciol/mgm compiled from its own sources behind
    oracle/ref_harness.cc) on pair 0: allocate_and_fill_sgm_costvolume + mgm() + subpixel_refinement_sgm, timed INSIDE
    the harness around those three calls (the dense <-> Dvec container copies of the harness are not the reference's
    work).  Returns seconds, or None where the library is not there / the census window cannot be set.  Round 6: with range
    images too (the reference on the SAME -m/-M images: what a ragged workload's CPU baseline is)."""
    from oracle.oracle import Oracle, Reference
    if not Reference.available():
        return None
    if w.get("iter", 1) > 1 or w.get("nan"):
        return None  # (TSGM_ITER lives in main(); a NaN cost leaves the reference's labels undefined)
    os.environ["CENSUS_NCC_WIN"] = str(w["win"])  # a smart parameter of the reference, cached on first use per process
    ref = Reference()
    if ref.census_win() != w["win"] or not hasattr(ref.lib, "ref_seconds"):
        return None
    Oracle(threads=threads)  # (sets the OpenMP thread count of the process: both libraries share libgomp)
    x = special_inputs(w, pair)
    u, v = x["u"], x["v"]
    os.environ["USE_TRUNCATED_LINEAR_POTENTIALS"] = "1" if w["FH"] else "0"
    if x["dminI"] is not None:
        if not ref.has_ranged():
            return None
        C = ref.costvolume_ranged(u, v, x["dminI"], x["dmaxI"], w["dmin"], w["dmax"], "none", cost_of(w), trunc_of(w))
        t = ref.seconds()
        S, o, c = ref.mgm_ranged(C, w["dmin"], x["dminI"], x["dmaxI"], w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, x["w8"])
        t += ref.seconds()
        o = np.where(np.isfinite(c), o, x["dminI"]).astype(np.float32)  # (a pixel without a finite S: the reference's label is uninitialised)
        ro, rc = ref.refine_ranged(S, w["dmin"], x["dminI"], x["dmaxI"], "vfit", o, c)
        t += ref.seconds()
        ro = np.where(np.isfinite(c), ro, np.nan).astype(np.float32)
    else:
        C = ref.costvolume(u, v, w["dmin"], w["dmax"], "none", cost_of(w), trunc_of(w))
        t = ref.seconds()
        w8 = ref.weights(u, w["aP2"], w["aThresh"]) if weighted(w) else x["w8"]  # (compute_mgm_weights: negligible, not timed)
        S, o, c = ref.mgm(C, w["dmin"], w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, w8)
        t += ref.seconds()
        ro, rc = ref.refine(S, w["dmin"], "vfit", o, c)
        t += ref.seconds()
    if want is not None:
        want["o"], want["c"] = ro, rc
    return t


def cpu_baseline(w, whole_first, threads, seconds_target=12.0, pair=0):
    """The CPU path timed on the GPU box's host cores (rank 0, N = 1): bounded samples of the same workload.
      reference  the REAL reference's three calls on the whole pair, `threads` OpenMP threads, median of 3 (where
                 oracle/_ref/libmgm_ref.so travelled to this box)
      port       oracle/mgm_oracle.c: the whole pair on `threads` threads, median of 3 (the parity leg's run is the first of
                 them), and ONE thread on a row band sized to ~`seconds_target` s, extrapolated linearly in rows
    The top-level value is the reference's where it exists (kind "reference"), else the port's best."""
    from mgm_amd import synth
    from oracle.oracle import Oracle
    nx, L = w["nx"], labels_of(w)
    cells = float(nx) * w["ny"] * L
    res = {"unit": "disparity-volumes/s", "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
    # -- port, all threads: median of 3 whole volumes
    runs = [whole_first] if whole_first is not None else []
    while len(runs) < 3:
        runs.append(oracle_whole_volume(w, threads, pair)[2])
    med = float(np.median(runs))
    port = {"value": 1.0 / med, "cores": threads, "kind": "port", "runs_s": [round(t, 3) for t in runs],
            "sample": "one whole %dx%dx%d volume (cost volume + %d-direction mgm + vfit), oracle/mgm_oracle.c on %d OpenMP threads, "
                      "median of %d runs" % (nx, w["ny"], L, w["NDIR"], threads, len(runs)),
            "mcell_updates_per_s": cells * w["NDIR"] / med / 1e6}
    # -- port, one thread, on a band of rows
    orc = Oracle(threads=1)

    def band(rows):
        from oracle.oracle import int_ranges
        u, v, gt = synth.stereo_pair(nx, rows, w["dmin"] * 3 // 4, max(0, w["dmax"] * 3 // 4), nch=w.get("nch", 1))
        t0 = time.perf_counter()
        if w.get("ragged"):
            g = gt.astype(np.float32)
            lo, hi = int_ranges(np.clip(g - w["ragged"], w["dmin"], w["dmax"]), np.clip(g + w["ragged"], w["dmin"], w["dmax"]))
            C = orc.costvolume_ranged(u, v, lo, hi, w["dmin"], w["dmax"], "none", cost_of(w), trunc_of(w), w["win"])
            S, o, c = orc.mgm_ranged(C, w["dmin"], lo, hi, w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, None)
            orc.refine_ranged(S, w["dmin"], lo, hi, "vfit", o, c)
            return time.perf_counter() - t0
        C = orc.costvolume(u, v, w["dmin"], w["dmax"], "none", cost_of(w), trunc_of(w), w["win"])
        w8 = orc.weights(u, w["aP2"], w["aThresh"]) if weighted(w) else None
        S, o, c = orc.mgm(C, w["dmin"], w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, w8)
        orc.refine(S, w["dmin"], "vfit", o, c)
        return time.perf_counter() - t0

    if not any(w.get(k) for k in ("iter", "w8", "nan")):  # (those inputs are not a function of a row band alone)
        per_row = band(8) / 8  # calibrate on a thin band, then size the sample
        rows = int(max(16, min(w["ny"], seconds_target / per_row)))
        dt = band(rows)
        port["one_thread"] = {"value": (rows / w["ny"]) / dt, "cores": 1,
                              "sample": "%dx%dx%d band (%d of %d rows), %.1f s, extrapolated linearly in rows" % (nx, rows, L, rows, w["ny"], dt),
                              "mcell_updates_per_s": nx * rows * L * w["NDIR"] / dt / 1e6}
    res["port"] = port
    # -- the reference itself
    ref_runs, ref_out = [], {}
    try:
        for _ in range(3):
            t = reference_whole_volume(w, threads, ref_out if not ref_runs else None, pair)
            if t is None:
                break
            ref_runs.append(t)
    except Exception as e:  # noqa: BLE001 -- a baseline that cannot run is reported, it does not fail the bench
        res["reference_error"] = repr(e)[:300]
    if ref_runs:
        rmed = float(np.median(ref_runs))
        res["reference"] = {"value": 1.0 / rmed, "cores": threads, "kind": "reference", "runs_s": [round(t, 3) for t in ref_runs],
                            "sample": "one whole %dx%dx%d volume: the reference's allocate_and_fill_sgm_costvolume + mgm() + "
                                      "subpixel_refinement_sgm (oracle/_ref/libmgm_ref.so, compiled from gfacciol/mgm's sources, -O3 "
                                      "-fopenmp) on %d OpenMP threads, median of %d runs, timed around the three calls"
                                      % (nx, w["ny"], L, threads, len(ref_runs)),
                            "mcell_updates_per_s": cells * w["NDIR"] / rmed / 1e6}
    top = res.get("reference") or port
    res.update({"value": top["value"], "cores": top["cores"], "kind": top["kind"], "sample": top["sample"]})
    return res, ref_out


def nd(a, b):
    """Differing float32 words of two arrays, NaN == NaN."""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return int(np.sum((a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))))


# ---- launcher -----------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a rendezvous environment: become the launcher of N ranks of the script that was
    started (this file -- or the test driver that imported it, tests/run_bench_stub.py)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(sys.argv[0])] + sys.argv[1:]
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


class OneLine:
    """The ONE json line of rank 0, printed exactly once -- by the main thread at the end, or by the watchdog if the
    guarded extras overrun (every rank then leaves through os._exit: a process group with a transfer pending on the device
    cannot be torn down in an orderly way)."""

    def __init__(self, rank):
        self.rank, self.lock, self.done, self.res, self.code = rank, threading.Lock(), False, None, 0
        self.guard_deadline = None

    def emit(self, hard, note=None):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.rank == 0 and self.res is not None:
                if note:
                    self.res.setdefault("extras_note", note)
                sys.stdout.flush()
                print(json.dumps(self.res), flush=True)
        if hard:
            os._exit(self.code)

    def guard(self, seconds):
        """From now on the process has `seconds` to call emit(); past that the watchdog does."""
        self.guard_deadline = time.monotonic() + seconds

        def run():
            while not self.done:
                if time.monotonic() > self.guard_deadline + (0.0 if self.rank == 0 else 2.0):
                    self.emit(True, "watchdog: the extras did not finish within %.0f s; the line holds what had been measured" % seconds)
                time.sleep(0.25)
        threading.Thread(target=run, daemon=True).start()
        # torch.distributed.run sends SIGTERM to the surviving ranks when one dies: rank 0 still prints what it has.  The
        # C-level handler writes the signal number to a pipe at once, whatever the main thread is blocked in.
        try:
            r, wfd = os.pipe()
            os.set_blocking(wfd, False)
            signal.signal(signal.SIGTERM, lambda *_: None)
            signal.set_wakeup_fd(wfd)

            def on_term():
                os.read(r, 1)
                self.emit(True, "SIGTERM during the extras (a rank died): the line holds what had been measured")
            threading.Thread(target=on_term, daemon=True).start()
        except (ValueError, OSError):
            pass


# ---- legs -------------------------------------------------------------------------------------------------------------
class Env:
    """What every leg needs: the process's rank, its groups and its context."""

    def __init__(self, args, stub):
        self.args, self.stub = args, stub
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist, self.ctrl, self.torch = None, None, None

    def setup(self):
        # dmabuf IPC: RCCL between processes (and any device-memory sharing) needs it on this driver; the launcher exports it
        # already, a rank started some other way gets it here, before the HIP runtime comes up
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        self.torch = torch
        args = self.args
        if self.world != args.gpus:
            sys.exit("bench.py --gpus %d runs inside a torch.distributed.run job of %d ranks" % (args.gpus, self.world))
        if not self.stub:
            if not torch.cuda.is_available():
                sys.exit("bench.py needs an MI355X: there is no CPU path in mgm_amd")
            torch.cuda.set_device(self.local)
        if self.world > 1 or args.mode in ("directions", "pairs2"):
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if self.stub:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world, timeout=timedelta(seconds=600))
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local))
                if self.world > 1:
                    # control plane of the guarded legs: CPU collectives (agreement, "rank 0 is busy" barriers) that neither
                    # occupy a GPU nor hang without a time-out
                    self.ctrl = dist.new_group(backend="gloo", timeout=timedelta(seconds=900))
            self.dist = dist
        self.n_ranks = self.dist.get_world_size() if self.dist is not None else 1  # the world RCCL actually initialised
        if self.stub:
            self.ctx = TEST_CONTEXT_FACTORY(self.local)
        else:
            import mgm_amd
            self.ctx = mgm_amd.Context(self.local)
            # a long-lived context: let it pick the fastest of a few physical placements of every new workspace (round 5:
            # identical launches sit on plateaus up to 16 % apart depending on the pages hipMalloc handed out; the tuning runs in
            # the warm-up steps, never in a timed region -- a timed block only ever reuses the workspace of the steps before it)
            self.ctx.set_placement_tries(int(os.environ.get("MGM_BENCH_PLACE_TRIES", "4")))

    def sync_all(self):
        if not self.stub:
            self.torch.cuda.synchronize()
        self.ctx.synchronize()
        if self.world > 1:
            self.dist.barrier()
            if not self.stub:
                self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        from mgm_amd import shard
        return shard.max_over_ranks(x, self.dist if self.world > 1 else None, device="cpu" if self.stub else "cuda")

    def ctrl_barrier(self):
        if self.world > 1:
            self.dist.barrier(group=self.ctrl)  # (stub: the default group is gloo)


def pairs_leg(env, w, B, steps, warmup, repeats, keep=False, pipeline=1):
    """Independent pairs, B per step and GPU: W warm-up steps, then EXACTLY K timed steps between barrier + synchronize
    brackets, MAX over ranks.  Returns the measurement (and, with `keep`, the device outputs of the last step).
    pipeline = D >= 2: the context gathers the aggregation calls of D consecutive steps into one batched launch
    (mgm_ctx_set_pipeline); the steps then use D sets of volumes and output images in turn -- what a caller with a stream of
    pairs would do -- and all work of the K steps still happens inside the timed region (the closing synchronisation runs
    whatever is still deferred)."""
    ctx, rank = env.ctx, env.rank
    nx, ny = w["nx"], w["ny"]
    D = max(1, pipeline)
    if D > 1:
        ctx.set_pipeline(D)
    dus, dvs = [], []
    for b in range(B):
        u, v, _ = pair_of(w, rank * B + b)  # every rank gets its own pairs (different seeds): independent units, no exchange
        dus.append(ctx.upload_image(u))
        dvs.append(ctx.upload_image(v))
    sets = [{"cv": [None] * B, "w8": [None] * B, "o": [ctx.new_image(nx, ny) for _ in range(B)], "c": [ctx.new_image(nx, ny) for _ in range(B)]}
            for _ in range(D)]  # (the W*H*L volumes and the weight planes are allocated by a set's first step and refilled after)
    count = [0]
    # round 5 workloads: range images (-m/-M), free-form weights, an uploaded volume with a NaN -- resident inputs, like the pairs
    rlo, rhi, w8fixed, nanvol, ilo, ihi = [], [], [], [], [], []
    for b in range(B):
        if w.get("ragged"):
            gt = pair_of(w, rank * B + b)[2].astype(np.float32)
            rlo.append(ctx.upload_image(np.clip(gt - w["ragged"], w["dmin"], w["dmax"])[None]))
            rhi.append(ctx.upload_image(np.clip(gt + w["ragged"], w["dmin"], w["dmax"])[None]))
        if w.get("w8") == "three":
            rng = np.random.default_rng(77 + rank * B + b)
            w8fixed.append(ctx.upload_image(rng.choice(np.array([1.0, 2.5, 4.0], np.float32), size=(8, ny, nx), p=[0.7, 0.2, 0.1])))
        if w.get("nan"):
            from mgm_amd import synth
            vol = synth.raw_volume(*((8, 8) if env.stub else (nx, ny)), labels_of(w), seed=5 + b)  # (the test double computes nothing)
            vol[vol.shape[0] // 2, vol.shape[1] // 2, 7] = np.nan
            nanvol.append(ctx.upload_volume(vol, w["dmin"]))
        if w.get("iter", 1) > 1:
            ilo.append(ctx.new_image(nx, ny))
            ihi.append(ctx.new_image(nx, ny))
    flat_lo = np.full((1, ny, nx), w["dmin"], np.float32)
    flat_hi = np.full((1, ny, nx), w["dmax"], np.float32)

    def step():
        # one step = one batch: the cost volume (and, weighted workloads, the edge weights) of every pair, ONE pass launch
        # over the batch, WTA + vfit per volume
        st = sets[count[0] % D]
        count[0] += 1
        if nanvol:
            st["cv"] = nanvol  # (uploaded once: the leg measures the aggregation of a volume that holds a NaN)
        elif rlo:
            st["cv"] = [ctx.costvolume_ranged_dev(du, dv, lo, hi, w["dmin"], w["dmax"], "none", cost_of(w), trunc_of(w), w["win"], into=cv)
                        for du, dv, lo, hi, cv in zip(dus, dvs, rlo, rhi, st["cv"])]
        else:
            st["cv"] = [ctx.costvolume_dev(du, dv, w["dmin"], w["dmax"], "none", cost_of(w), trunc_of(w), w["win"], into=cv)
                        for du, dv, cv in zip(dus, dvs, st["cv"])]
        if weighted(w):
            for b in range(B):
                st["w8"][b] = ctx.weights_dev(dus[b], w["aP2"], w["aThresh"], into=st["w8"][b])
        w8s = st["w8"] if weighted(w) else (w8fixed if w8fixed else None)
        ctx.aggregate_batch_dev(st["cv"], w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, w8s, "vfit",
                                st["o"], st["c"], want_S=False)
        for it in range(1, int(w.get("iter", 1))):  # main()'s loop, mgm.cc:377-388: ranges narrowed around the solution, winner search again
            for b in range(B):
                if it == 1:
                    ilo[b].update(flat_lo)
                    ihi[b].update(flat_hi)
                ctx.update_ranges_dev(st["o"][b], ilo[b], ihi[b], 3, 2)
                if B > 1:  # (the windowed search works on the Lr volumes of the context's LAST aggregation: one volume)
                    raise RuntimeError("TSGM_ITER > 1 workloads run one pair per step")
                ctx.wta_windowed_dev(st["cv"][b], w["NDIR"], 1, "vfit", ilo[b], ihi[b], st["o"][b], st["c"][b])
        return st

    def timed_block():
        env.sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()  # enqueue only: nothing synchronises inside the timed region
        env.sync_all()
        return env.max_over_ranks(time.perf_counter() - t0)

    for _ in range(max(1, warmup, D)):
        step()
    env.sync_all()
    ctx.timing(True)
    ctx.timing_reset()
    dt = timed_block()  # the reported measurement: exactly K steps between two barrier + synchronize brackets
    kern = {}
    for name, ms in ctx.timings():
        kern.setdefault(name, []).append(ms)
    ctx.timing(False)
    # further blocks of K steps: the same measurement again (stability), and a GPU phase long enough for a coarse
    # utilisation sampler to see
    reps = repeats if repeats is not None else int(min(8, max(0, np.ceil(10.0 / max(dt, 1e-3)) - 1)))
    rep_dt = [timed_block() for _ in range(reps)]
    m = {"dt": dt, "rep_dt": rep_dt, "avg": {k: float(np.mean(vs)) for k, vs in kern.items()},
         "per_step": {k: float(np.sum(vs)) / steps for k, vs in kern.items()}, "B": B, "steps": steps, "pipeline": D}
    last = sets[(count[0] - 1) % D]  # the set the LAST step wrote
    ctx.synchronize()
    if D > 1:
        ctx.set_pipeline(1)
    if keep:
        m["outs"], m["outcs"] = last["o"], last["c"]
    freed = set()
    for st in sets:
        for h in st["cv"] + st["w8"] + ([] if (keep and st is last) else st["o"] + st["c"]):
            if h is not None and id(h) not in freed:
                freed.add(id(h))
                h.free()
    for h in dus + dvs + rlo + rhi + w8fixed + ilo + ihi + [v for v in nanvol if id(v) not in freed]:
        h.free()
    return m


def roofline_of(w, B, avg, workload, step_ms=None, per_step=None):
    """the aggregation = K3 (pass kernel, one launch per batch) + K4-K6 (k_wta, one launch per volume);
    ALGORITHMIC bytes = 12 B per cell per direction (read C, read S, write S in fp32).  `frac` is that figure over the
    measured launch times; `frac_counter` prices the same time against the HBM bytes the PMC counters saw (committed summary
    of THESE kernels, else null); per-kernel figures use the bytes each kernel's DATA FORMAT implies (compact costs: C is
    one byte per label), so that none of them can exceed the peak."""
    nx, ny, L = w["nx"], w["ny"], labels_of(w)
    cells = float(nx) * ny * L
    pass_name = next((k for k in ("k_pass2", "k_pass", "k_pass_rel", "k_pass_exact") if k in avg), "k_pass2")
    agg_ms = avg[pass_name] + B * avg["k_wta"]
    if pass_name == "k_pass_exact" and per_step:  # (one launch per diagonal and pass: the step's SUM, not an average launch)
        agg_ms = per_step[pass_name] + per_step["k_wta"]
    if step_ms is not None:
        # pipelined steps: one pass launch serves D steps, so per-launch durations are not per-step figures; the aggregation
        # is priced against the WHOLE step's wall time instead -- an upper bound of its time (the step also holds K1 / K2)
        agg_ms = step_ms
    rel = pass_name == "k_pass_rel"
    hull_cells = cells
    if rel:  # the range-proportional kernels walk only the labels each pixel OWNS: those are the cells 8(d)'s 12 B are due on
        cells = float(nx) * ny * (2 * w["ragged"] + 1)
    alg_bytes = 12.0 * w["NDIR"] * cells * B
    achieved = alg_bytes / (agg_ms * 1e-3) / 1e9
    cbytes = cost_bytes(w)
    traffic, traffic_src = None, None
    for prof in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")), reverse=True):
        tj = json.load(open(os.path.join(ROOT, "profiles", prof)))
        if tj.get("workload") == workload and tj.get("pairs_per_step") == B and tj.get("kernel_source_sha") == kernel_source_hash():
            traffic, traffic_src = tj.get("aggregation_hbm_bytes_per_step"), "profiles/" + prof
            break
    fmt = {pass_name: (cbytes + 4.0) * w["NDIR"] * cells * B,          # reads C once per direction, writes one Lr volume per direction
           "k_wta": (4.0 * w["NDIR"] + cbytes) * cells + 8.0 * nx * ny,  # reads NDIR Lr volumes + C, writes two W*H maps
           "k_cost": cbytes * cells}                                     # writes C (the images are negligible)
    if rel:  # 64 / 128 cost slots of one / two bytes and as many fp32 Lr slots per pixel and direction, plus the 16-byte window record
        slots = (64.0 if 2 * w["ragged"] + 1 <= 62 else 128.0) * nx * ny
        rcb = 1.0 if (cost_of(w) == "census" or (cost_of(w) == "ad" and w.get("nch", 1) == 1)) else 2.0
        fmt = {pass_name: (rcb + 4.0) * w["NDIR"] * slots * B + 16.0 * nx * ny * w["NDIR"] * B,
               "k_wta": (4.0 * w["NDIR"] + rcb) * slots + 24.0 * nx * ny, "k_cost": cbytes * hull_cells}
    per_kernel = {k: {"format_bytes": b, "GBps": b / (avg[k] * 1e-3) / 1e9, "frac": b / (avg[k] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                  for k, b in fmt.items() if k in avg}
    # How to READ the fraction: `frac` prices SURVEY 8(d)'s 12 B per cell and direction against the 8 TB/s spec
    # peak -- but the kernels move FEWER bytes than that (compact costs; the ordered sum read once), so `frac` can exceed
    # what any copy kernel reaches on the part.  `moved_over_algorithmic` = counter bytes / algorithmic bytes (< 1: nothing is
    # re-read), `frac_of_achievable` = counter GB/s over the 6.3 TB/s a float4 copy reaches, `saturated` = that is >= 0.9:
    # no further bandwidth to be had for this launch shape.  Without a counter profile of these very kernel sources the
    # per-kernel FORMAT bytes stand in for the counters (`moved_basis` says which).
    fmt_total = sum(b * (B if k == "k_wta" else 1) for k, b in fmt.items() if k in avg and k != "k_cost")
    moved = traffic if traffic else fmt_total
    moved_gbs = moved / (agg_ms * 1e-3) / 1e9
    extra = {}
    if w.get("ragged"):  # what a range-proportional layout would be priced on: the labels that EXIST (mgm_costvolume.h:275-299)
        own = float(nx) * ny * (2 * w["ragged"] + 1)
        extra = {"existing_cells_over_hull_cells": own / hull_cells, "frac_range_proportional": 12.0 * w["NDIR"] * own * B / (agg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "frac_dense_hull_equivalent": 12.0 * w["NDIR"] * hull_cells * B / (agg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "note": ("k_pass_rel walks only the labels each pixel owns: frac prices THOSE cells; frac_dense_hull_equivalent is what a dense-hull "
                          "kernel would have to reach for the same time (may exceed 1)") if rel else
                         "frac prices the dense HULL (what the kernels walk); frac_range_proportional prices only the labels each pixel owns"}
    return {**extra, "bound": "hbm", "kernel": "%s (one launch per batch of %d volumes) + k_wta (one launch per volume): the %d-direction aggregation" % (pass_name, B, w["NDIR"]),
            "moved_over_algorithmic": moved / alg_bytes, "moved_basis": "pmc counters" if traffic else "format bytes of the kernels' data layout",
            "frac_of_achievable": moved_gbs / HBM_ACHIEVABLE_GBS, "achievable_peak": HBM_ACHIEVABLE_GBS,
            "saturated": bool(moved_gbs / HBM_ACHIEVABLE_GBS >= 0.9),
            "time_basis": "sum of the average launch durations (HIP events on the kernels' streams)" if step_ms is None else
                          "wall time per step (pipelined: one pass launch serves several steps)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_src,
            "achieved_counter": (traffic / (agg_ms * 1e-3) / 1e9) if traffic else None,
            "frac_counter": (traffic / (agg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "algorithmic_bytes_per_launch": alg_bytes, "aggregation_ms": agg_ms,
            "avg_launch_ms": {k: avg[k] for k in sorted(avg)}, "per_kernel": per_kernel}


def roofline_digest(res, vres, wname):
    """the driver's record keeps `roofline` but drops `variants` -- so the figures north_star's target sentence is
    about (ONE volume; a pair = two volumes, what `mgm u v out` launches) and the weakest legs go INTO `roofline`."""
    rf = res.get("roofline")
    if not isinstance(rf, dict):
        return

    def pick(key):
        v = vres.get(key)
        if not isinstance(v, dict) or "roofline_frac" not in v:
            return None
        return {"volumes_per_s": v["value"], "frac": v["roofline_frac"], "frac_counter": v.get("roofline_frac_counter"),
                "frac_of_achievable": v.get("frac_of_achievable"), "k3_ms": v.get("k3_ms"), "wta_ms": v.get("wta_ms"), "pass_kernel": v.get("pass_kernel"),
                **({"parity": v["parity"].get("status")} if isinstance(v.get("parity"), dict) else {})}

    rf["single_volume"] = pick("%s x1" % wname)  # (the same workload as the headline, ONE volume per launch)
    rf["pair"] = pick("%s x2" % wname)
    ranked = sorted(((k, v) for k, v in vres.items() if isinstance(v, dict) and isinstance(v.get("roofline_frac"), float)
                     and "on the dense hull" not in k and "pipeline" not in k), key=lambda kv: kv[1]["roofline_frac"])
    rf["worst_variants"] = [{"variant": k, **pick(k)} for k, _ in ranked[:3]]
    rf["ragged_variants"] = {k: pick(k) for k in ("cfg3r x1", "cfg3r x2", "cfg3r x4", "cfg3hr x1", "cfg3hr x4", "cfg3r50 x1", "cfg1sr x2") if pick(k)}


def replicas_leg(env, name, steps, warmup):
    """BASELINE config 5: replicas only -- 16 independent pairs per step and GPU, no communication."""
    w = WORKLOADS[name]
    B = 16
    m = pairs_leg(env, w, B, steps, warmup, 0)
    from mgm_amd import shard
    rf = roofline_of(w, B, m["avg"], name)
    pass_name = next((k for k in ("k_pass2", "k_pass", "k_pass_rel", "k_pass_exact") if k in m["avg"]), "k_pass2")
    return {"workload": "%s: %s" % (name, w["desc"]), "value": shard.job_rate([steps * B] * env.n_ranks, m["dt"]), "unit": "disparity-volumes/s",
            "n_gpus": env.n_ranks, "pairs_per_step_per_gpu": B, "steps": steps, "warmup": warmup, "ms_per_step": m["dt"] / steps * 1e3,
            "scaling": "weak", "parallelism": "replicas only: independent pairs, no data-path collective",
            "k2_ms": m["avg"].get("k_cost"), "k3_ms": m["avg"][pass_name], "wta_ms": m["avg"]["k_wta"], "roofline_frac": rf["frac"]}


def directions_legs(env, name, steps, warmup, timeout_s, transports=("single", "peer", "rccl"), res=None):
    """BASELINE config 4: ONE volume, its passes sharded by direction over the ranks.  Sub-legs, each guarded:
      single  the whole aggregation on rank 0's GPU (the N = 1 figure of the strong-scaling curve, and the result the
              others are compared with, bit for bit)
      peer    rank 0's process drives ALL N GPUs through mgm_multi_* with device-to-device copies for the exchange
      rccl    one process per GPU, mgm_amd/dist.py: agree, then the ordered slab exchange over RCCL, time-boxed
    Every rank must call this (the control barriers keep the idle ranks off the GPUs while rank 0 works alone).  `res`: the
    dictionary to fill -- the caller's, already hanging in the json line, so that whatever has been measured when a later
    sub-leg hangs is what the watchdog prints."""
    from mgm_amd import dist as mdist
    w = WORKLOADS[name]
    ctx, rank, world, torch, dist = env.ctx, env.rank, env.n_ranks, env.torch, env.dist
    nx, ny, L, NDIR = w["nx"], w["ny"], labels_of(w), w["NDIR"]
    res = {} if res is None else res
    res.update({"workload": "%s: %s" % (name, w["desc"]), "ranks": world, "unit": "disparity-volumes/s", "scaling": "strong",
                "steps": steps, "warmup": warmup})
    if env.stub:
        return stub_directions(env, res, steps)
    u, v, _ = pair_of(w, 0)  # every rank holds the SAME pair and builds the full cost volume itself
    du, dv = ctx.upload_image(u), ctx.upload_image(v)

    def build(c, a, b, into=None):
        return c.costvolume_dev(a, b, w["dmin"], w["dmax"], "none", "census", float("inf"), w["win"], into=into)

    def stage_ms(c):
        t = {}
        for n, ms in c.timings():
            t.setdefault(n, []).append(ms)
        return {k: float(np.mean(x)) for k, x in t.items()}

    ref = {}
    # ---- single ----
    if "single" in transports:
        if rank == 0:
            try:
                cv, o, c = None, None, None
                for it in range(warmup + 1):
                    if it == warmup:
                        ctx.synchronize()
                        ctx.timing(True)
                        ctx.timing_reset()
                        t0 = time.perf_counter()
                    for _ in range(steps if it == warmup else 1):
                        cv = build(ctx, du, dv, cv)
                        _, o, c = ctx.aggregate_dev(cv, w["P1"], w["P2"], NDIR, w["MGM"], w["FH"], 1, None, "vfit", out=o, outcost=c)
                ctx.synchronize()
                dt = time.perf_counter() - t0
                a = stage_ms(ctx)
                ctx.timing(False)
                pn = "k_pass2" if "k_pass2" in a else "k_pass"
                agg = a[pn] + a["k_wta"]
                res["single"] = {"value": steps / dt, "ms_per_volume": dt / steps * 1e3, "k2_ms": a.get("k_cost"), "k3_ms": a[pn], "wta_ms": a["k_wta"],
                                 "roofline_frac": 12.0 * NDIR * float(nx) * ny * L / (agg * 1e-3) / 1e9 / HBM_PEAK_GBS}
                ref["o"], ref["c"] = o.download()[0], c.download()[0]
                for h in (cv, o, c):
                    h.free()
            except Exception as e:  # noqa: BLE001
                res["single"] = {"error": repr(e)[:300]}
            ctx.trim()
        env.ctrl_barrier()
    # ---- peer: one process, all GPUs, device-to-device copies ----
    if "peer" in transports and world > 1:
        if rank == 0:
            try:
                import mgm_amd
                os.environ["MGM_MULTI_TRANSPORT"] = "peer"
                m = mgm_amd.Multi(list(range(world)))
                try:
                    ims = [(m.ctx[k].upload_image(u), m.ctx[k].upload_image(v)) for k in range(world)]
                    cvs = [None] * world
                    o, c = m.ctx[0].new_image(nx, ny), m.ctx[0].new_image(nx, ny)
                    for it in range(warmup + 1):
                        if it == warmup:
                            for k in range(world):
                                m.ctx[k].synchronize()
                            m.ctx[0].timing(True)
                            m.ctx[0].timing_reset()
                            t0 = time.perf_counter()
                        for _ in range(steps if it == warmup else 1):
                            cvs = [build(m.ctx[k], ims[k][0], ims[k][1], cvs[k]) for k in range(world)]
                            m.aggregate_dev(cvs, w["P1"], w["P2"], NDIR, w["MGM"], w["FH"], 1, "vfit", o, c)  # returns when the result is on device 0
                    dt = time.perf_counter() - t0
                    a = stage_ms(m.ctx[0])
                    pn = "k_pass2" if "k_pass2" in a else "k_pass"
                    res["peer"] = {"value": steps / dt, "ms_per_volume": dt / steps * 1e3, "transport": m.transport(), "k2_ms": a.get("k_cost"),
                                   "k3_ms": a.get(pn), "wta_ms": a.get("k_wta"), "what": "ONE process drives all %d GPUs (mgm_multi_*), slabs by device-to-device copies" % world}
                    po, pc = o.download()[0], c.download()[0]
                    if ref:
                        res["peer"]["differs_from_single"] = nd(po, ref["o"]) + nd(pc, ref["c"])
                    else:
                        ref["o"], ref["c"] = po, pc
                    try:  # what the links of this node carry, device to device (DESIGN.md section 6 assumes 153 GB/s)
                        res["peer"]["link_probe"] = mdist.peer_copy_probe(world)
                    except Exception as e:  # noqa: BLE001
                        res["peer"]["link_probe"] = {"error": repr(e)[:200]}
                finally:
                    m.close()
            except Exception as e:  # noqa: BLE001
                res["peer"] = {"error": repr(e)[:300]}
        env.ctrl_barrier()
    # ---- rccl: one process per GPU ----
    if "rccl" in transports and world > 1:
        variants = [("rccl", False)] + ([("rccl_overlap", True)] if mdist.n_rounds(NDIR, world) > 1 else [])
        cv = None
        try:  # the links as the exchange will use them: every rank to rank+d at once (guarded by the leg's own time-out)
            lp = mdist.link_probe(dist, None, "cuda", 256, 3, min(60.0, timeout_s))
            if rank == 0:
                res["rccl_link_probe"] = lp
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                res["rccl_link_probe"] = {"error": repr(e)[:200]}
            if isinstance(e, mdist.ExchangeTimeout):
                res["rccl"] = {"error": "link probe timed out: " + repr(e)[:200], "fatal": True}
                variants = []
        for key, overlap in variants:
            stats, err, got = {}, None, None
            try:
                for it in range(warmup + 1):
                    if it == warmup:
                        env.sync_all()
                        stats.clear()
                        ctx.timing(True)
                        ctx.timing_reset()
                        t0 = time.perf_counter()
                    for _ in range(steps if it == warmup else 1):
                        cv = build(ctx, du, dv, cv)
                        got = mdist.aggregate_direction_sharded(ctx, cv, w["P1"], w["P2"], NDIR, w["MGM"], w["FH"], 1, "vfit", dist, None, None,
                                                                env.ctrl, timeout_s, overlap, stats)
                env.sync_all()
                dt = env.max_over_ranks(time.perf_counter() - t0)
                a = stage_ms(ctx)
                ctx.timing(False)
            except Exception as e:  # noqa: BLE001 -- ExchangeTimeout: the group is wedged, no further collective on it
                err = e
            if err is not None:
                res[key] = {"error": repr(err)[:300]}
                if isinstance(err, mdist.ExchangeTimeout) or not isinstance(err, mdist.ExchangeError):
                    res[key]["fatal"] = True
                    break
                continue
            n = max(1, stats.get("steps", 1))
            pn = "k_pass2" if "k_pass2" in a else "k_pass"
            res[key] = {"value": steps / dt, "ms_per_volume": dt / steps * 1e3, "rccl_ranks": world, "overlap": overlap, "k2_ms": a.get("k_cost"),
                        "k3_ms": a.get(pn), "passes_ms": stats.get("passes_ms", 0.0) / n, "exchange_ms": stats.get("exchange_ms", 0.0) / n,
                        "exchange_rounds": mdist.n_rounds(NDIR, world), "exchange_ms_per_round": stats.get("exchange_ms", 0.0) / n / max(1, mdist.n_rounds(NDIR, world)),
                        "wta_ms": stats.get("wta_ms", 0.0) / n, "gather_ms": stats.get("gather_ms", 0.0) / n,
                        "what": "one process per GPU; rank 0's stage times (torch events on the library's stream): its passes, the wait for the "
                                "slabs after them, its rows' ordered sum + WTA + vfit, the all-gather of the rows"}
            if rank == 0 and ref:
                res[key]["differs_from_single"] = nd(got[0].cpu().numpy(), ref["o"]) + nd(got[1].cpu().numpy(), ref["c"])
        if cv is not None:
            cv.free()
    du.free(), dv.free()
    if rank == 0:
        # DESIGN.md section 6's model next to what was measured: its single-GPU terms from THIS run's `single` leg where they
        # exist, its link rate from the probes (the assumed 153 GB/s otherwise)
        sg = res.get("single", {}) if isinstance(res.get("single"), dict) else {}
        lp = res.get("rccl_link_probe", {}) if isinstance(res.get("rccl_link_probe"), dict) else {}
        link = lp.get("min") or ((res.get("peer", {}).get("link_probe", {}) or {}).get("min") if isinstance(res.get("peer"), dict) else None) or 153.0
        k3_one = {1: 7.8, 2: 8.5, 4: 14.0, 8: sg.get("k3_ms") or 26.5}  # ms, measured on one GPU (profiles/r03_cfg4_pass_blocks.txt)
        res["model"] = {"what": "DESIGN.md section 6: K2 + K3(passes per rank) + exchange over one link + k_wta on 1/n of the rows",
                        "link_gbps_used": link, "link_rate_source": "measured (probe)" if (lp.get("min") or link != 153.0) else "assumed",
                        "prediction": mdist.sharding_model(world, NDIR, 4.0 * nx * ny * L / 1e9, k3_one, sg.get("wta_ms") or 17.3, sg.get("k2_ms") or 1.25, link)}
        # ... and for the node sizes the driver's scaling run uses, whatever this run's world is (single-GPU terms of THIS run)
        res["model"]["by_world"] = {str(n): {k: v for k, v in mdist.sharding_model(n, NDIR, 4.0 * nx * ny * L / 1e9, k3_one, sg.get("wta_ms") or 17.3,
                                                                                    sg.get("k2_ms") or 1.25, link).items()
                                             if k in ("passes_per_rank", "gb_per_link", "exchange_ms", "total_ms")} for n in (2, 4, 8)}
        for k in ("rccl", "rccl_overlap", "peer"):
            if isinstance(res.get(k), dict) and "ms_per_volume" in res[k] and res["model"]["prediction"].get("total_ms"):
                res[k]["vs_model"] = res[k]["ms_per_volume"] / res["model"]["prediction"]["total_ms"]
    return pick_transport(res, world)


def pick_transport(res, world):
    """What the direction-sharded leg reports: the best transport that WORKED -- rccl (plain or overlapped), else the
    one-process peer-copy path -- and, with several ranks and NEITHER, "replicas": the leg then has no figure of its own and
    says so (the line's headline -- independent pairs per rank, no data-path collective -- is the node's multi-GPU
    result).  The line stays valid whichever of them failed: `transport`, `ranks`, `rccl_link_probe`, `fallback_chain` and
    `differs_from_single` are always there."""
    order = ("rccl", "rccl_overlap", "peer", "single")
    best = [(res[k]["value"], k) for k in order if isinstance(res.get(k), dict) and "value" in res[k] and (world == 1 or k != "single")]
    chain = []
    for k in order[:3]:
        if world > 1 and isinstance(res.get(k), dict):
            chain.append("%s: %s" % (k, "ok" if "value" in res[k] else "failed (%s)" % str(res[k].get("error", "?"))[:120]))
    if best:
        res["value"], res["transport"] = max(best)
        if isinstance(res.get("single"), dict) and "value" in res["single"]:
            res["speedup_vs_single"] = res["value"] / res["single"]["value"]
    elif world > 1:
        res["value"], res["transport"] = None, "replicas"
        chain.append("replicas: no transport for the ordered slab exchange worked on this node -- the headline of this line "
                     "(independent pairs per rank, no data-path collective) is the multi-GPU result")
    res["fallback_chain"] = chain
    res.setdefault("rccl_link_probe", None)
    res["differs_from_single"] = {k: res[k]["differs_from_single"] for k in order if isinstance(res.get(k), dict) and "differs_from_single" in res[k]}
    return res


def pairs2_leg(env, name, steps, warmup):
    """ONE stereo pair per step split over TWO GPUs by run: the left->right mgm() on the even rank, the right->left one on
    the odd rank (no data-path exchange during the aggregation at all), then the odd rank's disparity map travels to the even
    one (W*H floats, one point-to-point transfer) for the left-right check -- what main() does with a pair (mgm.cc:376-423).
    With N ranks N/2 pairs run side by side.  The split DESIGN.md section 6's model prefers below 8 GPUs.  On one rank both
    runs share one launch (the N = 1 point)."""
    from mgm_amd import dist as mdist
    w = WORKLOADS[name]
    ctx, rank, world, torch, dist = env.ctx, env.rank, env.n_ranks, env.torch, env.dist
    nx, ny, L = w["nx"], w["ny"], labels_of(w)
    res = {"workload": "%s: %s" % (name, w["desc"]), "ranks": world, "unit": "disparity-volumes/s", "scaling": "strong",
           "steps": steps, "warmup": warmup, "what": "one pair per step and pair of ranks: left->right run on the even rank, right->left on the odd "
                                                      "one, disparity map sent over for the left-right check"}
    if env.stub:
        res["value"], res["stub"] = 1.0, True
        return res
    if world > 1 and world % 2:
        res["error"] = "needs an even number of ranks"
        return res
    u, v, _ = pair_of(w, rank // 2 if world > 1 else 0)
    lr = world == 1 or rank % 2 == 0
    du, dv = ctx.upload_image(u), ctx.upload_image(v)
    cvs, outs, outcs = [None, None], [ctx.new_image(nx, ny) for _ in range(2)], [ctx.new_image(nx, ny) for _ in range(2)]
    chk = [ctx.new_image(nx, ny) for _ in range(2)]
    other = ctx.new_image(nx, ny)  # (the partner rank's map)

    def one_step():
        if world == 1:
            cvs[0] = ctx.costvolume_dev(du, dv, w["dmin"], w["dmax"], "none", "census", float("inf"), w["win"], into=cvs[0])
            cvs[1] = ctx.costvolume_dev(dv, du, -w["dmax"], -w["dmin"], "none", "census", float("inf"), w["win"], into=cvs[1])
            ctx.aggregate_batch_dev(cvs, w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, None, "vfit", outs, outcs)
            ctx.leftright_dev(outs[0], outs[1], 1.0, out=chk[0])
            return
        a, b, lo, hi = (du, dv, w["dmin"], w["dmax"]) if lr else (dv, du, -w["dmax"], -w["dmin"])
        cvs[0] = ctx.costvolume_dev(a, b, lo, hi, "none", "census", float("inf"), w["win"], into=cvs[0])
        ctx.aggregate_dev(cvs[0], w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, None, "vfit", out=outs[0], outcost=outcs[0])
        ctx.synchronize()  # (the transfer below runs on torch's stream)
        mine = mdist.device_view(ctx.lib.mgm_img_device_ptr(outs[0].h), (ny, nx))
        if lr:
            dist.recv(mdist.device_view(ctx.lib.mgm_img_device_ptr(other.h), (ny, nx)), src=rank + 1)
            torch.cuda.synchronize()
            ctx.leftright_dev(outs[0], other, 1.0, out=chk[0])
        else:
            dist.send(mine, dst=rank - 1)
            torch.cuda.synchronize()

    for _ in range(max(1, warmup)):
        one_step()
    env.sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    env.sync_all()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    npairs = max(1, world // 2)
    res.update({"value": 2.0 * npairs * steps / dt, "pairs_per_s": npairs * steps / dt, "ms_per_pair": dt / steps * 1e3})
    for h in [du, dv, other] + [x for x in cvs if x is not None] + outs + outcs + chk:
        h.free()
    return res


def stub_directions(env, res, steps):
    """Test driver only (TEST_CONTEXT_FACTORY): the control flow of the direction-sharded leg on CPU ranks -- agreement and the ordered slab
    exchange over gloo on small host tensors (mgm_amd/dist.py, the code the RCCL leg runs), no device work."""
    from mgm_amd import dist as mdist
    torch, dist = env.torch, env.dist
    world, rank = env.n_ranks, env.rank
    NDIR, ny, nx, L = 8, 13, 21, 10
    if world == 1:
        res["single"] = {"value": 1.0, "stub": True}
        res["value"], res["transport"] = 1.0, "single"
        return res
    fail_at = os.environ.get("MGM_STUB_FAIL_AT", "")  # test hook: "rccl" / "rccl,peer" -- those transports "fail"
    if "peer" in fail_at:
        res["peer"] = {"error": "stub: peer transport made to fail"}
    if "rccl" in fail_at:
        res["rccl_link_probe"] = {"error": "stub: link probe made to fail"}
        res["rccl"] = {"error": "stub: rccl transport made to fail", "fatal": True}
        if "peer" not in fail_at:
            res["peer"] = {"value": 1.0, "stub": True, "differs_from_single": 0, "transport": "peer"}
        return pick_transport(res, world)
    t0 = time.perf_counter()
    bad = 0
    for s in range(steps):
        first, count = mdist.passes_of_rank(NDIR, world, rank)
        gen = torch.Generator().manual_seed(1234 + s)
        vols = torch.rand((NDIR, ny, nx, L), generator=gen)  # (every rank can regenerate every pass: the check below)
        if not mdist.agree(True, dist, None, env.ctrl, "cpu"):
            raise RuntimeError("stub agreement failed")
        if os.environ.get("MGM_STUB_DIE_AT") == "exchange" and rank == world - 1 and s == 0:
            os.kill(os.getpid(), signal.SIGKILL)  # test hook: a rank dies between the agreement and its first transfer
        recv = mdist.exchange_lr([vols[p] for p in range(first, first + count)], NDIR, ny, dist, like=vols[0, :0], timeout_s=60.0)
        r0, nr = mdist.row_slabs(ny, world)[rank]
        bad += int((recv != vols[:, r0:r0 + nr]).sum())
    dt = env.max_over_ranks(time.perf_counter() - t0)
    res["rccl"] = {"value": steps / dt, "ms_per_volume": dt / steps * 1e3, "rccl_ranks": world, "overlap": False, "stub": True,
                   "differs_from_single": bad, "exchange_rounds": mdist.n_rounds(NDIR, world)}
    res["rccl_link_probe"] = mdist.link_probe(dist, None, "cpu", 1, 1, 30.0)  # (the probe's own code, over gloo, 1 MB)
    if rank == 0:
        res["model"] = {"prediction": mdist.sharding_model(world, NDIR, 4.0 * nx * ny * L / 1e9, {1: 1.0, 2: 1.0, 4: 1.0, 8: 1.0}, 1.0, 0.1,
                                                           res["rccl_link_probe"]["min"] or 1.0)}
    return pick_transport(res, world)


# ---- main ---------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity gate (profiling runs)")
    ap.add_argument("--repeats", type=int, default=None,
                    help="further timed blocks of K steps after the reported one (their rates go to `repeat_values`); "
                         "default: as many as make the GPU phase last about 10 s, at most 8")
    ap.add_argument("--batch", type=int, default=None, choices=list(range(1, 17)),
                    help="pairs per step and GPU: their volumes share ONE launch of the pass kernel.  Default: "
                         "12 (204 GB of Lr volumes at 1920x1080x256 x 8 directions), 16 for workloads of at most 128 labels "
                         "(two / four of those volumes share every wave), 1 for cfg4")
    ap.add_argument("--pipeline", type=int, default=1, choices=list(range(1, 17)),
                    help="D >= 2: the context gathers the aggregation calls of D consecutive steps into one batched launch "
                         "(mgm_ctx_set_pipeline) -- a caller with a stream of single pairs or small batches")
    ap.add_argument("--mode", default="pairs", choices=["pairs", "directions", "pairs2"],
                    help="'pairs' = independent pairs (weak scaling, the headline); 'directions' = only the direction-sharded "
                         "leg: ONE volume per step, its passes sharded over the GPUs with the ordered slab exchange (strong); "
                         "'pairs2' = one stereo pair per step and pair of GPUs, its two mgm() runs on one GPU each (strong)")
    ap.add_argument("--extras", default="auto", choices=["auto", "on", "off"],
                    help="the cfg5-replicas and cfg4-directions legs after the headline; auto = on for the plain command line")
    ap.add_argument("--extras-timeout", type=float, default=420.0, help="seconds the guarded extras may take altogether (round 6: the gated variants run CPU legs too)")
    ap.add_argument("--exchange-timeout", type=float, default=90.0, help="seconds one direction-sharded step may take")
    args = ap.parse_args()
    plain = args.workload is None and args.batch is None and args.mode == "pairs" and args.pipeline == 1
    extras = args.extras == "on" or (args.extras == "auto" and plain)
    wname = args.workload or ("cfg4" if args.mode in ("directions", "pairs2") else "cfg3")
    w = WORKLOADS[wname]
    stub = TEST_CONTEXT_FACTORY is not None  # (tests/run_bench_stub.py: gloo ranks on CPU, a context that computes nothing)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args)
    env = Env(args, stub)
    env.setup()
    rank, n_ranks, ctx = env.rank, env.n_ranks, env.ctx
    line = OneLine(rank)
    nx, ny, L = w["nx"], w["ny"], labels_of(w)
    cells = float(nx) * ny * L

    if args.mode == "pairs2":  # one pair split by run over two GPUs, as the headline
        line.guard(args.extras_timeout)
        d = pairs2_leg(env, wname, args.steps, args.warmup)
        if rank == 0:
            line.res = {"metric": "disparity-volumes/sec (W*H*L cost volume -> 8-dir MGM -> WTA+vfit)", "value": d.get("value"),
                        "unit": "disparity-volumes/s", "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
                        "ms_per_step": d.get("ms_per_pair"), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                        "data": "synthetic" if not stub else "stub (no device work)",
                        "config": {"workload": "%s: %s" % (wname, w["desc"]), "W": nx, "H": ny, "L": L, "pairs_per_step": max(1, n_ranks // 2),
                                   "parallelism": "one pair per two GPUs: its left->right and right->left runs on one GPU each"},
                        "pairs2": d}
        line.emit(env.dist is not None)
        ctx.close()
        return 0
    if args.mode == "directions":  # the direction-sharded leg alone, as the headline
        line.guard(args.extras_timeout)
        d = directions_legs(env, wname, args.steps, args.warmup, args.exchange_timeout, ("single", "rccl") if n_ranks > 1 else ("single",))
        if rank == 0:
            key = d.get("transport", "single")
            sub = d.get(key, {})
            line.res = {"metric": "disparity-volumes/sec (W*H*L cost volume -> 8-dir MGM -> WTA+vfit)", "value": d.get("value"),
                        "unit": "disparity-volumes/s", "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
                        "ms_per_step": sub.get("ms_per_volume"), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                        "data": "synthetic" if not stub else "stub (no device work)",
                        "config": {"workload": "%s: %s" % (wname, w["desc"]), "W": nx, "H": ny, "L": L, "pairs_per_step": 1,
                                   "parallelism": "one volume, %d-way direction sharding, ordered slab exchange (%s)" % (n_ranks, key)},
                        "directions": d}
        line.emit(env.dist is not None)
        ctx.close()
        return 0

    if args.batch is None:
        args.batch = 1 if wname == "cfg4" else (16 if L <= 128 else 12)
    B = args.batch
    m = pairs_leg(env, w, B, args.steps, args.warmup, args.repeats, keep=True, pipeline=args.pipeline)
    dt = m["dt"]
    try:  # device memory the headline launch leaves in use: the context's grow-only workspace (NDIR fp32 Lr volumes per batched volume) + the kept maps
        free_b, total_b = ctx.mem_info()
        mem_in_use_gb = (total_b - free_b) / 1e9
    except Exception:  # noqa: BLE001
        mem_in_use_gb = None
    if rank == 0:
        from mgm_amd import shard
        vols_per_block = args.steps * n_ranks * B
        line.res = res = {
            "metric": "disparity-volumes/sec (W*H*L cost volume -> 8-dir MGM -> WTA+vfit)",
            "value": shard.job_rate([args.steps * B] * n_ranks, dt),  # whole-job aggregate: every rank did K batches of B volumes
            "unit": "disparity-volumes/s", "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not stub else "stub (no device work)",
            "config": {"workload": "%s: %s" % (wname, w["desc"]), "W": nx, "H": ny, "L": L, "pairs_per_step": B,
                       "NDIR": w["NDIR"], "TSGM": w["MGM"], "potential": "FH" if w["FH"] else "Hirschmueller",
                       "P1": w["P1"], "P2": w["P2"], "cost": cost_of(w), "channels": w.get("nch", 1), "census_win": w["win"], "refine": "vfit",
                       "edge_weights": ("-aP2 %g -aThresh %g" % (w["aP2"], w["aThresh"])) if weighted(w) else None,
                       "pipeline_depth": args.pipeline,
                       "parallelism": ("independent pairs, %d per step and GPU, no data-path collective" % B) if n_ranks > 1 else "1 GPU"},
            "roofline": {**roofline_of(w, B, m["avg"], wname, step_ms=(dt / args.steps * 1e3) if args.pipeline > 1 else None, per_step=m["per_step"]),
                         "workspace_gb": {"lr_volumes": B * max(1, args.pipeline) * w["NDIR"] * (64.0 * nx * ny if w.get("ragged") else cells) * 4.0 / 1e9,
                                          "device_memory_in_use_after_the_timed_steps": mem_in_use_gb,
                                          "note": "NDIR fp32 Lr volumes per batched volume, all passes concurrent (DESIGN.md 3); "
                                                  "mgm_ctx_set_workspace_limit cuts a batch into sub-batches"}},
            # (say it HERE, not only in DESIGN.md) what scales on a node and what does not
            "multi_gpu_note": ("`value` is the weak-scaling mode: every rank aggregates its own pairs, no data-path collective (replicas). "
                               "Sharding ONE volume by direction (`directions`, cfg4) is link-bound by construction -- fp32 Lr row slabs, summed in pass "
                               "order, not all-reduced -- and DESIGN.md section 6's model puts it at 0.24x / 0.81x / 2.1x of one GPU on 2 / 4 / 8 GPUs at "
                               "153 GB/s per link; the strong-scaling mode to prefer below 8 GPUs is `cfg4_pairs2` (one mgm() run of the pair per GPU)."
                               + ("" if n_ranks > 1 else "  (This line is the N = 1 leg: the very code path every rank of an N-rank run takes, minus the "
                                                         "barrier's peers; `directions.model` holds the model's prediction for 2 / 4 / 8 GPUs.)")),
            "kernel_ms_per_step": m["per_step"],
            "repeat_values": [vols_per_block / t for t in m["rep_dt"]],
            "parity": None,
            "mcell_updates_per_s": vols_per_block * cells * w["NDIR"] / dt / 1e6}

    # ---- parity gate: pair 0 of the last timed step against the CPU oracle; CPU baseline --------------------------------
    parity, whole_s, T = None, None, 1
    gate = (args.steps + args.warmup) % B  # which pair of the batch is gated: not always the first
    if rank == 0 and not stub:
        from oracle.oracle import usable_cpus
        T = min(32, usable_cpus())
        if not args.no_parity:
            if cells > PARITY_MAX_CELLS:
                parity = {"status": "skipped", "why": "%.1f G cells: beyond the in-run oracle (covered by tests/test_gpu_fullsize.py)" % (cells / 1e9)}
            else:
                got_o, got_c = m["outs"][gate].download()[0], m["outcs"][gate].download()[0]
                ref_o, ref_c, whole_s = oracle_whole_volume(w, T, gate)
                bad = nd(ref_o, got_o.reshape(ref_o.shape)) + nd(ref_c, got_c.reshape(ref_c.shape))
                parity = {"status": "bit-exact" if bad == 0 else "FAILED", "differing_words": bad, "pair": gate,
                          "what": "refined disparity and cost maps (2 x %dx%d float32) of pair %d of %d (drawn from the step count) from the last "
                                  "timed step vs oracle/mgm_oracle.c on the same pair (%d threads, %.1f s)" % (nx, ny, gate, B, T, whole_s)}
                if bad:
                    line.code = 3
            res["parity"] = parity
        if n_ranks == 1 and not args.no_cpu_baseline and cells <= PARITY_MAX_CELLS:
            res["cpu_baseline"], ref_out = cpu_baseline(w, whole_s, T, pair=gate)
            if ref_out and parity is not None and parity.get("status") != "skipped":  # the reference's own maps, while we have them
                got_o, got_c = m["outs"][gate].download()[0], m["outcs"][gate].download()[0]
                parity["vs_reference_differing_words"] = nd(ref_out["o"], got_o.reshape(ref_out["o"].shape)) + nd(ref_out["c"], got_c.reshape(ref_out["c"].shape))
                if parity["vs_reference_differing_words"]:
                    parity["status"], line.code = "FAILED", 3
    for h in m["outs"] + m["outcs"]:
        h.free()

    # ---- the extra legs (guarded) ------------------------------------------------------------------------------------
    if extras:
        line.guard(args.extras_timeout)
        ctx.trim()
        env.ctrl_barrier()  # (rank 0 may have spent a minute in the CPU legs)
        # what north_star names beside the headline combination, in the driver's own line (round 4): per-edge weights, AD on a
        # colour pair, clipped NCC, label counts off the kernels' widths (200, 151), 768 labels, and a stream of single pairs
        # through a pipelined context -- short legs (4 pairs
        # per step where the workspace allows, 5 steps), same measurement as the headline, no CPU legs
        try:
            vres = {}
            if rank == 0:
                res["variants"] = vres
            # round 5: the ONE-PAIR path first -- plain launches of one and two volumes (what `mgm u v out` and INTEGRATION.md's
            # binding run: mgm.cc:376-385 + 405-414 are two volumes per pair), the figures VERDICT r4 asked to see driver-timed
            for vname, vb, vd in (("cfg3", 1, 1), ("cfg3", 2, 1), ("cfg2", 1, 1), ("cfg2", 2, 1), ("cfg1s", 2, 1),
                                  ("cfg3w", 4, 1), ("cfg3hw", 4, 1), ("cfg3ad", 4, 1), ("cfg3ncc", 4, 1), ("cfg3L200", 4, 1), ("cfg1s", 4, 1),
                                  ("cfg3L768", 1, 1), ("cfg3", 1, 4), ("cfg2", 1, 8),
                                  # round 5: what had no number before -- range images (-m/-M) and TSGM_ITER, and the
                                  # fall-back kernels (free-form weights, 1536 labels, negative penalties, NaN costs)
                                  ("cfg3r", 1, 1), ("cfg3r", 2, 1), ("cfg3r", 4, 1), ("cfg3hr", 1, 1), ("cfg3hr", 4, 1),
                                  ("cfg3r50", 1, 1), ("cfg1sr", 2, 1),
                                  # ... and the same ragged volumes on the dense HULL (MGM_HIP_REL=0: no range-proportional copy), the run
                                  # the range-proportional kernels are to be compared with (vd = -1 marks them)
                                  ("cfg3r", 1, -1), ("cfg3r", 4, -1), ("cfg3hr", 4, -1),
                                  ("cfg3i2", 1, 1), ("cfg3w3", 1, 1), ("cfg3hw3", 1, 1), ("cfg3L1536", 1, 1),
                                  ("cfg3neg", 1, 1), ("cfg3nan", 1, 1)):
                vw = WORKLOADS[vname]
                hull = vd < 0
                vd = abs(vd)
                vsteps = 8 if vd > 1 else (2 if vname in ("cfg3nan", "cfg3L1536") else 5)  # (the slow fall-backs: 0.25-0.4 s per step)
                # round 6: the ragged / TSGM_ITER / free-form-weight legs are gated on the CPU oracle too --
                # one pair of the leg's last step against oracle/mgm_oracle.c (orc_mgm_ranged for range images) -- and the ragged
                # ones carry the REFERENCE on the same range images as their CPU baseline (one run, timed inside the harness)
                gated = (not stub and not hull and vd == 1 and not args.no_parity and n_ranks == 1
                         and (vw.get("ragged") or vname in ("cfg3i2", "cfg3w3")))
                if hull:
                    os.environ["MGM_HIP_REL"] = "0"
                try:
                    vm = pairs_leg(env, vw, vb, vsteps, 1, 0, keep=gated, pipeline=vd)
                finally:
                    os.environ.pop("MGM_HIP_REL", None)
                vpar, vcpu = None, None
                if gated and rank == 0:
                    try:
                        vg = (vsteps + 1) % vb
                        got_o, got_c = vm["outs"][vg].download()[0], vm["outcs"][vg].download()[0]
                        ref_o, ref_c, osec = oracle_whole_volume(vw, T, vg)
                        bad = nd(ref_o, got_o.reshape(ref_o.shape)) + nd(ref_c, got_c.reshape(ref_c.shape))
                        vpar = {"status": "bit-exact" if bad == 0 else "FAILED", "differing_words": bad, "pair": vg,
                                "what": "refined disparity and cost maps of pair %d of %d vs oracle/mgm_oracle.c (%d threads, %.1f s)" % (vg, vb, T, osec)}
                        if bad:
                            line.code = 3
                        if vw.get("ragged") and not args.no_cpu_baseline and vb == 1:
                            rout = {}
                            rsec = reference_whole_volume(vw, T, rout, vg)
                            if rsec is not None:
                                vcpu = {"value": 1.0 / rsec, "unit": "disparity-volumes/s", "cores": T, "kind": "reference",
                                        "sample": "one whole %dx%d volume on the same range images: the reference's allocate_and_fill_sgm_costvolume + "
                                                  "mgm() + subpixel_refinement_sgm (oracle/_ref/libmgm_ref.so) on %d OpenMP threads, one run, timed around "
                                                  "the three calls" % (vw["nx"], vw["ny"], T)}
                                fin = np.isfinite(rout["c"])
                                vpar["vs_reference_differing_words"] = nd(np.where(fin, rout["o"], 0), np.where(fin, got_o.reshape(rout["o"].shape), 0)) + \
                                    nd(rout["c"], got_c.reshape(rout["c"].shape))
                                if vpar["vs_reference_differing_words"]:
                                    vpar["status"], line.code = "FAILED", 3
                    except Exception as e:  # noqa: BLE001
                        vpar = {"status": "error", "why": repr(e)[:300]}
                if gated:
                    for h in vm.get("outs", []) + vm.get("outcs", []):
                        h.free()
                if rank == 0:
                    vr = roofline_of(vw, vb, vm["avg"], vname, step_ms=(vm["dt"] / vsteps * 1e3) if vd > 1 else None, per_step=vm["per_step"])
                    pn = next((k for k in ("k_pass2", "k_pass", "k_pass_rel", "k_pass_exact") if k in vm["avg"]), "k_pass2")
                    vres["%s x%d%s%s" % (vname, vb, (" pipeline %d" % vd) if vd > 1 else "", " on the dense hull" if hull else "")] = {
                        "workload": vw["desc"], "value": shard.job_rate([vsteps * vb] * n_ranks, vm["dt"]), "unit": "disparity-volumes/s",
                        "roofline_frac": vr["frac"], "roofline_frac_counter": vr["frac_counter"], "frac_of_achievable": vr["frac_of_achievable"],
                        "saturated": vr["saturated"],
                        **({"frac_range_proportional": vr["frac_range_proportional"]} if "frac_range_proportional" in vr else {}),
                        "pass_kernel": pn,
                        "time_basis": vr["time_basis"], "k2_ms": vm["avg"].get("k_cost"), "k3_ms": vm["avg"].get(pn),
                        "wta_ms": vm["avg"].get("k_wta"),
                        **({"parity": vpar} if vpar is not None else {}), **({"cpu_baseline": vcpu} if vcpu is not None else {})}
                ctx.trim()
            if rank == 0:  # range-proportional over dense hull, same volumes, same box, same run
                for k in ("cfg3r x1", "cfg3r x4", "cfg3hr x4"):
                    if k in vres and k + " on the dense hull" in vres:
                        vres[k]["over_dense_hull"] = vres[k]["value"] / vres[k + " on the dense hull"]["value"]
                roofline_digest(res, vres, wname)
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                res.setdefault("variants", {})["error"] = repr(e)[:300]
        ctx.trim()
        try:
            r5 = replicas_leg(env, "cfg5", max(5, args.steps), min(args.warmup, 2))
            if rank == 0:
                res["cfg5_replicas"] = r5
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                res["cfg5_replicas"] = {"error": repr(e)[:300]}
        ctx.trim()
        d = {}
        if rank == 0:
            res["directions"] = d  # (filled as the sub-legs finish: a later one that hangs leaves the earlier ones in the line)
        try:
            directions_legs(env, "cfg4", max(2, min(args.steps, 5)), 1, args.exchange_timeout, res=d)
        except Exception as e:  # noqa: BLE001
            d["error"] = repr(e)[:300]
        if n_ranks >= 2 and n_ranks % 2 == 0 and not d.get("rccl", {}).get("fatal") and not d.get("rccl_overlap", {}).get("fatal"):
            ctx.trim()
            try:  # (the split the model prefers below 8 GPUs: one run of the pair per GPU)
                p2 = pairs2_leg(env, "cfg4", max(2, min(args.steps, 5)), 1)
            except Exception as e:  # noqa: BLE001
                p2 = {"error": repr(e)[:300]}
            if rank == 0:
                res["cfg4_pairs2"] = p2
    failed = line.code != 0
    if rank == 0 and failed:
        print("bench.py: PARITY GATE FAILED: %s" % json.dumps(res.get("parity")), file=sys.stderr, flush=True)
    if env.dist is not None:
        line.emit(True)  # (os._exit: librccl leaves a version banner in the C stdio buffer that would be flushed after the json)
    line.emit(False)
    ctx.close()
    return line.code


if __name__ == "__main__":
    sys.exit(main())
