"""Development aid: run one full-size aggregation with MGM_HIP_DEBUG_STATS=1."""
import os, sys
os.environ.setdefault("MGM_HIP_DEBUG_STATS", "1")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
# the timers / switches live in the development build of the pass kernels: tools/sweep_build.sh dev -DMGM_P2_DEV=1
DEV = os.path.join(ROOT, "mgm_amd", "lib", "variants", "dev", "libmgm_hip.so")
if os.path.exists(DEV):
    os.environ.setdefault("MGM_HIP_LIB", DEV)
import numpy as np
import mgm_amd
from mgm_amd import synth
cfgs = {"cfg2": (1920, 1080, 128, 4, 2, 0, 8.0, 32.0), "cfg3h": (1920, 1080, 256, 8, 3, 0, 8.0, 32.0),
        "cfg3": (1920, 1080, 256, 8, 3, 1, 2.0, 20000.0)}
for name in sys.argv[1:] or ["cfg3h"]:
    nx, ny, L, NDIR, MGM, FH, P1, P2 = cfgs[name]
    ctx = mgm_amd.Context(0)
    if os.environ.get("MGM_STATS_REAL", "0") == "1":  # a census cost volume of a synthetic pair instead of random costs
        u, v, _ = synth.stereo_pair(nx, ny, -(L - 1) * 3 // 4, 0)
        du, dv = ctx.upload_image(u), ctx.upload_image(v)
        cv = ctx.costvolume_dev(du, dv, -(L - 1), 0, "none", "census", float("inf"), 5)
    else:
        cv = ctx.upload_volume(synth.raw_volume(nx, ny, L), 0)
    ctx.timing(True)
    for rep in range(2):
        ctx.timing_reset()
        _, o, c = ctx.aggregate_dev(cv, P1, P2, NDIR, MGM, FH, 1, None, "vfit")
        ctx.synchronize()
        print(name, ctx.timings(), flush=True)
    ctx.close()
