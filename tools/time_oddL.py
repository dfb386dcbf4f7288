"""Development aid: one aggregation of an odd label count (the reference's own example: 700x500, -r -120 -R 30 = 151
labels, -O 4, TSGM 2), padded second build vs first build (MGM_HIP_PAD=0)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import mgm_amd
from mgm_amd import synth
nx, ny, L = 700, 500, 151
for integer in (True, False):
    C = synth.raw_volume(nx, ny, L, seed=1, maxcost=200)
    if not integer:
        C = C * np.float32(3.1)
    for NDIR, MGM, FH, P1, P2 in ((4, 2, 0, 8.0, 32.0), (8, 3, 1, 2.0, 20000.0)):
        ctx = mgm_amd.Context(0)
        cv = ctx.upload_volume(C, -120)
        ctx.timing(True)
        for rep in range(3):
            ctx.timing_reset()
            ctx.aggregate_dev(cv, P1, P2, NDIR, MGM, FH, 1, None, "vfit")
            ctx.synchronize()
        print("PAD=%s integer=%s NDIR=%d MGM=%d FH=%d:" % (os.environ.get("MGM_HIP_PAD", "1"), integer, NDIR, MGM, FH),
              [(k, round(v, 3)) for k, v in ctx.timings()], flush=True)
        ctx.close()
