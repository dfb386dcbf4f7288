"""Development aid (GPU box): wall time of the operand-order-faithful pass kernel (mgm_pass_exact.hip) on a full-size volume
with NaN costs, next to the fast kernels on the same volume without them."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mgm_amd
from mgm_amd import synth
nx, ny, L = 1920, 1080, 256
C = synth.raw_volume(nx, ny, L)
ctx = mgm_amd.Context(0)
for tag, vol in (("clean", C), ("nan", None)):
    if vol is None:
        C[100, 200, 7] = np.nan
        vol = C
    cv = ctx.upload_volume(vol, 0)
    for mode in ((8, 3, 0, 8.0, 32.0), (8, 3, 1, 2.0, 20000.0)):
        ctx.timing(True); ctx.timing_reset()
        t0 = time.perf_counter()
        _, o, c = ctx.aggregate_dev(cv, mode[3], mode[4], mode[0], mode[1], mode[2], 1, None, "vfit")
        ctx.synchronize()
        dt = time.perf_counter() - t0
        ks = {}
        for n, ms in ctx.timings():
            ks[n] = ks.get(n, 0) + ms
        ctx.timing(False)
        print(tag, "FH" if mode[2] else "Hirschmueller", "wall %.3f s" % dt, {k: round(v, 1) for k, v in ks.items()}, flush=True)
        o.free(); c.free()
    cv.free()
