"""Development aid (GPU box): one LARGE ragged volume through the range-proportional kernels and through the hull kernels
(64-bit offsets, hand-off region of several GB): python tools/rel_bigcase.py [nx ny]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import mgm_amd  # noqa: E402
from mgm_amd import synth  # noqa: E402
from test_gpu_rel import ndiff, ranges  # noqa: E402

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 3072)
dmin, dmax, half = -255, 0, 24
u, v, gt = synth.stereo_pair(nx, ny, -190, 0, seed=5)
lo, hi = ranges(gt, dmin, dmax, half, 7)
res = {}
for mode in ("2", "0"):
    os.environ["MGM_HIP_REL"] = mode
    with mgm_amd.Context(0) as ctx:
        cv = ctx.costvolume(u, v, lo, hi, "none", "census", float("inf"), 5)
        ctx.timing(True)
        t0 = time.time()
        _, o, c = ctx.aggregate_dev(cv, 2.0, 20000.0, 8, 3, 1, 1, None, "vfit")
        a, b = o.download(), c.download()
        res[mode] = (a, b, [(n, round(ms, 2)) for n, ms in ctx.timings()], time.time() - t0)
for mode in res:
    print("MGM_HIP_REL=%s" % mode, res[mode][2], "%.2f s" % res[mode][3])
print("%dx%d: differing labels %d, differing costs %d" % (nx, ny, ndiff(res["2"][0], res["0"][0]), ndiff(res["2"][1], res["0"][1])))
