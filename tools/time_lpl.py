"""Development aid: K3 time per volume of the cfg2 mode (-O 4, TSGM 2, Hirschmueller) at 64 / 128 / 256 labels, 4 volumes per launch:
how much of a step is fixed cost."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import mgm_amd
from mgm_amd import synth
nx, ny = 1920, 1080
for L in (64, 128, 256):
    ctx = mgm_amd.Context(0)
    cvs = [ctx.upload_volume(synth.raw_volume(nx, ny, L, seed=b), 0) for b in range(4)]
    ctx.timing(True)
    for rep in range(3):
        ctx.timing_reset()
        ctx.aggregate_batch_dev(cvs, 8.0, 32.0, 4, 2, 0, 1, None, "vfit")
        ctx.synchronize()
    t = dict()
    for k, v in ctx.timings():
        t[k] = t.get(k, 0) + v
    print("L=%d: per volume" % L, {k: round(v / 4, 3) for k, v in t.items()}, flush=True)
    ctx.close()
