import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import mgm_amd
from mgm_amd import synth
def run(nb, nx, ny, L, NDIR, MGM, FH, P1, P2, third):
    ctx = mgm_amd.Context(0)
    cvs = []
    for b in range(nb):
        C = synth.raw_volume(nx, ny, L, seed=18 * 17 + b, inf_frac=0.05)
        if third and b % 2 == 0: C = (C * np.float32(1.0 / 3.0)).astype(np.float32)
        cvs.append(ctx.upload_volume(C, 0))
    try:
        ctx.aggregate_batch_dev(cvs, P1, P2, NDIR, MGM, FH, 1, None, None)
        ctx.synchronize()
        print((nb, nx, ny, L, NDIR, MGM, FH, P1, P2, third), "ok", flush=True)
    except mgm_amd.MgmError as e:
        print((nb, nx, ny, L, NDIR, MGM, FH, P1, P2, third), "FAILED", e, flush=True)
    ctx.close()
inf = float("inf")
run(5, 29, 5, 5, 4, 3, 0, 2.0, inf, True)
run(1, 29, 5, 5, 4, 3, 0, 2.0, inf, True)
run(5, 29, 5, 5, 4, 3, 0, 2.0, inf, False)
run(5, 29, 5, 5, 4, 3, 0, 2.0, 32.0, True)
run(5, 29, 5, 64, 4, 3, 0, 2.0, inf, True)
run(2, 29, 5, 5, 4, 3, 0, 2.0, inf, True)
run(1, 29, 5, 5, 3, 3, 0, 2.0, inf, True)
run(1, 29, 5, 5, 4, 2, 0, 2.0, inf, True)
run(1, 60, 5, 5, 4, 3, 0, 2.0, inf, True)
