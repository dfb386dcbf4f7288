import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import mgm_amd
from mgm_amd import synth
from oracle.oracle import Oracle
from helpers import ndiff
orc = Oracle(threads=4)
ctx = mgm_amd.Context(0)
def case(nx, ny, L, NDIR, MGM, FH, P1, P2, seed, inf_frac, integer=True):
    C = synth.raw_volume(nx, ny, L, seed=seed, inf_frac=inf_frac)
    if not integer: C = (C * np.float32(1.0 / 3.0)).astype(np.float32)
    So, oo, co, lra = orc.mgm(C, 0, P1, P2, NDIR, MGM, FH, 1, None, dump_lr=True)
    cv = ctx.upload_volume(C, 0)
    S, o, c = ctx.aggregate(cv, P1, P2, NDIR, MGM, FH, 1, None, None, want_S=True)
    print((nx, ny, L, NDIR, MGM, FH, P1, P2), "S diff", ndiff(S.download(), So), "cost diff", ndiff(c, co))
    for p in range(NDIR):
        lr = ctx.debug_lr(cv, p)
        d = (lr.view(np.uint32) != lra[p].view(np.uint32)) & ~(np.isnan(lr) & np.isnan(lra[p]))
        if d.any():
            ys, xs, os_ = np.nonzero(d)
            print("  pass", p, "differs in", int(d.sum()), "words; first y,x,o =", ys[0], xs[0], os_[0], "rows", sorted(set(ys.tolist()))[:12], "cols", sorted(set(xs.tolist()))[:12],
                  "got", lr[ys[0], xs[0], os_[0]], "want", lra[p][ys[0], xs[0], os_[0]])
    S.free(); cv.free()
case(4, 24, 383, 7, 2, 0, 1.5, 20000.0, 68, 0.05)
case(4, 24, 384, 7, 2, 0, 1.5, 20000.0, 68, 0.0)
case(4, 24, 256, 7, 2, 0, 1.5, 20000.0, 68, 0.0)
case(40, 24, 256, 8, 3, 0, 8.0, 32.0, 68, 0.0)
case(40, 24, 256, 8, 3, 1, 2.0, 20000.0, 68, 0.05)
