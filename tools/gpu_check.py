"""Diagnostic parity sweep: HIP path vs the CPU oracle, stage by stage.

Run on a GPU box:  python tools/gpu_check.py [--quick]
Prints one line per case with the number of differing float32 words in the cost
volume, each pass's Lr, the corrected S, the labels and the costs.  This is a
development aid; the judged parity tests are tests/test_gpu_*.py.
"""
import argparse
import faulthandler
import itertools
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mgm_amd  # noqa: E402
from mgm_amd import synth  # noqa: E402
from oracle.oracle import Oracle, bits_equal  # noqa: E402


def ndiff(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    na, nb = np.isnan(a), np.isnan(b)
    return int(np.sum((a.view(np.uint32) != b.view(np.uint32)) & ~(na & nb)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--case-timeout", type=int, default=60)
    ap.add_argument("--first", default="")
    args = ap.parse_args()
    orc = Oracle()
    ctx = mgm_amd.Context(0)
    print(mgm_amd.load_library().mgm_version().decode(), flush=True)
    bad = 0
    nb = ctx.selftest_div3()
    print('SELFTEST div3: mismatching inputs =', nb, flush=True)
    bad += nb != 0

    # ---- cost volume ----
    for nch, (nx, ny), (dmin, dmax), (pre, dist), win, td in [
        (1, (40, 23), (-7, 8), ("none", "ad"), 3, np.inf),
        (3, (40, 23), (-7, 8), ("none", "ad"), 3, np.inf),
        (1, (40, 23), (-7, 8), ("none", "census"), 3, np.inf),
        (1, (40, 23), (-7, 8), ("none", "census"), 5, np.inf),
        (3, (40, 23), (-30, 40), ("none", "census"), 3, np.inf),
        (3, (40, 23), (-30, 40), ("none", "census"), 5, np.inf),
        (1, (70, 23), (30, 100), ("none", "census"), 7, 20.0),
        (1, (40, 23), (-7, 8), ("census", "ad"), 3, np.inf),
        (3, (33, 17), (-70, 70), ("none", "sd"), 3, 50.0),
    ]:
        u, v, _ = synth.stereo_pair(nx, ny, max(dmin, -nx // 4), min(dmax, nx // 4), nch=nch)
        a = orc.costvolume(u, v, dmin, dmax, pre, dist, td, win)
        du, dv = ctx.upload_image(u), ctx.upload_image(v)
        cv = ctx.costvolume_dev(du, dv, dmin, dmax, pre, dist, td, win)
        b = cv.download()
        d = ndiff(a, b)
        bad += d != 0
        print("COST nch=%d %dx%d [%d,%d] %s/%s win=%d td=%s : diff=%d" % (nch, nx, ny, dmin, dmax, pre, dist, win, td, d),
              flush=True)
        cv.free(); du.free(); dv.free()

    # ---- aggregation ----
    shapes = [(40, 37, 12), (70, 35, 64), (35, 70, 100), (50, 40, 128)]
    if not args.quick:
        shapes += [(130, 70, 151), (64, 48, 256), (48, 40, 300), (40, 36, 512)]
    for (nx, ny, L) in shapes:
        C = synth.raw_volume(nx, ny, L, inf_frac=0.03)
        dmin = -5
        cv = ctx.upload_volume(C, dmin)
        combos = [(8, 3, 0, 8.0, 32.0), (4, 2, 0, 8.0, 32.0), (8, 4, 0, 8.0, 32.0), (8, 1, 0, 8.0, 32.0),
                  (8, 3, 1, 2.0, 20000.0), (4, 2, 1, 2.0, 9.0), (8, 4, 1, 1.5, np.inf), (3, 1, 1, 2.0, 9.0)]
        if args.first == "mgm1":
            combos = [combos[3]] * 3 + combos
        for (NDIR, MGM, FH, P1, P2) in combos:
            for wmode in (0, 1):
                w8 = None
                if wmode:
                    rng = np.random.default_rng(7)
                    w8 = np.where(rng.random((8, ny, nx)) < 0.5, 4.0 if FH == 0 else 0.3, 1.0).astype(np.float32)
                faulthandler.dump_traceback_later(args.case_timeout, exit=True)
                print("  start NDIR=%d MGM=%d FH=%d w=%d" % (NDIR, MGM, FH, wmode), flush=True)
                Sa, oa, ca, lra = orc.mgm(C, dmin, P1, P2, NDIR, MGM, FH, 1, w8, dump_lr=True)
                print("  oracle done", flush=True)
                t0 = time.time()
                Sb, ob, cb = ctx.aggregate(cv, P1, P2, NDIR, MGM, FH, 1, w8, None, want_S=True)
                dt = time.time() - t0
                print("  hip done", flush=True)
                lrd = [ndiff(lra[p], ctx.debug_lr(cv, p)) for p in range(NDIR)]
                Sd = ndiff(Sa, Sb.download())
                fin = np.isfinite(ca)
                od = int(np.sum(oa[fin] != ob[fin]))
                cd = ndiff(ca, cb)
                # fused vfit
                _, orf, crf = ctx.aggregate(cv, P1, P2, NDIR, MGM, FH, 1, w8, "vfit", want_S=False)
                ora, cra = orc.refine(Sa, dmin, "vfit", np.where(fin, oa, dmin), ca)
                rd = ndiff(ora[fin], orf[fin]) + ndiff(cra, crf)
                # stand-alone refine
                o2, c2 = ctx.refine(Sb, "vfit", np.where(fin, ob, dmin), cb)
                rd2 = ndiff(ora[fin], o2[fin]) + ndiff(cra, c2)
                Sb.free()
                ok = (sum(lrd) == 0 and Sd == 0 and od == 0 and cd == 0 and rd == 0 and rd2 == 0)
                bad += not ok
                print("AGG %dx%dx%d NDIR=%d MGM=%d FH=%d P=(%g,%g) w=%d : Lr=%s S=%d out=%d cost=%d vfit=%d/%d  %.0fms %s"
                      % (nx, ny, L, NDIR, MGM, FH, P1, P2, wmode, lrd, Sd, od, cd, rd, rd2, dt * 1e3,
                         "ok" if ok else "MISMATCH"), flush=True)
        cv.free()

    if args.big:
        for (nx, ny, L, NDIR, MGM, FH, P1, P2) in [(1920, 1080, 128, 4, 2, 0, 8.0, 32.0),
                                                    (1920, 1080, 256, 8, 3, 0, 8.0, 32.0),
                                                    (1920, 1080, 256, 8, 3, 1, 2.0, 20000.0)]:
            C = synth.raw_volume(nx, ny, L)
            cv = ctx.upload_volume(C, 0)
            ctx.timing(True)
            for rep in range(3):
                ctx.timing_reset()
                _, o, c = ctx.aggregate_dev(cv, P1, P2, NDIR, MGM, FH, 1, None, "vfit")
                ctx.synchronize()
                print("BIG %dx%dx%d NDIR=%d MGM=%d FH=%d :" % (nx, ny, L, NDIR, MGM, FH), ctx.timings(), flush=True)
                o.free(); c.free()
            ctx.timing(False)
            cv.free()
    print("TOTAL BAD", bad)
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
