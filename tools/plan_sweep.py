#!/usr/bin/env python3
"""Does the launch plan generalise?    Runs ON THE GPU BOX.

For shapes the heuristics of mgm_plan.hip were NOT tuned on -- 1280x720x96, 2560x1440x160, 3840x2160x128, 700x500x151,
4096x4096x192 -- at 1, 2 and 4 volumes per launch, time the pass kernel (a) under the plan the library picks by itself and
(b) under every combination of the overrides MGM_HIP_TUNE offers (workgroups per CU, per-XCD queues and their block size,
strips), each in a fresh process (the switches are read once per process), and print one table row per (shape, batch):
the heuristic's time, the best override's time and setting, and the loss in percent.

    python tools/plan_sweep.py [--quick] > profiles/r04_plan_sweep.txt
"""
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(1280, 720, 96, 8, 3, 0), (2560, 1440, 160, 8, 3, 1), (3840, 2160, 128, 4, 2, 0), (700, 500, 151, 4, 2, 0), (4096, 4096, 192, 8, 3, 0),
          (1920, 1080, 256, 8, 3, 1)]  # nx, ny, L, NDIR, MGM, FH  (the last one is a shape the plan WAS tuned on: the control)
CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
import mgm_amd
from mgm_amd import synth
nx, ny, L, NDIR, MGM, FH, B, steps = map(int, sys.argv[1:9])
P1, P2 = (2.0, 20000.0) if FH else (8.0, 32.0)
with mgm_amd.Context(0) as c:
    cvs = []
    for b in range(B):
        u, v, _ = synth.stereo_pair(nx, ny, -(L - 1) * 3 // 4, 0, seed=11 + b)
        du, dv = c.upload_image(u), c.upload_image(v)
        cvs.append(c.costvolume_dev(du, dv, -(L - 1), 0, "none", "census", float("inf"), 5))
    outs = outcs = None
    for it in range(2 + steps):
        if it == 2:
            c.synchronize(); c.timing(True); c.timing_reset()
        _, outs, outcs = c.aggregate_batch_dev(cvs, P1, P2, NDIR, MGM, FH, 1, None, "vfit", outs, outcs)
    c.synchronize()
    t = {}
    for n, ms in c.timings():
        t.setdefault(n, []).append(ms)
    print(json.dumps({k: float(np.mean(v)) for k, v in t.items()}))
''' % ROOT


def run(shape, B, tune, steps):
    env = dict(os.environ)
    env.pop("MGM_HIP_TUNE", None)
    if tune:
        env["MGM_HIP_TUNE"] = tune
    r = subprocess.run([sys.executable, "-c", CHILD] + [str(x) for x in shape] + [str(B), str(steps)], env=env, capture_output=True, text=True, timeout=600)
    if r.returncode:
        return None
    t = json.loads(r.stdout.strip().splitlines()[-1])
    return t.get("k_pass2", t.get("k_pass"))


def main():
    quick = "--quick" in sys.argv
    overrides = []
    for wg, xq, st in itertools.product((1, 2), (0, 1), (0, 1)):
        ks = (0, 2) if xq else (None,)
        for k in ks:
            overrides.append("wg_per_cu=%d,xcdq=%d,strips=%d" % (wg, xq, st) + (",xcdq_k=%d" % k if k is not None else ""))
    if quick:
        overrides = overrides[::3]
    print("# launch plan of mgm_plan.hip vs every override (pass kernel ms per launch; fresh process per setting; census 5x5 costs)")
    print("# %-22s %2s  %10s  %10s  %-42s %7s" % ("shape (NDIR/TSGM/FH)", "B", "heuristic", "best", "best setting", "loss"))
    worst = 0.0
    for shape in SHAPES:
        for B in (1, 2, 4):
            if 4.0 * shape[0] * shape[1] * max(shape[2], 64) * shape[3] * B > 200e9:
                continue  # (does not fit the device)
            steps = 3 if shape[0] * shape[1] > 4e6 else 6
            # (single launches of this kind vary by several percent from process to process -- the same plan measured 9.5 and
            # 10.3 ms in two sweeps of round 4 --, so: the heuristic is the best of three processes, every override one, and
            # an override that beats the heuristic has to do so again, twice, with its best of three)
            h = min(x for x in (run(shape, B, "", steps) for _ in range(3)) if x is not None)
            best, best_o = h, "(the heuristic)"
            for o in overrides:
                t = run(shape, B, o, steps)
                if t is not None and t < best:
                    t = min([t] + [x for x in (run(shape, B, o, steps) for _ in range(2)) if x is not None])
                    h = min([h] + [x for x in (run(shape, B, "", steps),) if x is not None])
                    if t < min(best, h):
                        best, best_o = t, o
            best = min(best, h)
            if best == h:
                best_o = "(the heuristic)"
            loss = (h / best - 1.0) * 100.0
            worst = max(worst, loss)
            print("%-24s %2d  %10.3f  %10.3f  %-42s %6.1f%%" % ("%dx%dx%d (%d/%d/%d)" % shape, B, h, best, best_o, loss), flush=True)
    print("# worst loss of the heuristic against the best override: %.1f %%" % worst)


if __name__ == "__main__":
    main()
