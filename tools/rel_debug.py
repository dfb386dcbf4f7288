"""Development aid: one random case of tests/test_gpu_rel.py by seed, with the full parameter tuple and where the two paths differ."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import mgm_amd  # noqa: E402
from mgm_amd import synth  # noqa: E402
from test_gpu_rel import ndiff, ranges  # noqa: E402


def case(seed):
    rng = np.random.default_rng(991000 + seed)
    nx, ny = int(rng.integers(20, 140)), int(rng.integers(18, 100))
    dmin = -int(rng.integers(20, 120))
    dmax = int(rng.integers(0, 30))
    half = int(rng.integers(1, 29))
    FH = int(rng.integers(0, 2))
    MGM = int(rng.choice([1, 3, 4]))
    NDIR = int(rng.choice([1, 2, 4, 8]))
    P1 = float(rng.choice([0.75, 1.5, 2.0, 8.0]))
    P2 = float(rng.choice([9.0, 32.0, 40.0, 20000.0]))
    wkind = rng.choice(["none", "three", "image"])
    refine = rng.choice([None, "vfit", "parabola", "cubic", "parabolaOCV"])
    fix = int(rng.integers(0, 2))
    win = int(rng.choice([3, 5]))
    u, v, gt = synth.stereo_pair(nx, ny, dmin * 3 // 4, max(0, dmax * 3 // 4), seed=seed)
    lo, hi = ranges(gt, dmin, dmax, half, seed, jitter=int(rng.integers(0, 3)))
    print("seed", seed, dict(nx=nx, ny=ny, dmin=dmin, dmax=dmax, half=half, FH=FH, MGM=MGM, NDIR=NDIR, P1=P1, P2=P2, wkind=str(wkind), refine=refine, fix=fix, win=win),
          "max window", int((hi - lo).max()) + 1)
    res = {}
    for tag, rel, tune in (("rel", "2", ""), ("rel_gather", "2", "rel_direct=0"), ("hull", "0", "")):
        os.environ["MGM_HIP_REL"] = rel
        os.environ["MGM_HIP_TUNE"] = tune
        with mgm_amd.Context(0) as ctx:
            cv = ctx.costvolume(u, v, lo, hi, "none", "census", float("inf"), win)
            w8 = None
            if wkind == "three":
                w8 = ctx.upload_image(np.random.default_rng(seed).choice(np.array([1.0, 2.5, 4.0], np.float32), size=(8, ny, nx), p=[0.6, 0.25, 0.15]))
            elif wkind == "image":
                w8 = ctx.weights_dev(ctx.upload_image(u), 4.0, 12.0)
            ctx.timing(True)
            _, o, c = ctx.aggregate_dev(cv, P1, P2, NDIR, MGM, FH, fix, w8, None)
            names = [n for n, _ in ctx.timings()]
            res[tag] = (o.download(), c.download(), names)
    for tag in ("rel", "rel_gather"):
        a, b = res[tag], res["hull"]
        d = np.argwhere(~((a[1] == b[1]) | (np.isnan(a[1]) & np.isnan(b[1]))))
        print(" ", tag, [n for n in a[2] if "pass" in n or "cost" in n], "vs", [n for n in b[2] if "pass" in n], "label diffs", int(ndiff(a[0], b[0])), "cost diffs", len(d),
              "first", d[:4].tolist(), "rows", sorted(set(d[:, 1].tolist()))[:12], "cols", sorted(set(d[:, 2].tolist()))[:12])


for s in sys.argv[1:]:
    case(int(s))
