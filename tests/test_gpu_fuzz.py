"""Seeded random small cases against the oracle: odd image sizes (down to 1x1), label counts on both sides of every
kernel-selection threshold, every mode -- the corners the structured sweeps do not visit."""
import os

import numpy as np
import pytest

from helpers import ndiff
from mgm_amd import synth

pytestmark = pytest.mark.gpu

# a longer campaign on demand: MGM_FUZZ_N=4000 MGM_FUZZ_BASE=100000 python -m pytest tests/test_gpu_fuzz.py -m gpu
FUZZ_N = int(os.environ.get("MGM_FUZZ_N", "0"))
FUZZ_BASE = int(os.environ.get("MGM_FUZZ_BASE", "0"))
FUZZ_SCALE = int(os.environ.get("MGM_FUZZ_SCALE", "1"))  # image sides up to 40 x 30 times this (several bands per pass from 2 on)

LABELS = [1, 2, 3, 5, 31, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 383, 384, 385, 511, 512,
          513, 700, 768, 769, 1000, 1024, 1025]  # (round 4: 513..1024 take the second build at 12 / 16 labels per lane)


@pytest.mark.parametrize("seed", range(FUZZ_BASE, FUZZ_BASE + (FUZZ_N or 160)))
def test_random_case_vs_oracle(ctx, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    nx, ny = int(rng.integers(1, 40 * FUZZ_SCALE)), int(rng.integers(1, 30 * FUZZ_SCALE))
    if seed % 5 == 0 and FUZZ_SCALE == 1:
        nx, ny = [(1, 1), (1, 7), (7, 1), (2, 2), (3, 3), (2, 9), (9, 2), (4, 3)][(seed // 5) % 8]
    L = LABELS[int(rng.integers(0, len(LABELS)))]
    if nx * ny * L > 60000 * FUZZ_SCALE ** 2:
        L = LABELS[int(rng.integers(0, 8))]
    maxcost = int(rng.choice([24, 24, 254, 3000]))  # (beyond 254: no one-byte form -- fp32 costs for an uploaded volume)
    NDIR = int(rng.integers(1, 9))
    MGM = int(rng.integers(1, 5))
    FH = int(rng.integers(0, 2))
    P1, P2 = [(8.0, 32.0), (2.0, 9.0), (1.5, 20000.0), (2.0, np.inf), (0.5, 3.25)][int(rng.integers(0, 5))]
    fix = int(rng.integers(0, 2))
    dmin = int(rng.integers(-300, 300))
    integer = rng.random() < 0.6
    C = synth.raw_volume(nx, ny, L, seed=seed, maxcost=maxcost, inf_frac=float(rng.choice([0.0, 0.05, 0.5])))
    if not integer:
        C = (C * np.float32(1.0 / 3.0)).astype(np.float32)
    w8 = None
    if rng.random() < 0.3:
        w8 = np.where(rng.random((8, ny, nx)) < 0.5, np.float32(rng.choice([4.0, 0.3, 1.0 / 3.0])), np.float32(1)).astype(np.float32)
    refine = [None, "vfit", "parabola", "cubic", "parabolaOCV"][int(rng.integers(0, 5))]
    So, oo, co = oracle.mgm(C, dmin, P1, P2, NDIR, MGM, FH, fix, w8)
    if refine:
        oo2 = np.where(np.isfinite(co), oo, dmin).astype(np.float32)
        ro, rc = oracle.refine(So, dmin, refine, oo2, co)
    else:
        ro, rc = oo, co
    cv = ctx.upload_volume(C, dmin)
    S, o, c = ctx.aggregate(cv, P1, P2, NDIR, MGM, FH, fix, w8, refine, want_S=True)
    tag = (nx, ny, L, NDIR, MGM, FH, P1, P2, fix, integer, w8 is not None, refine)
    assert ndiff(S.download(), So) == 0, tag
    assert ndiff(c, rc) == 0, tag
    fin = np.isfinite(co)
    assert ndiff(o[fin], ro[fin]) == 0, tag
    S.free(), cv.free()


@pytest.mark.parametrize("seed", range(FUZZ_BASE, FUZZ_BASE + (FUZZ_N // 2 or 60)))
def test_random_costvolume_vs_oracle(ctx, oracle, seed):
    rng = np.random.default_rng(5000 + seed)
    nch = int(rng.choice([1, 3]))
    nx, ny = int(rng.integers(1, 50)), int(rng.integers(1, 25))
    vnx, vny = (nx, ny) if rng.random() < 0.7 else (max(1, nx - int(rng.integers(0, 5))), max(1, ny - int(rng.integers(0, 4))))
    L = int(rng.choice([1, 7, 33, 64, 100, 128, 192, 256]))
    dmin = int(rng.integers(-L, 10))
    pre = str(rng.choice(["none", "census", "sobelx", "gblur"]))
    dist = str(rng.choice(["ad", "sd", "census", "ncc", "btad", "btsd"]))
    win = int(rng.choice([3, 5, 7]))
    if "census" in (pre, dist) and (nch * (win * win - 1)) % 8:
        win = 3
    td = float(rng.choice([np.inf, 30.0, 7.0, 2.5]))
    if rng.random() < 0.4:  # widths that are multiples of four: every group of four pixels of k_cost_diffx / k_cost_btx / k_cost_census8x is whole
        vnx += (nx + 3) // 4 * 4 - nx
        nx = (nx + 3) // 4 * 4
    u = rng.integers(0, 256, size=(nch, ny, nx)).astype(np.float32)
    v = rng.integers(0, 256, size=(nch, vny, vnx)).astype(np.float32)
    if rng.random() < 0.15:  # half-integer samples: differences without a compact form (the volume is filled again in fp32)
        u, v = u * np.float32(0.5), v * np.float32(0.5)
    a = oracle.costvolume(u, v, dmin, dmin + L - 1, pre, dist, td, win)
    cv = ctx.costvolume_dev(ctx.upload_image(u), ctx.upload_image(v), dmin, dmin + L - 1, pre, dist, td, win)
    tag = (nch, nx, ny, vnx, vny, L, dmin, pre, dist, win, td)
    assert ndiff(cv.download(), a) == 0, tag
    # ... and through the aggregation: the compact copy K2 wrote next to the fp32 volume (one or two bytes per cost, or
    # none) is what the pass kernels and the winner search read
    if not np.isnan(a).any():
        NDIR, MGM, FH = int(rng.integers(1, 9)), int(rng.integers(1, 5)), int(rng.integers(0, 2))
        P1, P2 = [(8.0, 32.0), (2.0, 9.0), (6.0, 60000.0)][int(rng.integers(0, 3))]
        So, oo, co = oracle.mgm(a, dmin, P1, P2, NDIR, MGM, FH, 1)
        S, o, c = ctx.aggregate_dev(cv, P1, P2, NDIR, MGM, FH, 1, None, None, want_S=True)
        assert ndiff(S.download(), So) == 0 and ndiff(c.download()[0], co) == 0, (tag, NDIR, MGM, FH, P1, P2)
        fin = np.isfinite(co)
        assert ndiff(o.download()[0][fin], oo[fin]) == 0, (tag, NDIR, MGM, FH)
        S.free()
    cv.free()


@pytest.mark.parametrize("seed", range(FUZZ_BASE, FUZZ_BASE + (FUZZ_N // 4 or 60)))
def test_random_batch_vs_oracle(ctx, oracle, seed):
    """Several volumes per launch (volumes per wave at <= 128 labels, padding, compact and fp32 costs mixed by chance)."""
    rng = np.random.default_rng(9000 + seed)
    nb = int(rng.integers(1, 17))
    nx, ny = int(rng.integers(1, 40 * FUZZ_SCALE)), int(rng.integers(1, 30 * FUZZ_SCALE))
    L = int(rng.choice([5, 31, 64, 65, 100, 127, 128, 129, 192, 200, 256, 300, 384, 512]))
    if nb * nx * ny * L > 400000 * FUZZ_SCALE ** 2:
        L = int(rng.choice([31, 64, 100, 128]))
        nb = min(nb, 4)
    NDIR = int(rng.integers(1, 9))
    MGM = int(rng.integers(1, 5))
    FH = int(rng.integers(0, 2))
    P1, P2 = [(8.0, 32.0), (2.0, 9.0), (1.5, 20000.0), (2.0, np.inf), (0.5, 3.25)][int(rng.integers(0, 5))]
    fix = int(rng.integers(0, 2))
    dmin = int(rng.integers(-300, 300))
    allint = rng.random() < 0.7
    weighted = rng.random() < 0.25
    refine = [None, "vfit", "cubic"][int(rng.integers(0, 3))]
    Cs, cvs, w8s, w8h = [], [], [], []
    for b in range(nb):
        C = synth.raw_volume(nx, ny, L, seed=seed * 17 + b, inf_frac=float(rng.choice([0.0, 0.05, 0.4])))
        if not allint and rng.random() < 0.5:  # this volume keeps fp32 costs: the launch falls back to one kind for all
            C = (C * np.float32(1.0 / 3.0)).astype(np.float32)
        Cs.append(C)
        cvs.append(ctx.upload_volume(C, dmin))
        if weighted:
            w = np.where(rng.random((8, ny, nx)) < 0.5, np.float32(rng.choice([4.0, 0.3])), np.float32(1)).astype(np.float32)
            w8h.append(w)
            w8s.append(ctx.upload_image(w))
    S, outs, outcs = ctx.aggregate_batch_dev(cvs, P1, P2, NDIR, MGM, FH, fix, w8s if weighted else None, refine, want_S=True)
    tag = (nb, nx, ny, L, NDIR, MGM, FH, P1, P2, fix, allint, weighted, refine)
    for b in range(nb):
        So, oo, co = oracle.mgm(Cs[b], dmin, P1, P2, NDIR, MGM, FH, fix, w8h[b] if weighted else None)
        if refine:
            oo2 = np.where(np.isfinite(co), oo, dmin).astype(np.float32)
            ro, rc = oracle.refine(So, dmin, refine, oo2, co)
        else:
            ro, rc = oo, co
        assert ndiff(S[b].download(), So) == 0, (b, tag)
        assert ndiff(outcs[b].download().reshape(ny, nx), rc) == 0, (b, tag)
        fin = np.isfinite(co)
        assert ndiff(outs[b].download().reshape(ny, nx)[fin], ro[fin]) == 0, (b, tag)
    for x in cvs + S + w8s:
        x.free()
