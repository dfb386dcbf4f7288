"""Live comparison of the oracle with the compiled reference (oracle/_ref), where it
exists: the build container always, the GPU box when the prebuilt .so travelled.
Seeded random sweeps over every mode; bit-exact."""
import numpy as np
import pytest

from helpers import labels_equal, ndiff

CASES = [(NDIR, MGM, FH, P1, P2, wmode)
         for NDIR in (1, 4, 8) for MGM in (1, 2, 3, 4)
         for (FH, P1, P2) in ((0, 8.0, 32.0), (1, 2.0, 9.0), (1, 1.5, 20000.0), (1, 2.0, np.inf))
         for wmode in (0, 1, 2)]


def test_aggregation_sweep(oracle, reference):
    rng = np.random.default_rng(11)
    for (nx, ny, L) in [(23, 17, 12), (9, 31, 5)]:
        C = rng.integers(0, 25, size=(ny, nx, L)).astype(np.float32)
        C[rng.random((ny, nx, L)) < 0.05] = np.inf
        C[..., 0][~np.isfinite(C).any(axis=2)] = 0
        for (NDIR, MGM, FH, P1, P2, wmode) in CASES:
            w8 = None
            if wmode:
                w8 = np.where(rng.random((8, ny, nx)) < 0.5, 4.0 if wmode == 1 else 0.3, 1.0).astype(np.float32)
            a = oracle.mgm(C, -3, P1, P2, NDIR, MGM, FH, 1, w8)
            b = reference.mgm(C, -3, P1, P2, NDIR, MGM, FH, 1, w8)
            tag = (nx, ny, L, NDIR, MGM, FH, P1, P2, wmode)
            assert ndiff(a[0], b[0]) == 0, tag
            assert ndiff(a[2], b[2]) == 0, tag
            assert labels_equal(a[1], b[1], a[2]), tag
            oo = np.where(np.isfinite(a[2]), a[1], -3).astype(np.float32)
            for meth in ("vfit", "cubic"):
                ra, rb = oracle.refine(a[0], -3, meth, oo, a[2]), reference.refine(b[0], -3, meth, oo, b[2])
                assert ndiff(ra[0], rb[0]) == 0 and ndiff(ra[1], rb[1]) == 0, tag + (meth,)


def test_costvolume_sweep(oracle, reference):
    win = reference.census_win()  # whatever CENSUS_NCC_WIN this process has (default 3)
    rng = np.random.default_rng(12)
    for nch in (1, 3):
        u = rng.integers(0, 256, size=(nch, 19, 31)).astype(np.float32)
        v = np.roll(u, 3, axis=2) + rng.integers(-2, 3, size=u.shape).astype(np.float32)
        v2 = rng.integers(0, 256, size=(nch, 21, 27)).astype(np.float32)  # a different-size right image
        for (dmin, dmax) in [(-5, 4), (2, 9), (-40, -33), (35, 41)]:
            for vv in (v, v2):
                for pre, dist in [("none", "ad"), ("none", "sd"), ("none", "census"), ("census", "ad"), ("sobelx", "ad"),
                                  ("gblur", "sd"), ("sobel_x", "ad"), ("none", "foo"), ("none", "ncc"), ("gblur", "ncc"),
                                  ("none", "btad"), ("sobelx", "btsd")]:
                    if "census" in (pre, dist) and (nch * (win * win - 1)) % 8:
                        continue
                    for td in (np.inf, 20.0):
                        a = oracle.costvolume(u, vv, dmin, dmax, pre, dist, td, win)
                        b = reference.costvolume(u, vv, dmin, dmax, pre, dist, td)
                        assert ndiff(a, b) == 0, (nch, dmin, dmax, pre, dist, td)
        assert ndiff(oracle.weights(u, 4.0, 5.0), reference.weights(u, 4.0, 5.0)) == 0
        if (nch * (win * win - 1)) % 8 == 0:
            assert np.array_equal(oracle.census(u, win // 2), reference.census(u, win // 2))


# ---- the steps main() applies right after the path, on maps with NaN labels and +INF costs ---------------------------------
from oracle import oracle as orc_mod  # noqa: E402


def nan_label_map(ny, nx, seed, frac_nan=0.15, frac_inf=0.03, lo=-20.0, hi=8.0, subpixel=True):
    """A disparity map as the NaN-faithful path leaves it: labels in [lo, hi] (sub-pixel where refined), NaN where a pixel
    had no finite S, in patches and isolated; optionally +-INF samples (cost maps)."""
    rng = np.random.default_rng(seed)
    m = rng.uniform(lo, hi, (ny, nx)).astype(np.float32)
    if not subpixel:
        m = np.rint(m).astype(np.float32)
    m[rng.random((ny, nx)) < frac_nan] = np.nan
    y0, x0 = rng.integers(0, ny - 4), rng.integers(0, nx - 6)
    m[y0:y0 + 4, x0:x0 + 6] = np.nan  # a patch wider than the small windows: all-NaN windows occur
    m[rng.random((ny, nx)) < frac_inf] = np.inf
    m[rng.random((ny, nx)) < frac_inf / 2] = -np.inf
    return m


@pytest.mark.skipif(not orc_mod.RefPost.available(), reason="oracle/_ref/libmgm_refpost.so was not built here")
@pytest.mark.parametrize("radius", [1, 2, 3, 9])
def test_median_restatement_vs_reference_with_nan_labels(radius):
    """oracle/post.py::median against the reference's median_filter (img_tools.h:203-238) on maps holding NaN and +-INF."""
    from oracle import post
    rp = orc_mod.RefPost()
    for seed in range(4):
        m = np.stack([nan_label_map(23, 31, 100 * radius + seed, frac_inf=0.03 if seed % 2 else 0.0) for _ in range(1 + seed % 2)])
        assert orc_mod.bits_equal(rp.median(m, radius), post.median(m, radius)), (radius, seed)


@pytest.mark.skipif(not orc_mod.RefPost.available(), reason="oracle/_ref/libmgm_refpost.so was not built here")
def test_leftright_restatement_vs_reference_with_nan_labels():
    """oracle/post.py::leftright against the reference's leftright_test (mgm.cc:68-91): NaN / infinite labels in either map,
    a right map of another width, several thresholds."""
    from oracle import post
    rp = orc_mod.RefPost()
    for seed, (tau, rnx) in enumerate([(1.0, 31), (0.5, 31), (2.0, 27), (1.0, 40), (0.0, 31)]):
        d = nan_label_map(19, 31, 7000 + seed, frac_inf=0.02)
        o = -nan_label_map(19, rnx, 7100 + seed, frac_inf=0.02, lo=-20.0, hi=8.0)
        assert orc_mod.bits_equal(rp.leftright(d, o, tau), post.leftright(d, o, tau)), (seed, tau, rnx)
        di = nan_label_map(19, 31, 7200 + seed, frac_inf=0.0, subpixel=False)  # integer labels: exact ties at the threshold
        assert orc_mod.bits_equal(rp.leftright(di, np.rint(o), tau), post.leftright(di, np.rint(o), tau)), (seed, "int")


# ---- RAGGED ranges (round 6): per-pixel Dvec ranges below the command line ---------------------------------------------------
def ragged_ranges(rng, ny, nx, hmin, hmax, kind):
    """Float range images inside the hull [hmin, hmax]; `kind` picks how wild they are."""
    if kind == "smooth":  # a slanted surface +- a window, small jitter: the -m/-M files of a coarse-to-fine run
        yy, xx = np.mgrid[0:ny, 0:nx]
        mid = hmin + (hmax - hmin) * (0.2 + 0.6 * xx / max(1, nx - 1)) + 0.1 * yy
        half = rng.integers(1, max(2, (hmax - hmin) // 4))
        lo = mid - half + rng.integers(-1, 2, size=(ny, nx))
        hi = mid + half + rng.integers(-1, 2, size=(ny, nx))
    elif kind == "wild":  # independent per pixel: neighbours overlap partly, nest, touch or are disjoint
        lo = rng.integers(hmin, hmax, size=(ny, nx)).astype(np.float64)
        hi = lo + rng.integers(0, 1 + (hmax - hmin) // 2, size=(ny, nx))
    else:  # "mixed": mostly the full hull, patches of narrow windows and one-label pixels
        lo = np.full((ny, nx), float(hmin))
        hi = np.full((ny, nx), float(hmax))
        m = rng.random((ny, nx)) < 0.3
        lo[m] = rng.integers(hmin, hmax, size=int(m.sum()))
        hi[m] = lo[m] + rng.integers(0, 4, size=int(m.sum()))
    lo = np.clip(lo, hmin, hmax)
    hi = np.clip(np.maximum(hi, lo), hmin, hmax)
    frac = rng.random((ny, nx)) * 0.9  # the reference's Dvec constructor truncates float ranges toward zero
    lo_f, hi_f = (lo + np.where(lo >= 0, frac, -frac)), (hi + np.where(hi >= 0, frac, -frac))
    lo_f[0, 0], hi_f[0, 0] = hmin, hmax  # the hull is attained
    return lo_f.astype(np.float32), hi_f.astype(np.float32)


def in_range_mask(lo, hi, hmin, L):
    d = hmin + np.arange(L)[None, None, :]
    return (d >= lo[..., None]) & (d <= hi[..., None])


RAGGED_MODES = [(NDIR, MGM, FH, P1, P2, wmode)
                for NDIR in (1, 4, 8) for MGM in (1, 2, 3, 4)
                for (FH, P1, P2) in ((0, 8.0, 32.0), (1, 2.0, 9.0), (1, 1.5, 20000.0), (1, 2.0, np.inf), (0, 4.0, np.inf))
                for wmode in (0, 1, 2)]


@pytest.mark.parametrize("kind", ["smooth", "wild", "mixed"])
def test_ragged_aggregation_sweep(oracle, reference, kind):
    """orc_mgm_ranged == the reference's mgm() called with range images: S inside every pixel's range, labels, costs, every
    update function (update_cost2 / costW / cost2_trunclinear incl. the boundary fix-up / costW_trunclinear), weights,
    with and without the over-count fix, and every refinement on the ragged S."""
    if not reference.has_ranged():
        pytest.skip("oracle/_ref/libmgm_ref.so predates the ranged entry points")
    rng = np.random.default_rng({"smooth": 21, "wild": 22, "mixed": 23}[kind])
    for (nx, ny, hmin, hmax) in [(23, 17, -9, 8), (9, 31, 3, 14), (14, 12, -30, -4)]:
        L = hmax - hmin + 1
        for rep, (NDIR, MGM, FH, P1, P2, wmode) in enumerate(RAGGED_MODES):
            dminI, dmaxI = ragged_ranges(rng, ny, nx, hmin, hmax, kind)
            lo, hi = orc_mod.int_ranges(dminI, dmaxI)
            C = rng.integers(0, 25, size=(ny, nx, L)).astype(np.float32)
            C[rng.random((ny, nx, L)) < 0.04] = np.inf
            own = in_range_mask(lo, hi, hmin, L)
            dead = ~(np.isfinite(C) & own).any(axis=2)  # no finite cost in the own range -> zeros (mgm_costvolume.h:414-421)
            C[dead] = 0
            w8 = None
            if wmode:
                w8 = np.where(rng.random((8, ny, nx)) < 0.5, 4.0 if wmode == 1 else 0.3, 1.0).astype(np.float32)
            FIX = int(rep % 2 == 0)
            a = oracle.mgm_ranged(C, hmin, lo, hi, P1, P2, NDIR, MGM, FH, FIX, w8)
            b = reference.mgm_ranged(C, hmin, dminI, dmaxI, P1, P2, NDIR, MGM, FH, FIX, w8)
            tag = (kind, nx, ny, hmin, hmax, NDIR, MGM, FH, P1, P2, wmode, FIX)
            assert ndiff(a[0], b[0]) == 0, tag   # S on the hull (+INF outside the ranges on both sides)
            assert ndiff(a[2], b[2]) == 0, tag
            assert labels_equal(a[1], b[1], a[2]), tag
            oo = np.where(np.isfinite(a[2]), a[1], lo).astype(np.float32)
            for meth in ("vfit", "parabola", "cubic", "parabolaOCV"):
                ra = oracle.refine_ranged(a[0], hmin, lo, hi, meth, oo, a[2])
                rb = reference.refine_ranged(b[0], hmin, dminI, dmaxI, meth, oo, b[2])
                assert ndiff(ra[0], rb[0]) == 0 and ndiff(ra[1], rb[1]) == 0, tag + (meth,)


def test_ragged_aggregation_narrowed_S_ranges(oracle, reference):
    """main()'s TSGM_ITER loop (mgm.cc:377-388): the cost volume keeps the ranges it was allocated with, mgm() is called with
    narrowed (or shifted: update_dmin_dmax can leave CC's range) images -- S, the search and the refinement follow those."""
    if not reference.has_ranged():
        pytest.skip("oracle/_ref/libmgm_ref.so predates the ranged entry points")
    rng = np.random.default_rng(31)
    nx, ny, hmin, hmax = 21, 15, -12, 9
    L = hmax - hmin + 1
    for rep in range(12):
        dminI, dmaxI = ragged_ranges(rng, ny, nx, hmin, hmax, ["smooth", "wild", "mixed"][rep % 3])
        lo, hi = orc_mod.int_ranges(dminI, dmaxI)
        shmin, shmax = hmin - 4, hmax + 5
        sminI, smaxI = ragged_ranges(rng, ny, nx, shmin, shmax, ["wild", "smooth"][rep % 2])
        slo, shi = orc_mod.int_ranges(sminI, smaxI)
        C = rng.integers(0, 25, size=(ny, nx, L)).astype(np.float32)
        NDIR, MGM, FH = (1, 4, 8)[rep % 3], (1, 2, 3, 4)[rep % 4], rep % 2
        FIX = int(rep % 4 < 2)
        a = oracle.mgm_ranged(C, hmin, lo, hi, 2.0, 12.0, NDIR, MGM, FH, FIX, None, (slo, shi, shmin, shmax))
        b = reference.mgm_ranged(C, hmin, dminI, dmaxI, 2.0, 12.0, NDIR, MGM, FH, FIX, None, (sminI, smaxI, shmin, shmax))
        tag = (rep, NDIR, MGM, FH, FIX)
        assert ndiff(a[0], b[0]) == 0, tag
        assert ndiff(a[2], b[2]) == 0, tag
        assert labels_equal(a[1], b[1], a[2]), tag
        oo = np.where(np.isfinite(a[2]), a[1], slo).astype(np.float32)
        ra = oracle.refine_ranged(a[0], shmin, slo, shi, "vfit", oo, a[2])
        rb = reference.refine_ranged(b[0], shmin, sminI, smaxI, "vfit", oo, b[2])
        assert ndiff(ra[0], rb[0]) == 0 and ndiff(ra[1], rb[1]) == 0, tag


def test_ragged_costvolume_sweep(oracle, reference):
    if not reference.has_ranged():
        pytest.skip("oracle/_ref/libmgm_ref.so predates the ranged entry points")
    win = reference.census_win()
    rng = np.random.default_rng(41)
    for nch in (1, 3):
        u = rng.integers(0, 256, size=(nch, 19, 31)).astype(np.float32)
        v = np.roll(u, 3, axis=2) + rng.integers(-2, 3, size=u.shape).astype(np.float32)
        v2 = rng.integers(0, 256, size=(nch, 21, 27)).astype(np.float32)
        for (hmin, hmax) in [(-7, 6), (25, 36), (-45, -30)]:  # (the last two: ranges that leave the right image altogether)
            for kind in ("smooth", "wild", "mixed"):
                dminI, dmaxI = ragged_ranges(rng, 19, 31, hmin, hmax, kind)
                lo, hi = orc_mod.int_ranges(dminI, dmaxI)
                for vv in (v, v2):
                    for pre, dist in [("none", "ad"), ("none", "census"), ("sobelx", "sd"), ("none", "ncc"), ("gblur", "btad")]:
                        if "census" in (pre, dist) and (nch * (win * win - 1)) % 8:
                            continue
                        for td in (np.inf, 20.0):
                            a = oracle.costvolume_ranged(u, vv, lo, hi, hmin, hmax, pre, dist, td, win)
                            b = reference.costvolume_ranged(u, vv, dminI, dmaxI, hmin, hmax, pre, dist, td)
                            assert ndiff(a, b) == 0, (nch, hmin, hmax, kind, pre, dist, td)


def test_ragged_oracle_with_uniform_ranges_is_the_uniform_oracle(oracle):
    """The two restatements agree where both apply (no reference needed: runs on the GPU box too)."""
    rng = np.random.default_rng(51)
    nx, ny, L, hmin = 19, 13, 11, -4
    C = rng.integers(0, 25, size=(ny, nx, L)).astype(np.float32)
    lo = np.full((ny, nx), hmin, np.int32)
    hi = np.full((ny, nx), hmin + L - 1, np.int32)
    for (NDIR, MGM, FH, P1, P2, wmode) in RAGGED_MODES[::5]:
        w8 = None if not wmode else np.where(rng.random((8, ny, nx)) < 0.5, 4.0, 1.0).astype(np.float32)
        a = oracle.mgm_ranged(C, hmin, lo, hi, P1, P2, NDIR, MGM, FH, 1, w8)
        b = oracle.mgm(C, hmin, P1, P2, NDIR, MGM, FH, 1, w8)
        assert ndiff(a[0], b[0]) == 0 and ndiff(a[1], b[1]) == 0 and ndiff(a[2], b[2]) == 0
