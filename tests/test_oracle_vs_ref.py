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
