"""mgm_wta_windowed_dev / mgm_update_ranges_dev: the part of main()'s TSGM_ITER loop (mgm.cc:377-388) that changes from
one iteration to the next.  End-to-end parity with the reference binary is in test_gpu_cli.py (TSGM_ITER=2,3);
here: the invariants of the two entry points against a numpy restatement of mgm_core.cc:592-609 on the oracle's S."""
import numpy as np
import pytest

from helpers import ndiff
from mgm_amd import synth

pytestmark = pytest.mark.gpu


def numpy_windowed_wta(S, dmin, lo, hi, NDIR, fix):
    """first strict minimum of the corrected S over [lo, hi] per pixel; window labels outside the volume hold
    0 - (NDIR-1)*inf (fix) or 0 (no fix)"""
    ny, nx, L = S.shape
    with np.errstate(invalid="ignore"):
        vout = np.float32(0) - np.float32(NDIR - 1) * np.float32(np.inf) if fix else np.float32(0)
    out = np.full((ny, nx), np.nan, np.float32)
    cost = np.full((ny, nx), np.inf, np.float32)
    for y in range(ny):
        for x in range(nx):
            best, arg = np.float32(np.inf), None
            for d in range(int(lo[y, x]), int(hi[y, x]) + 1):
                v = S[y, x, d - dmin] if 0 <= d - dmin < L else vout
                if np.isfinite(v) and best > v:
                    best, arg = v, d
            if arg is not None:
                out[y, x], cost[y, x] = arg, best
    return out, cost


@pytest.mark.parametrize("fix", [1, 0])
def test_windowed_wta_matches_restatement(ctx, oracle, fix):
    nx, ny, L, dmin, NDIR = 41, 23, 64, -20, 8
    C = synth.raw_volume(nx, ny, L, seed=77, inf_frac=0.02)
    cv = ctx.upload_volume(C, dmin)
    _, o0, c0 = ctx.aggregate_dev(cv, 8.0, 32.0, NDIR, 3, 0, fix, None, None)
    So, oo, co = oracle.mgm(C, dmin, 8.0, 32.0, NDIR, 3, 0, fix)
    rng = np.random.default_rng(3)
    centre = rng.integers(dmin - 4, dmin + L + 4, size=(ny, nx))
    lo = (centre - rng.integers(1, 6, size=(ny, nx))).astype(np.float32) + np.float32(0.4)   # (int) truncates towards zero
    hi = (centre + rng.integers(1, 6, size=(ny, nx))).astype(np.float32) + np.float32(0.7)
    want_o, want_c = numpy_windowed_wta(So, dmin, lo, hi, NDIR, fix)
    o, c = ctx.wta_windowed_dev(cv, NDIR, fix, None, ctx.upload_image(lo), ctx.upload_image(hi))
    got_o, got_c = o.download()[0], c.download()[0]
    assert ndiff(got_c, want_c) == 0
    fin = np.isfinite(want_c)
    assert ndiff(got_o[fin], want_o[fin]) == 0
    # the whole range as window: mgm()'s own answer
    full_lo = ctx.upload_image(np.full((ny, nx), dmin, np.float32))
    full_hi = ctx.upload_image(np.full((ny, nx), dmin + L - 1, np.float32))
    o, c = ctx.wta_windowed_dev(cv, NDIR, fix, None, full_lo, full_hi)
    assert ndiff(c.download()[0], co) == 0 and ndiff(o.download()[0][np.isfinite(co)], oo[np.isfinite(co)]) == 0
    cv.free()


def test_update_ranges_restatement(ctx):
    nx, ny = 37, 19
    rng = np.random.default_rng(9)
    d = rng.integers(-30, 10, size=(ny, nx)).astype(np.float32) + rng.random((ny, nx)).astype(np.float32)
    d[rng.random((ny, nx)) < 0.1] = np.nan
    lo = np.full((ny, nx), -40, np.float32)
    hi = np.full((ny, nx), 20, np.float32)
    dl, dh = ctx.upload_image(lo), ctx.upload_image(hi)
    ctx.update_ranges_dev(ctx.upload_image(d), dl, dh, 3, 2)
    gmin, gmax = np.nanmin(d), np.nanmax(d)
    want_lo, want_hi = lo.copy(), hi.copy()
    for y in range(ny):
        for x in range(nx):
            a, b = np.float32(np.inf), np.float32(-np.inf)
            for dy in range(-2, 3):
                for dx in range(-2, 3):
                    v = d[min(max(y + dy, 0), ny - 1), min(max(x + dx, 0), nx - 1)]
                    a = min(a, (v if np.isfinite(v) else gmin) - np.float32(3))
                    b = max(b, (v if np.isfinite(v) else gmax) + np.float32(3))
            want_lo[y, x], want_hi[y, x] = a, b
    assert ndiff(dl.download()[0], want_lo) == 0 and ndiff(dh.download()[0], want_hi) == 0


def test_windowed_needs_the_last_aggregation(ctx):
    import mgm_amd
    a = ctx.upload_volume(synth.raw_volume(20, 10, 64, seed=1), 0)
    b = ctx.upload_volume(synth.raw_volume(20, 10, 64, seed=2), 0)
    ctx.aggregate_dev(a, 8.0, 32.0, 8, 3)
    lo, hi = ctx.upload_image(np.zeros((10, 20), np.float32)), ctx.upload_image(np.full((10, 20), 63, np.float32))
    with pytest.raises(mgm_amd.MgmError) as e:
        ctx.wta_windowed_dev(b, 8, 1, None, lo, hi)
    assert e.value.code == mgm_amd.MGM_ERR_INVALID
    a.free(), b.free()


def test_trim_releases_the_workspace_and_the_next_call_regrows_it(ctx):
    """mgm_ctx_trim: results after it are the same; the windowed WTA has nothing to work on until the next aggregation."""
    import mgm_amd
    C = synth.raw_volume(60, 40, 64, seed=77)
    cv = ctx.upload_volume(C, -10)
    _, o1, c1 = ctx.aggregate(cv, 8.0, 32.0, 8, 3, 0, 1, None, "vfit", want_S=False)
    ctx.trim()
    lo, hi = ctx.upload_image(np.full((40, 60), -10, np.float32)), ctx.upload_image(np.full((40, 60), 53, np.float32))
    with pytest.raises(mgm_amd.MgmError) as e:
        ctx.wta_windowed_dev(cv, 8, 1, "vfit", lo, hi)
    assert e.value.code == mgm_amd.MGM_ERR_INVALID
    _, o2, c2 = ctx.aggregate(cv, 8.0, 32.0, 8, 3, 0, 1, None, "vfit", want_S=False)
    assert ndiff(o1, o2) == 0 and ndiff(c1, c2) == 0
    for h in (cv, lo, hi):
        h.free()
