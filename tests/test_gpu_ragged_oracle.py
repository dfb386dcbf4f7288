"""RAGGED volumes (per-pixel disparity ranges) against the ragged ORACLE -- oracle/mgm_oracle.c::orc_mgm_ranged, which
tests/test_oracle_vs_ref.py pins on the compiled reference's mgm() called with range images (S, labels, costs, every update
function, the boundary fix-up).  Round 5 compared the range-proportional kernels (k_pass_rel / k_wta_rel) with the dense-hull
kernels of the same library only; here BOTH are compared with the reference's arithmetic, below the command line:

  * the cost volume on the hull (K2: k_cost_census_rel -> k_rel_expand, or the general kernel),
  * every pass's Lr volume (mgm_debug_download_lr: the range-proportional slabs expanded to the hull) -- stronger than S, which is
    their ordered sum,
  * S itself where the library hands it out (the dense-hull path),
  * labels, costs and every refinement,

at small sizes over the kernel's limits (windows of 60 / 61 / 62 labels and 63 = one past it, heights that are not a multiple of
the 16-line band, neighbour shifts of more than 32 slots), and at BASELINE's 1920x1080 (cfg3r / cfg3hr: one volume and a batch of
four).  The timings tell which kernel ran."""
import os

import numpy as np
import pytest

import mgm_amd
from helpers import ndiff
from mgm_amd import synth
from oracle import oracle as orc_mod

pytestmark = pytest.mark.gpu


def own_mask(lo, hi, hmin, L):
    d = hmin + np.arange(L)[None, None, :]
    return (d >= lo[..., None]) & (d <= hi[..., None])


def window_ranges(gt, dmin, dmax, below, above, seed, jitter):
    rng = np.random.default_rng(seed)
    lo = gt - below + (rng.integers(-jitter, jitter + 1, size=gt.shape) if jitter else 0)
    hi = gt + above + (rng.integers(-jitter, jitter + 1, size=gt.shape) if jitter else 0)
    lo, hi = np.clip(lo, dmin, dmax), np.clip(hi, dmin, dmax)
    hi = np.maximum(hi, lo)
    return lo.astype(np.float32), hi.astype(np.float32)


def exact_width_ranges(gt, dmin, dmax, width, seed):
    """Every pixel's window is EXACTLY `width` labels wide (the kernel's limit is 62), placed around the true disparity."""
    rng = np.random.default_rng(seed)
    lo = np.clip(gt - width // 2 + rng.integers(-2, 3, size=gt.shape), dmin, dmax - width + 1)
    return lo.astype(np.float32), (lo + width - 1).astype(np.float32)


def jumpy_ranges(nx, ny, dmin, dmax, width, seed):
    """Narrow windows whose position jumps by up to the whole hull between neighbours: shifts far beyond 32 slots, windows
    that do not overlap their neighbours' at all."""
    rng = np.random.default_rng(seed)
    lo = rng.integers(dmin, dmax - width + 2, size=(ny, nx))
    lo[:, ::3] = dmin  # columns pinned to the two ends of the hull
    lo[::4, :] = dmax - width + 1
    return lo.astype(np.float32), (lo + width - 1 - rng.integers(0, max(1, width // 2), size=(ny, nx))).astype(np.float32)


def run_hip(ctx, cv, mode, P1, P2, NDIR, MGM, FH, fix, w8, refine, want_S=False, lr=True):
    os.environ["MGM_HIP_REL"] = mode
    try:
        ctx.timing(True)
        ctx.timing_reset()
        S, o, c = ctx.aggregate_dev(cv, P1, P2, NDIR, MGM, FH, fix, w8, refine, want_S=want_S)
        names = [n for n, _ in ctx.timings()]
        ctx.timing(False)
        out = dict(o=o.download()[0], c=c.download()[0], names=names)
        if lr:
            out["lr"] = [ctx.debug_lr(cv, p) for p in range(NDIR)]
        if S is not None:
            out["S"] = S.download()
            S.free()
        o.free(), c.free()
        return out
    finally:
        os.environ.pop("MGM_HIP_REL", None)


SMALL = [
    # name, nx, ny, dmin, dmax, ranges, FH, MGM, NDIR, P1, P2, weights, refine, fix, expect k_pass_rel with MGM_HIP_REL=2
    ("w62", 97, 45, -90, 10, ("exact", 62), 1, 3, 8, 2.0, 20000.0, None, "vfit", 1, True),
    ("w61", 97, 45, -90, 10, ("exact", 61), 0, 3, 8, 8.0, 32.0, None, "cubic", 1, True),
    ("w60_t4", 80, 37, -90, 10, ("exact", 60), 1, 4, 8, 1.5, 9.0, None, "parabola", 0, True),
    # round 6: 128 slots per pixel take windows of up to 126 labels; two-byte cost codes take colour AD and SD
    ("w63_in_128_slots", 97, 45, -90, 10, ("exact", 63), 1, 3, 8, 2.0, 20000.0, None, "vfit", 1, True),
    ("w100_hirsch_t4", 80, 37, -160, 10, ("exact", 100), 0, 4, 8, 8.0, 32.0, None, "cubic", 1, True),
    ("w125_fh", 70, 35, -200, 10, ("exact", 125), 1, 3, 8, 2.0, 30.0, None, "parabola", 0, True),
    ("w126_fh_weights", 66, 35, -200, 10, ("exact", 126), 1, 3, 8, 2.0, 20000.0, "image", "vfit", 1, True),
    ("w126_hirsch_three_weights", 66, 35, -200, 10, ("exact", 126), 0, 3, 4, 8.0, 32.0, "three", None, 1, True),
    ("w126_t2_hirsch", 66, 35, -200, 10, ("exact", 126), 0, 2, 8, 8.0, 32.0, None, "vfit", 1, True),
    ("w127_past_the_limit", 66, 35, -200, 10, ("exact", 127), 1, 3, 8, 2.0, 20000.0, None, "vfit", 1, False),
    ("wide_jumps_fh", 71, 52, -300, 20, ("jumpy", 90), 1, 3, 8, 2.0, 20000.0, None, "vfit", 1, True),
    ("ad_colour_two_bytes_fh", 90, 40, -60, 0, ("win", 12, 14, 3), 1, 3, 8, 6.0, 60.0, None, "vfit", 1, True, "ad", 3),
    ("ad_colour_two_bytes_hirsch_weights", 90, 40, -60, 0, ("win", 12, 14, 3), 0, 4, 8, 24.0, 96.0, "image", "cubic", 1, True, "ad", 3),
    ("ad_colour_t2_wide", 70, 36, -150, 0, ("exact", 101), 0, 2, 4, 24.0, 96.0, None, None, 1, True, "ad", 3),
    ("sd_grey_two_bytes_wide_fh", 70, 36, -150, 0, ("exact", 90), 1, 3, 8, 40.0, 4000.0, None, "vfit", 0, True, "sd", 1),
    ("ncc_fp32_costs_fh", 90, 40, -60, 0, ("win", 12, 14, 3), 1, 3, 8, 2.0, 30.0, None, "vfit", 1, True, "ncc", 1),
    ("btad_fp32_costs_hirsch_weights", 90, 40, -60, 0, ("win", 12, 14, 3), 0, 4, 8, 8.0, 32.0, "image", "cubic", 1, True, "btad", 1),
    ("btsd_colour_fp32_costs_t2_wide", 70, 36, -150, 0, ("exact", 90), 0, 2, 4, 24.0, 96.0, None, None, 1, True, "btsd", 3),
    ("ad_grey_one_byte", 90, 40, -60, 0, ("win", 12, 14, 3), 1, 3, 8, 2.0, 30.0, None, "vfit", 1, True, "ad", 1),
    ("h17", 64, 17, -60, 0, ("win", 20, 22, 3), 1, 3, 8, 2.0, 20000.0, None, "vfit", 1, True),
    ("h33_w3", 50, 33, -60, 0, ("win", 20, 22, 3), 0, 3, 8, 8.0, 32.0, "three", "parabolaOCV", 1, True),
    ("h47_image_weights", 120, 47, -60, 0, ("win", 12, 17, 2), 1, 4, 8, 2.0, 40.0, "image", "vfit", 1, True),
    ("jumps_fh", 71, 52, -200, 20, ("jumpy", 9), 1, 3, 8, 2.0, 20000.0, None, "vfit", 1, True),
    ("jumps_hirsch", 71, 52, -200, 20, ("jumpy", 14), 0, 4, 8, 8.0, 32.0, None, None, 0, True),
    ("jumps_t1_o4", 83, 40, -120, 0, ("jumpy", 30), 1, 1, 4, 2.0, 30.0, None, "vfit", 1, True),
    ("t2_hirsch", 90, 41, -60, 0, ("win", 10, 12, 3), 0, 2, 8, 8.0, 32.0, None, "vfit", 1, True),   # update_cost2 (round 6: on the range-proportional kernels)
    ("t2_hirsch_jumps_o4", 75, 38, -150, 0, ("jumpy", 20), 0, 2, 4, 8.0, 32.0, None, "cubic", 0, True),
    ("t2_hirsch_ones_as_weights", 64, 33, -60, 0, ("win", 10, 12, 3), 0, 2, 8, 8.0, 32.0, "ones", "vfit", 1, True),  # planes of ones: the reference runs UNWEIGHTED (update_cost2)
    ("t2_hirsch_image_weights", 64, 33, -60, 0, ("win", 10, 12, 3), 0, 2, 8, 8.0, 32.0, "image", None, 1, True),     # update_costW with two neighbours
    ("t2_fh_boundary_fix", 90, 41, -60, 0, ("win", 10, 12, 3), 1, 2, 8, 2.0, 30.0, None, "vfit", 1, True),  # update_cost2_trunclinear + fix-up (round 6: k_pass_rel FH2)
    ("t2_fh_jumps", 61, 44, -100, 0, ("jumpy", 11), 1, 2, 4, 2.0, 30.0, None, "cubic", 0, True),   # neighbours' ranges disjoint from the pixel's: the fix-up's ramps
    ("t2_fh_jumps_narrow", 57, 39, -200, 0, ("jumpy", 3), 1, 2, 8, 1.5, 20000.0, None, "vfit", 1, True),
    ("t2_fh_wide_128_slots", 66, 35, -200, 10, ("exact", 110), 1, 2, 8, 2.0, 40.0, None, "parabola", 1, True),
    ("t2_fh_colour_ad", 80, 36, -60, 0, ("win", 9, 13, 4), 1, 2, 4, 6.0, 90.0, None, "vfit", 1, True, "ad", 3),
    ("t2_fh_ones_as_weights", 64, 33, -60, 0, ("win", 10, 12, 3), 1, 2, 8, 2.0, 30.0, "ones", "vfit", 1, True),
    ("t2_fh_image_weights", 61, 44, -100, 0, ("jumpy", 11), 1, 2, 4, 2.0, 30.0, "image", "vfit", 1, True),  # update_costW_trunclinear, two neighbours
    # shapes at the corners of the anti-diagonal walk of the form-1 passes (bands of diagonals that start late, end early, or hold one pixel)
    ("three_rows_fh", 90, 3, -40, 0, ("win", 8, 10, 2), 1, 3, 8, 2.0, 30.0, None, "vfit", 1, True),
    ("three_columns_hirsch", 3, 70, -40, 0, ("win", 8, 10, 2), 0, 3, 8, 8.0, 32.0, None, None, 1, True),
    ("one_row", 50, 1, -40, 0, ("win", 8, 10, 2), 1, 3, 8, 2.0, 30.0, None, None, 0, True),
    ("one_column", 1, 50, -40, 0, ("win", 8, 10, 2), 0, 2, 8, 8.0, 32.0, None, None, 0, True),
    ("square_16", 16, 16, -40, 0, ("win", 8, 10, 2), 1, 3, 8, 2.0, 30.0, "image", "vfit", 1, True),
    ("square_17_t2_fh", 17, 17, -40, 0, ("win", 8, 10, 2), 1, 2, 8, 2.0, 30.0, None, "cubic", 1, True),
    ("tall_15x33_three_weights", 15, 33, -40, 0, ("jumpy", 12), 1, 3, 8, 2.0, 30.0, "three", "vfit", 1, True),
    ("wide_33x15_t1", 33, 15, -40, 0, ("jumpy", 12), 0, 1, 8, 8.0, 32.0, None, "parabola", 0, True),
    ("wide_128_slots_49x31", 49, 31, -200, 10, ("exact", 100), 1, 3, 8, 2.0, 40.0, None, "vfit", 1, True),
    # ... and many bands of anti-diagonals either way round (more lines than pixels per line, and the reverse)
    ("portrait_300x700_fh", 300, 700, -60, 0, ("win", 10, 12, 3), 1, 3, 8, 2.0, 20000.0, None, "vfit", 1, True),
    ("landscape_700x300_hirsch_weights", 700, 300, -60, 0, ("win", 10, 12, 3), 0, 3, 8, 8.0, 32.0, "image", None, 1, True),
]


@pytest.mark.parametrize("case", SMALL, ids=lambda c: c[0])
def test_ragged_small_vs_oracle(oracle, case):
    name, nx, ny, dmin, dmax, rk, FH, MGM, NDIR, P1, P2, wkind, refine, fix, expect_rel = case[:15]
    cost, nch = (case[15], case[16]) if len(case) > 15 else ("census", 1)
    u, v, gt = synth.stereo_pair(nx, ny, dmin * 3 // 4, min(0, dmax), seed=101 + nx + ny, nch=nch)
    if rk[0] == "exact":
        dminI, dmaxI = exact_width_ranges(gt, dmin, dmax, rk[1], 7 + ny)
    elif rk[0] == "win":
        dminI, dmaxI = window_ranges(gt, dmin, dmax, rk[1], rk[2], 7 + ny, rk[3])
    else:
        dminI, dmaxI = jumpy_ranges(nx, ny, dmin, dmax, rk[1], 7 + ny)
    lo, hi = orc_mod.int_ranges(dminI, dmaxI)
    hmin, hmax = int(lo.min()), int(hi.max())
    L = hmax - hmin + 1
    own = own_mask(lo, hi, hmin, L)
    Ca = oracle.costvolume_ranged(u, v, lo, hi, hmin, hmax, "none", cost, np.inf, 5)
    oracle.set_threads(min(16, len(os.sched_getaffinity(0))))
    with mgm_amd.Context(0) as ctx:
        os.environ["MGM_HIP_REL"] = "2"
        cv = ctx.costvolume(u, v, dminI, dmaxI, "none", cost, float("inf"), 5)
        os.environ.pop("MGM_HIP_REL")
        assert cv.dims == (nx, ny, hmin, hmax)
        w8 = w8h = None
        if wkind == "three":
            w8h = np.random.default_rng(3).choice(np.array([1.0, 2.5, 4.0], np.float32), size=(8, ny, nx), p=[0.6, 0.25, 0.15])
            w8 = ctx.upload_image(w8h)
        elif wkind == "image":
            w8 = ctx.weights_dev(ctx.upload_image(u), 4.0, 12.0)
            w8h = w8.download()
            assert ndiff(w8h, oracle.weights(u, 4.0, 12.0)) == 0
        elif wkind == "ones":
            w8h = np.ones((8, ny, nx), np.float32)
            w8 = ctx.upload_image(w8h)
        try:
            Sa, oa, ca, lra = oracle.mgm_ranged(Ca, hmin, lo, hi, P1, P2, NDIR, MGM, FH, fix, w8h, dump_lr=True)
        finally:
            oracle.set_threads(1)
        if refine:  # (a pixel without a finite S keeps its NaN label on both sides)
            oa_r, ca_r = oracle.refine_ranged(Sa, hmin, lo, hi, refine, oa, ca)
        else:
            oa_r, ca_r = oa, ca
        for mode in ("2", "0"):
            got = run_hip(ctx, cv, mode, P1, P2, NDIR, MGM, FH, fix, w8, refine, want_S=True)  # (round 6: S from the range-proportional kernels too, k_rel_S)
            ran_rel = "k_pass_rel" in got["names"]
            if mode == "0":
                assert not ran_rel
            elif expect_rel is not None:
                assert ran_rel == expect_rel, (name, got["names"])
                assert ("k_rel_S" in got["names"]) == expect_rel, (name, got["names"])
            for p in range(NDIR):
                d = int(np.sum((got["lr"][p].view(np.uint32) != lra[p].view(np.uint32)) & own))
                assert d == 0, (name, mode, "Lr of pass %d" % p, d)
            assert ndiff(got["c"], ca_r) == 0, (name, mode, "cost")
            assert ndiff(got["o"], oa_r) == 0, (name, mode, "disparity")
            if "S" in got:
                d = int(np.sum((got["S"].view(np.uint32) != Sa.view(np.uint32)) & own))
                assert d == 0, (name, mode, "S", d)
        # the hull of the cost volume, last: asking for it expands a volume whose only copy is the range-proportional one
        Ch = cv.download()
        assert int(np.sum((Ch.view(np.uint32) != Ca.view(np.uint32)) & own)) == 0, name
        assert np.all(np.isposinf(Ch[~own])), name


def test_ragged_negative_penalty_takes_the_dense_kernels(oracle):
    """k_pass_rel keeps the launch's tag in the sign bits of what it hands over, so negative penalties (or weights)
    must not reach it: a ragged volume with P1 < 0 / a negative weight runs on the dense kernels and is still the reference's."""
    nx, ny, dmin, dmax = 90, 50, -70, 0
    u, v, gt = synth.stereo_pair(nx, ny, -50, 0, seed=5)
    dminI, dmaxI = window_ranges(gt, dmin, dmax, 9, 11, 3, 2)
    lo, hi = orc_mod.int_ranges(dminI, dmaxI)
    hmin, hmax = int(lo.min()), int(hi.max())
    Ca = oracle.costvolume_ranged(u, v, lo, hi, hmin, hmax, "none", "census", np.inf, 5)
    wneg = np.ones((8, ny, nx), np.float32)
    wneg[:, ::5, ::7] = -0.5
    with mgm_amd.Context(0) as ctx:
        cv = ctx.costvolume(u, v, dminI, dmaxI, "none", "census", float("inf"), 5)
        for (FH, P1, P2, w8h) in ((0, -2.0, 32.0, None), (1, -1.0, 30.0, None), (0, 8.0, 32.0, wneg)):
            w8 = ctx.upload_image(w8h) if w8h is not None else None
            got = run_hip(ctx, cv, "2", P1, P2, 8, 3, FH, 1, w8, None, lr=False)
            assert "k_pass_rel" not in got["names"], (FH, P1, got["names"])
            _, oa, ca = oracle.mgm_ranged(Ca, hmin, lo, hi, P1, P2, 8, 3, FH, 1, w8h, want_S=False)
            assert ndiff(got["c"], ca) == 0 and ndiff(got["o"], oa) == 0, (FH, P1, P2)
            # ... and the range-proportional kernels still work on the same context afterwards (no stale tag, no watchdog)
            got = run_hip(ctx, cv, "2", 8.0, 32.0, 8, 3, FH, 1, None, None, lr=False)
            assert "k_pass_rel" in got["names"]
            _, oa, ca = oracle.mgm_ranged(Ca, hmin, lo, hi, 8.0, 32.0, 8, 3, FH, 1, None, want_S=False)
            assert ndiff(got["c"], ca) == 0 and ndiff(got["o"], oa) == 0, (FH, "after")


def test_windowed_search_after_exact_path_does_not_read_stale_rel_volumes(oracle):
    """aggregate on the range-proportional copy, then the same volume with P2 = +INF (the operand-order-faithful
    kernel), then mgm_wta_windowed_dev -- which must search the LAST aggregation's Lr volumes, not the earlier one's."""
    nx, ny, dmin, dmax = 70, 40, -50, 0
    u, v, gt = synth.stereo_pair(nx, ny, -35, 0, seed=8)
    dminI, dmaxI = window_ranges(gt, dmin, dmax, 8, 8, 3, 1)
    lo, hi = orc_mod.int_ranges(dminI, dmaxI)
    hmin, hmax = int(lo.min()), int(hi.max())
    Ca = oracle.costvolume_ranged(u, v, lo, hi, hmin, hmax, "none", "census", np.inf, 5)
    with mgm_amd.Context(0) as ctx:
        cv = ctx.costvolume(u, v, dminI, dmaxI, "none", "census", float("inf"), 5)
        got = run_hip(ctx, cv, "2", 8.0, 32.0, 4, 3, 0, 1, None, None, lr=False)
        assert "k_pass_rel" in got["names"]
        got = run_hip(ctx, cv, "2", 8.0, float("inf"), 4, 3, 0, 1, None, None, lr=False)
        assert "k_pass_rel" not in got["names"]
        wlo, whi = ctx.upload_image(dminI[None]), ctx.upload_image(dmaxI[None])
        o2, c2 = ctx.wta_windowed_dev(cv, 4, 1, None, wlo, whi)
        _, oa, ca = oracle.mgm_ranged(Ca, hmin, lo, hi, 8.0, np.inf, 4, 3, 0, 1, None, want_S=False)
        fin = np.isfinite(ca)
        assert ndiff(c2.download()[0], ca) == 0
        assert np.array_equal(o2.download()[0][fin], oa[fin])


# ---- BASELINE's size: cfg3r / cfg3hr (bench.py: 1920x1080, hull of 256 labels, windows of +-24 around the true disparity) ----
FULL = [("cfg3r", 1, 2.0, 20000.0), ("cfg3hr", 0, 8.0, 32.0)]


def full_pair(seed):
    nx, ny, dmin, dmax = 1920, 1080, -255, 0
    u, v, gt = synth.stereo_pair(nx, ny, dmin * 3 // 4, 0, seed=seed)
    g = gt.astype(np.float32)
    return u, v, np.clip(g - 24, dmin, dmax), np.clip(g + 24, dmin, dmax)


@pytest.mark.parametrize("cfg", FULL, ids=lambda c: c[0])
def test_full_size_ragged_vs_oracle(oracle, cfg):
    """One 1920x1080 volume through k_cost_census_rel -> k_pass_rel -> k_wta_rel, and a batch of four in one launch: the Lr volume
    of the first and the last pass (one volume), labels, costs and the vfit refinement of every volume, bit for bit."""
    name, FH, P1, P2 = cfg
    NDIR, MGM, dmin, dmax = 8, 3, -255, 0
    nthreads = min(32, len(os.sched_getaffinity(0)))
    pairs = [full_pair(20150907 + 17 * k) for k in range(4)]
    with mgm_amd.Context(0) as ctx:
        cvs, refs = [], []
        for k, (u, v, dminI, dmaxI) in enumerate(pairs):
            du, dv = ctx.upload_image(u), ctx.upload_image(v)
            cvs.append(ctx.costvolume_ranged_dev(du, dv, ctx.upload_image(dminI[None]), ctx.upload_image(dmaxI[None]), dmin, dmax,
                                                 "none", "census", float("inf"), 5))
        for k, (u, v, dminI, dmaxI) in enumerate(pairs):
            lo, hi = orc_mod.int_ranges(dminI, dmaxI)
            oracle.set_threads(nthreads)
            try:
                Ca = oracle.costvolume_ranged(u, v, lo, hi, dmin, dmax, "none", "census", np.inf, 5)
                if k == 0:
                    Sa, oa, ca, lra = oracle.mgm_ranged(Ca, dmin, lo, hi, P1, P2, NDIR, MGM, FH, 1, None, dump_lr=(0, NDIR - 1))
                else:
                    Sa, oa, ca = oracle.mgm_ranged(Ca, dmin, lo, hi, P1, P2, NDIR, MGM, FH, 1, None)
                ra, rca = oracle.refine_ranged(Sa, dmin, lo, hi, "vfit", oa, ca)
            finally:
                oracle.set_threads(1)
            refs.append((ra, rca))
            if k == 0:
                got = run_hip(ctx, cvs[0], "1", P1, P2, NDIR, MGM, FH, 1, None, "vfit", lr=False)
                assert "k_pass_rel" in got["names"] and "k_expand" not in got["names"], got["names"]
                assert ndiff(got["c"], rca) == 0 and ndiff(got["o"], ra) == 0, (name, "x1")
                own = own_mask(lo, hi, dmin, dmax - dmin + 1)
                for n, p in enumerate((0, NDIR - 1)):
                    lr = ctx.debug_lr(cvs[0], p)
                    d = int(np.sum((lr.view(np.uint32) != lra[n].view(np.uint32)) & own))
                    assert d == 0, (name, "Lr of pass %d" % p, d)
                    del lr
                del lra, own
            del Ca, Sa
        ctx.timing(True)
        ctx.timing_reset()
        _, outs, outcs = ctx.aggregate_batch_dev(cvs, P1, P2, NDIR, MGM, FH, 1, None, "vfit")
        names = [n for n, _ in ctx.timings()]
        ctx.timing(False)
        assert "k_pass_rel" in names, names
        for k in range(4):
            assert ndiff(outcs[k].download()[0], refs[k][1]) == 0 and ndiff(outs[k].download()[0], refs[k][0]) == 0, (name, "x4", k)


def test_big_ragged_volume_rel_and_hull_vs_oracle(oracle):
    """tools/rel_bigcase.py as a test, against the oracle: 2048x1536 (a 96-band column chain, heights and widths beyond full HD),
    hull of 320 labels, windows of 41-61 labels with jitter -- range-proportional kernels and dense hull."""
    nx, ny, dmin, dmax = 2048, 1536, -319, 0
    u, v, gt = synth.stereo_pair(nx, ny, -240, 0, seed=77)
    dminI, dmaxI = window_ranges(gt, dmin, dmax, 24, 28, 5, 4)
    lo, hi = orc_mod.int_ranges(dminI, dmaxI)
    hmin, hmax = int(lo.min()), int(hi.max())
    oracle.set_threads(min(32, len(os.sched_getaffinity(0))))
    try:
        Ca = oracle.costvolume_ranged(u, v, lo, hi, hmin, hmax, "none", "census", np.inf, 5)
        Sa, oa, ca = oracle.mgm_ranged(Ca, hmin, lo, hi, 2.0, 20000.0, 8, 3, 1, 1, None)
        ra, rca = oracle.refine_ranged(Sa, hmin, lo, hi, "vfit", oa, ca)
    finally:
        oracle.set_threads(1)
    del Sa, Ca
    with mgm_amd.Context(0) as ctx:
        cv = ctx.costvolume(u, v, dminI, dmaxI, "none", "census", float("inf"), 5)
        for mode in ("1", "0"):
            got = run_hip(ctx, cv, mode, 2.0, 20000.0, 8, 3, 1, 1, None, "vfit", lr=False)
            assert ("k_pass_rel" in got["names"]) == (mode == "1"), got["names"]
            assert ndiff(got["c"], rca) == 0 and ndiff(got["o"], ra) == 0, mode


# ---- random cases over everything the widened range-proportional path takes, against the ORACLE -------------------------------
RFUZZ_N = int(os.environ.get("MGM_FUZZ_N", "0"))
RFUZZ_BASE = int(os.environ.get("MGM_FUZZ_BASE", "0"))


@pytest.mark.parametrize("seed", range(RFUZZ_BASE, RFUZZ_BASE + (RFUZZ_N or 32)))
def test_ragged_random_cases_vs_oracle(oracle, seed):
    """Random shapes, hulls, window widths up to 127 labels (64 or 128 slots per pixel; 127 = the dense hull), range images that are
    smooth, jumpy or nested, every cost form (one- / two-byte codes, fp32), every update function (TSGM 1..4, both potentials, with and
    without weights -- planes of ones included), refinements, over-count fix: range-proportional kernels AND dense hull against
    orc_mgm_ranged (labels, costs, the first and last pass's Lr).  MGM_FUZZ_N / MGM_FUZZ_BASE run campaigns."""
    rng = np.random.default_rng(424200 + seed)
    nx, ny = int(rng.integers(24, 110)), int(rng.integers(18, 70))
    cost, nch = [("census", 1), ("census", 1), ("ad", 1), ("ad", 3), ("sd", 1), ("ncc", 1), ("btad", 1), ("btsd", 3)][int(rng.integers(0, 8))]
    win = 5 if nch == 1 else 3
    dmin, dmax = -int(rng.integers(30, 220)), int(rng.integers(0, 12))
    kind = rng.choice(["win", "exact", "jumpy"])
    u, v, gt = synth.stereo_pair(nx, ny, dmin * 3 // 4, 0, seed=seed, nch=nch)
    if kind == "exact":
        width = int(min(rng.choice([5, 30, 61, 62, 63, 100, 126, 127]), dmax - dmin))
        dminI, dmaxI = exact_width_ranges(gt, dmin, dmax, width, seed)
    elif kind == "win":
        dminI, dmaxI = window_ranges(gt, dmin, dmax, int(rng.integers(1, 60)), int(rng.integers(1, 60)), seed, int(rng.integers(0, 5)))
    else:
        dminI, dmaxI = jumpy_ranges(nx, ny, dmin, dmax, int(min(rng.integers(2, 100), dmax - dmin)), seed)
    FH = int(rng.integers(0, 2))
    MGM = int(rng.integers(1, 5))
    NDIR = int(rng.choice([1, 2, 4, 8]))
    scale = {"census": 1.0, "ad": 3.0 * nch, "sd": 40.0, "ncc": 2.0, "btad": 2.0, "btsd": 60.0}[cost]
    P1 = float(rng.choice([0.75, 1.5, 2.0, 8.0])) * scale
    P2 = float(rng.choice([9.0, 32.0, 40.0, 20000.0])) * scale
    wkind = rng.choice(["none", "none", "three", "image", "ones"])
    refine = rng.choice([None, "vfit", "parabola", "cubic", "parabolaOCV"])
    fix = int(rng.integers(0, 2))
    lo, hi = orc_mod.int_ranges(dminI, dmaxI)
    hmin, hmax = int(lo.min()), int(hi.max())
    own = own_mask(lo, hi, hmin, hmax - hmin + 1)
    Ca = oracle.costvolume_ranged(u, v, lo, hi, hmin, hmax, "none", cost, np.inf, win)
    what = (seed, nx, ny, cost, nch, dmin, dmax, kind, FH, MGM, NDIR, P1, P2, wkind, refine, fix, int((hi - lo).max()) + 1)
    with mgm_amd.Context(0) as ctx:
        os.environ["MGM_HIP_REL"] = "2"
        cv = ctx.costvolume(u, v, dminI, dmaxI, "none", cost, float("inf"), win)
        os.environ.pop("MGM_HIP_REL")
        w8 = w8h = None
        if wkind == "three":
            w8h = rng.choice(np.array([1.0, 2.5, 4.0], np.float32), size=(8, ny, nx), p=[0.6, 0.25, 0.15])
        elif wkind == "image":
            w8h = oracle.weights(u, 4.0, 12.0)
        elif wkind == "ones":
            w8h = np.ones((8, ny, nx), np.float32)
        if w8h is not None:
            w8 = ctx.upload_image(w8h)
        oracle.set_threads(min(8, len(os.sched_getaffinity(0))))
        try:
            Sa, oa, ca, lra = oracle.mgm_ranged(Ca, hmin, lo, hi, P1, P2, NDIR, MGM, FH, fix, w8h, dump_lr=(0, NDIR - 1))
        finally:
            oracle.set_threads(1)
        oa_r, ca_r = oracle.refine_ranged(Sa, hmin, lo, hi, refine, oa, ca) if refine else (oa, ca)
        for mode in ("2", "0"):
            got = run_hip(ctx, cv, mode, P1, P2, NDIR, MGM, FH, fix, w8, refine, lr=False)
            ran_rel = "k_pass_rel" in got["names"]
            is_ragged = bool((lo != lo.flat[0]).any() or (hi != hi.flat[0]).any())
            if mode == "2":
                width = int((hi - lo).max()) + 1
                # (the one combination that keeps the dense hull: update_cost2_trunclinear on 128 slots of fp32 costs -- its LDS rings do not fit)
                lds = FH and MGM == 2 and wkind in ("none", "ones") and cost in ("ncc", "btad", "btsd") and width > 62
                assert ran_rel == (is_ragged and width <= 126 and not lds), (what, got["names"])
            assert ndiff(got["c"], ca_r) == 0 and ndiff(got["o"], oa_r) == 0, (what, mode, "maps")
            for n, p in enumerate(sorted({0, NDIR - 1})):
                lr = ctx.debug_lr(cv, p)
                d = int(np.sum((lr.view(np.uint32) != lra[min(n, len(lra) - 1)].view(np.uint32)) & own))
                assert d == 0, (what, mode, "Lr of pass %d" % p, d)


def test_windowed_search_and_refill_across_formats(oracle):
    """main()'s TSGM_ITER loop on a ragged volume of WIDE windows (128 slots per pixel): mgm_wta_windowed_dev searches the
    range-proportional Lr volumes again in narrowed / shifted windows (S allocated from those, mgm_core.cc:426) -- against
    orc_mgm_ranged with separate S ranges; then the SAME volume handle is refilled with narrow windows (64 slots), then wide
    ones again: the copy's format follows, results stay the oracle's."""
    nx, ny, dmin, dmax = 88, 47, -180, 10
    u, v, gt = synth.stereo_pair(nx, ny, -130, 0, seed=12)
    P1, P2, NDIR, MGM, FH = 2.0, 40.0, 8, 3, 1
    with mgm_amd.Context(0) as ctx:
        du, dv = ctx.upload_image(u), ctx.upload_image(v)
        cv = None
        for rep, width in enumerate((101, 31, 118)):
            dminI, dmaxI = exact_width_ranges(gt, dmin, dmax, width, 40 + rep)
            lo, hi = orc_mod.int_ranges(dminI, dmaxI)
            hmin, hmax = dmin, dmax  # (the device form names its hull: the same for every refill of the handle)
            dlo, dhi = ctx.upload_image(dminI[None]), ctx.upload_image(dmaxI[None])
            cv = ctx.costvolume_ranged_dev(du, dv, dlo, dhi, hmin, hmax, "none", "census", float("inf"), 5, into=cv)
            Ca = oracle.costvolume_ranged(u, v, lo, hi, hmin, hmax, "none", "census", np.inf, 5)
            got = run_hip(ctx, cv, "1", P1, P2, NDIR, MGM, FH, 1, None, "vfit", lr=False)
            assert "k_pass_rel" in got["names"], got["names"]
            Sa, oa, ca = oracle.mgm_ranged(Ca, hmin, lo, hi, P1, P2, NDIR, MGM, FH, 1, None)
            ra, rca = oracle.refine_ranged(Sa, hmin, lo, hi, "vfit", oa, ca)
            assert ndiff(got["c"], rca) == 0 and ndiff(got["o"], ra) == 0, (width, "aggregate")
            # a second search in windows narrowed around the solution and pushed partly outside the pixel's own range / the hull
            rng = np.random.default_rng(rep)
            wl = np.clip(np.where(np.isfinite(oa), oa, lo) - rng.integers(1, 9, size=oa.shape), hmin - 5, hmax).astype(np.float32)
            wh = np.clip(wl + rng.integers(3, 20, size=oa.shape), wl + 1, hmax + 6).astype(np.float32)
            slo, shi = orc_mod.int_ranges(wl, wh)
            shmin, shmax = int(slo.min()), int(shi.max())
            for fix, refine in ((1, "vfit"), (0, "cubic")):
                if fix == 0:  # (the Lr volumes are those of the last aggregation: run it again without the fix)
                    run_hip(ctx, cv, "1", P1, P2, NDIR, MGM, FH, 0, None, None, lr=False)
                o2, c2 = ctx.wta_windowed_dev(cv, NDIR, fix, refine, ctx.upload_image(wl[None]), ctx.upload_image(wh[None]))
                S2, oo, cc = oracle.mgm_ranged(Ca, hmin, lo, hi, P1, P2, NDIR, MGM, FH, fix, None, (slo, shi, shmin, shmax))
                r2, rc2 = oracle.refine_ranged(S2, shmin, slo, shi, refine, oo, cc)
                assert ndiff(c2.download()[0], rc2) == 0 and ndiff(o2.download()[0], r2) == 0, (width, "windowed", fix, refine)
