"""Full-size checks at BASELINE.json's shapes (1920x1080, 128 / 256 labels): size-independent properties, oracle
checks on the parts of the result that only depend on a crop of the input (single-threaded oracle), and the two
bench workloads whole against the OpenMP oracle."""
import os

import numpy as np
import pytest

from helpers import labels_equal, ndiff
from mgm_amd import synth

pytestmark = pytest.mark.gpu

W, H = 1920, 1080


@pytest.fixture(scope="module")
def vol256():
    return synth.raw_volume(W, H, 256, seed=3)


def test_pass0_rows_depend_only_on_rows_above(ctx, oracle, vol256):
    """Pass 0 (mgm_core.cc:463) scans top->bottom: the first K rows of its result equal the
    result on the volume cropped to K rows.  The crop is small enough for the oracle."""
    K = 20
    cv = ctx.upload_volume(vol256, 0)
    for (MGM, FH, P1, P2) in ((3, 0, 8.0, 32.0), (3, 1, 2.0, 20000.0), (4, 0, 8.0, 32.0)):
        _, o, c = ctx.aggregate(cv, P1, P2, 1, MGM, FH, 1, None, None, want_S=False)
        lr = ctx.debug_lr(cv, 0)
        Sa, oa, ca, lra = oracle.mgm(vol256[:K + 1], 0, P1, P2, 1, MGM, FH, 1, None, dump_lr=True)
        assert ndiff(lr[:K], lra[0][:K]) == 0, (MGM, FH)
        assert ndiff(c[:K], ca[:K]) == 0 and labels_equal(o[:K], oa[:K], ca[:K])
    cv.free()


def test_point_symmetry_with_two_directions(ctx, vol256):
    """Passes 0 and 1 are images of each other under a 180-degree rotation, and a two-term
    fp32 sum commutes, so aggregate(rot180(C), NDIR=2) == rot180(aggregate(C, NDIR=2))."""
    C = vol256[:, :, :128].copy()
    Cr = np.ascontiguousarray(C[::-1, ::-1, :])
    cv, cvr = ctx.upload_volume(C, 0), ctx.upload_volume(Cr, 0)
    for (MGM, FH, P1, P2) in ((2, 0, 8.0, 32.0), (3, 1, 2.0, 20000.0)):
        S, o, c = ctx.aggregate(cv, P1, P2, 2, MGM, FH, 1, None, None, want_S=True)
        Sr, orr, cr = ctx.aggregate(cvr, P1, P2, 2, MGM, FH, 1, None, None, want_S=True)
        assert ndiff(S.download(), Sr.download()[::-1, ::-1, :]) == 0
        assert ndiff(c, cr[::-1, ::-1]) == 0 and np.array_equal(o, orr[::-1, ::-1])
        S.free(), Sr.free()
    cv.free(), cvr.free()


@pytest.mark.parametrize("cfg", [("cfg2", 128, 4, 2, 0, 8.0, 32.0), ("cfg3-hirschmueller", 256, 8, 3, 0, 8.0, 32.0),
                                 ("cfg3", 256, 8, 3, 1, 2.0, 20000.0)], ids=lambda c: c[0])
def test_repeatable_and_builds_agree_at_full_size(cfg, vol256):
    """Two runs are bit-identical (no race in the inter-band hand-off shows up as a changed
    bit), and the two independent builds of K3 give the same S, labels and costs."""
    import os
    import mgm_amd
    _, L, NDIR, MGM, FH, P1, P2 = cfg
    C = vol256[:, :, :L].copy()
    runs = []
    for build in ("0", "0", "1"):
        os.environ["MGM_HIP_PASS_BUILD"] = build
        c = mgm_amd.Context(0)
        cv = c.upload_volume(C, 0)
        S, o, oc = c.aggregate(cv, P1, P2, NDIR, MGM, FH, 1, None, "vfit", want_S=True)
        runs.append((S.download(), o, oc))
        c.close()
    os.environ.pop("MGM_HIP_PASS_BUILD")
    for r in runs[1:]:
        assert ndiff(runs[0][0], r[0]) == 0 and ndiff(runs[0][1], r[1]) == 0 and ndiff(runs[0][2], r[2]) == 0
    # WTA consistency: the reported cost is the minimum of S over finite entries, at the reported label
    S, o, oc = runs[0]
    ref_min = np.where(np.isfinite(S), S, np.inf).min(axis=2)
    ref_arg = np.where(np.isfinite(S), S, np.inf).argmin(axis=2)
    unrefined_cost = np.take_along_axis(S, ref_arg[..., None], axis=2)[..., 0]
    assert np.array_equal(ref_min, unrefined_cost)
    assert np.all(np.abs(o - ref_arg) <= 1.0 + 1e-6)  # vfit moves the label by at most one


def test_quarter_size_end_to_end_vs_oracle(ctx, oracle):
    """Images -> census 5x5 cost volume -> 8-direction TSGM=3 (both potentials) -> WTA -> vfit,
    480x270x256: the largest size the scalar oracle finishes in a few seconds."""
    nx, ny, dmin, dmax = 480, 270, -255, 0
    u, v, _ = synth.stereo_pair(nx, ny, -200, 0)
    Ca = oracle.costvolume(u, v, dmin, dmax, "none", "census", np.inf, 5)
    du, dv = ctx.upload_image(u), ctx.upload_image(v)
    cv = ctx.costvolume_dev(du, dv, dmin, dmax, "none", "census", float("inf"), 5)
    assert ndiff(Ca, cv.download()) == 0
    oracle.set_threads(8)
    for (FH, P1, P2) in ((0, 8.0, 32.0), (1, 2.0, 20000.0)):
        Sa, oa, ca = oracle.mgm(Ca, dmin, P1, P2, 8, 3, FH, 1)
        ra, rca = oracle.refine(Sa, dmin, "vfit", oa, ca)
        S, o, c = ctx.aggregate(cv, P1, P2, 8, 3, FH, 1, None, "vfit", want_S=True)
        assert ndiff(Sa, S.download()) == 0
        assert ndiff(ra, o) == 0 and ndiff(rca, c) == 0
        S.free()
    oracle.set_threads(1)
    for h in (cv, du, dv):
        h.free()


def test_full_size_batch_equals_single(ctx, vol256):
    """Four full-size volumes in one pass launch (two bands per CU, the occupancy bench.py runs at) give each volume
    exactly the result of its own launch (one band per CU, chain-bound)."""
    P1, P2, NDIR, MGM, FH = 2.0, 20000.0, 8, 3, 1
    rng = np.random.default_rng(17)
    vols = [vol256] + [np.roll(vol256, int(rng.integers(1, 200)), axis=k % 3) for k in range(3)]
    cvs = [ctx.upload_volume(v, -255) for v in vols]
    ref = []
    for cv in cvs:
        _, o, c = ctx.aggregate_dev(cv, P1, P2, NDIR, MGM, FH, 1, None, "vfit")
        ref.append((o.download(), c.download()))
    _, outs, outcs = ctx.aggregate_batch_dev(cvs, P1, P2, NDIR, MGM, FH, 1, None, "vfit")
    for b in range(len(cvs)):
        assert ndiff(outs[b].download(), ref[b][0]) == 0 and ndiff(outcs[b].download(), ref[b][1]) == 0, b
    for cv in cvs:
        cv.free()


@pytest.mark.parametrize("cfg", [("cfg3", -255, 0, 5, 8, 3, 1, 2.0, 20000.0), ("cfg2", -127, 0, 3, 4, 2, 0, 8.0, 32.0)], ids=lambda c: c[0])
def test_full_size_workloads_vs_oracle(ctx, oracle, cfg):
    """The bench workloads at their full size, 1920x1080, against the OpenMP oracle (every host core the box
    gives the process): corrected S, labels, costs and the vfit refinement, bit for bit."""
    _, dmin, dmax, win, NDIR, MGM, FH, P1, P2 = cfg
    nx, ny = 1920, 1080
    u, v, _ = synth.stereo_pair(nx, ny, dmin * 3 // 4, 0, seed=20150907)
    du, dv = ctx.upload_image(u), ctx.upload_image(v)
    cv = ctx.costvolume_dev(du, dv, dmin, dmax, "none", "census", float("inf"), win)
    C = cv.download()
    oracle.set_threads(min(32, len(os.sched_getaffinity(0))))
    try:
        Sa, oa, ca = oracle.mgm(C, dmin, P1, P2, NDIR, MGM, FH, 1)
        ra, rca = oracle.refine(Sa, dmin, "vfit", oa, ca)
    finally:
        oracle.set_threads(1)
    S, o, c = ctx.aggregate(cv, P1, P2, NDIR, MGM, FH, 1, None, "vfit", want_S=True)
    assert ndiff(rca, c) == 0 and ndiff(ra, o) == 0
    assert ndiff(Sa, S.download()) == 0
    for h in (S, cv, du, dv):
        h.free()
