"""Two-valued edge weights (k_pass2, W2): the planes compute_mgm_weights makes (mgm_weights.h:63-85) hold 1 and ONE other value,
so the producer publishes both transforms and every reader picks per neighbour -- compact costs, self-validating slabs, deep
rings and per-XCD queues instead of the general weighted kernels.  Against the oracle's update_costW / update_costW_trunclinear
(mgm_core.cc:95-144, 229-281) on images large enough for the queues, every TSGM, both potentials, weights above and below 1,
image-driven and random planes, batches; and the cases that must fall back to the general kernels (three values, a weight of
0, fp32 costs)."""
import numpy as np
import pytest

import mgm_amd
from helpers import ndiff
from mgm_amd import synth

pytestmark = pytest.mark.gpu


def big_threads(oracle):
    from oracle.oracle import usable_cpus
    oracle.set_threads(min(16, usable_cpus()))


def check(ctx, oracle, C, dmin, w, P1, P2, NDIR, MGM, FH):
    cv = ctx.upload_volume(C, dmin)
    S, o, k = ctx.aggregate(cv, P1, P2, NDIR, MGM, FH, 1, w, None, want_S=True)  # (host weights: uploaded by the wrapper)
    Sa, oa, ca = oracle.mgm(C, dmin, P1, P2, NDIR, MGM, FH, 1, w)
    bad = ndiff(S.download(), Sa), ndiff(k, ca), ndiff(o, oa)
    for h in (cv, S):
        h.free()
    return bad


@pytest.mark.parametrize("a", [4.0, 0.3])
@pytest.mark.parametrize("FH,P1,P2", [(0, 8.0, 32.0), (1, 2.0, 20000.0), (1, 1.5, 9.0)])
@pytest.mark.parametrize("MGM", [1, 2, 3, 4])
@pytest.mark.parametrize("L", [64, 128, 192, 256])
def test_two_valued_weights_vs_oracle(ctx, oracle, L, MGM, FH, P1, P2, a):
    nx, ny = 333, 241  # 17 / 23 bands per pass: the launch takes the queues
    rng = np.random.default_rng(L * 100 + MGM * 10 + FH)
    C = np.rint(synth.raw_volume(nx, ny, L, seed=L + MGM, maxcost=60, inf_frac=0.02)).astype(np.float32)
    w = np.where(rng.random((8, ny, nx)) < 0.45, np.float32(a), np.float32(1.0)).astype(np.float32)
    big_threads(oracle)
    try:
        assert check(ctx, oracle, C, -L // 2, w, P1, P2, 8, MGM, FH) == (0, 0, 0)
    finally:
        oracle.set_threads(1)


def test_image_driven_weights_end_to_end(ctx, oracle):
    """The reference's own chain: compute_mgm_weights on the left image (K7), census costs (K1, K2), weighted FH aggregation."""
    nx, ny, dmin, dmax = 480, 270, -127, 0
    u, v, _ = synth.stereo_pair(nx, ny, -90, 0, seed=77)
    du, dv = ctx.upload_image(u), ctx.upload_image(v)
    cv = ctx.costvolume_dev(du, dv, dmin, dmax, "none", "census", float("inf"), 5)
    dw = ctx.weights_dev(du, 4.0, 12.0)
    big_threads(oracle)
    try:
        for FH, P1, P2 in ((1, 2.0, 20000.0), (0, 8.0, 32.0)):
            _, o, k = ctx.aggregate_dev(cv, P1, P2, 8, 3, FH, 1, dw, "vfit")
            Ca = oracle.costvolume(u, v, dmin, dmax, "none", "census", np.inf, 5)
            wa = oracle.weights(u, 4.0, 12.0)
            assert ndiff(dw.download(), wa) == 0
            Sa, oa, ca = oracle.mgm(Ca, dmin, P1, P2, 8, 3, FH, 1, wa)
            ra, rca = oracle.refine(Sa, dmin, "vfit", oa, ca)
            assert ndiff(k.download()[0], rca) == 0 and ndiff(o.download()[0], ra) == 0, FH
    finally:
        oracle.set_threads(1)


def test_batches_with_different_second_values(ctx, oracle):
    """Every volume of a batch brings its own second weight value (P1*a, P2*a travel per volume)."""
    nx, ny, L = 320, 200, 128
    Cs, ws, cvs, dws = [], [], [], []
    for b, a in enumerate([4.0, 0.5, 2.0]):
        rng = np.random.default_rng(40 + b)
        Cs.append(np.rint(synth.raw_volume(nx, ny, L, seed=400 + b, maxcost=40)).astype(np.float32))
        ws.append(np.where(rng.random((8, ny, nx)) < 0.5, np.float32(a), np.float32(1.0)).astype(np.float32))
        cvs.append(ctx.upload_volume(Cs[-1], -64))
        dws.append(ctx.upload_image(ws[-1]))
    _, outs, outcs = ctx.aggregate_batch_dev(cvs, 2.0, 20000.0, 8, 3, 1, 1, dws, None)
    big_threads(oracle)
    try:
        for b in range(3):
            Sa, oa, ca = oracle.mgm(Cs[b], -64, 2.0, 20000.0, 8, 3, 1, 1, ws[b])
            assert ndiff(outcs[b].download()[0], ca) == 0 and ndiff(outs[b].download()[0], oa) == 0, b
    finally:
        oracle.set_threads(1)


@pytest.mark.parametrize("kind", ["three values", "a zero weight", "fp32 costs"])
def test_what_the_two_valued_kernels_do_not_cover_still_works(ctx, oracle, kind):
    nx, ny, L = 333, 241, 128
    rng = np.random.default_rng(5)
    C = np.rint(synth.raw_volume(nx, ny, L, seed=9, maxcost=60)).astype(np.float32)
    w = np.where(rng.random((8, ny, nx)) < 0.45, np.float32(4.0), np.float32(1.0)).astype(np.float32)
    if kind == "three values":
        w[rng.random((8, ny, nx)) < 0.1] = np.float32(2.5)
    elif kind == "a zero weight":
        w[w == 4.0] = 0.0
    else:
        C = C + np.float32(0.5)  # no compact form
    big_threads(oracle)
    try:
        assert check(ctx, oracle, C, -64, w, 8.0, 32.0, 8, 3, 0) == (0, 0, 0)
    finally:
        oracle.set_threads(1)


@pytest.mark.parametrize("FH,P1,P2", [(0, 8.0, 32.0), (1, 2.0, 20000.0)])
@pytest.mark.parametrize("MGM", [1, 3, 4])
def test_batch_with_planes_of_ones_beside_real_weights(ctx, oracle, MGM, FH, P1, P2):
    """One launch, four volumes, two of them with weight planes that are all ones: with TSGM != 2 the reference's update is
    the same function with DeltaI = 1.0 (mgm_core.cc:563-575), so the launch runs weighted for all -- two-valued kernels
    included -- and every volume gets the result of its own call."""
    nx, ny, L = 333, 120, 128
    rng = np.random.default_rng(MGM * 10 + FH)
    Cs = [np.rint(synth.raw_volume(nx, ny, L, seed=400 + b, maxcost=60, inf_frac=0.02)).astype(np.float32) for b in range(4)]
    ws = [np.ones((8, ny, nx), np.float32) if b % 2 else np.where(rng.random((8, ny, nx)) < 0.45, np.float32(4.0), np.float32(1.0)).astype(np.float32)
          for b in range(4)]
    cvs = [ctx.upload_volume(C, -L // 2) for C in Cs]
    dws = [ctx.upload_image(w) for w in ws]
    S, outs, outcs = ctx.aggregate_batch_dev(cvs, P1, P2, 8, MGM, FH, 1, dws, None, want_S=True)
    big_threads(oracle)
    try:
        for b in range(4):
            Sa, oa, ca = oracle.mgm(Cs[b], -L // 2, P1, P2, 8, MGM, FH, 1, ws[b])
            assert (ndiff(S[b].download(), Sa), ndiff(outcs[b].download()[0], ca), ndiff(outs[b].download()[0], oa)) == (0, 0, 0), b
    finally:
        oracle.set_threads(1)
    for h in cvs + dws + list(S) + outs + outcs:
        h.free()
