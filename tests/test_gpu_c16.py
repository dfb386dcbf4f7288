"""Two-byte compact costs (round 4): absolute differences of colour pairs (up to 765) and squared differences (up to 65025) are
integers that do not fit the one-byte form; K2 writes them as 16-bit words next to the fp32 volume, the unweighted pass kernels
with deep rings and k_wta read those (half the cost traffic of fp32).  Everything against the oracle, bit for bit: the cost
volume, S, labels and refined maps -- at label counts of every kernel width, a count that runs padded (BASELINE config 1's 151),
with the combinations that must fall back to the fp32 volume (weights, FH with TSGM 2) and with costs that overflow the form."""
import numpy as np
import pytest

from helpers import ndiff
from mgm_amd import synth

pytestmark = pytest.mark.gpu


def threads(oracle, n=16):
    from oracle.oracle import usable_cpus
    oracle.set_threads(min(n, usable_cpus()))


def run_case(ctx, oracle, nx, ny, dmin, dmax, cost, nch, NDIR, MGM, FH, P1, P2, w=None, scale=1.0, trunc=np.inf):
    u, v, _ = synth.stereo_pair(nx, ny, dmin * 3 // 4, max(0, dmax * 3 // 4), seed=dmax - dmin + MGM, nch=nch)
    u, v = (u * np.float32(scale)).astype(np.float32), (v * np.float32(scale)).astype(np.float32)
    du, dv = ctx.upload_image(u), ctx.upload_image(v)
    cv = ctx.costvolume_dev(du, dv, dmin, dmax, "none", cost, float(trunc), 3)
    dw = ctx.upload_image(w) if w is not None else None
    S, o, k = ctx.aggregate_dev(cv, P1, P2, NDIR, MGM, FH, 1, dw, "vfit", want_S=True)
    threads(oracle)
    try:
        Ca = oracle.costvolume(u, v, dmin, dmax, "none", cost, trunc, 3)
        Sa, oa, ca = oracle.mgm(Ca, dmin, P1, P2, NDIR, MGM, FH, 1, w)
        ra, rca = oracle.refine(Sa, dmin, "vfit", oa, ca)
    finally:
        oracle.set_threads(1)
    bad = (ndiff(cv.download(), Ca), ndiff(S.download(), Sa), ndiff(k.download()[0], rca), ndiff(o.download()[0], ra))
    for h in (du, dv, cv, S, o, k) + ((dw,) if dw is not None else ()):
        h.free()
    return bad


@pytest.mark.parametrize("L", [64, 128, 192, 256, 384, 512])
@pytest.mark.parametrize("mode", [(8, 3, 0, 24.0, 96.0), (8, 3, 1, 6.0, 60000.0), (4, 2, 0, 24.0, 96.0), (8, 4, 1, 4.5, 40.0), (8, 1, 1, 6.0, 300.0)],
                         ids=["O8-T3", "O8-T3-FH", "O4-T2", "O8-T4-FH", "O8-T1-FH"])
def test_colour_absolute_differences(ctx, oracle, L, mode):
    NDIR, MGM, FH, P1, P2 = mode
    assert run_case(ctx, oracle, 333, 241, -(L - 1) + L // 4, L // 4, "ad", 3, NDIR, MGM, FH, P1, P2) == (0, 0, 0, 0)


@pytest.mark.parametrize("cost,nch,scale", [("sd", 1, 1.0), ("sd", 3, 0.5), ("ad", 3, 100.0), ("sd", 1, 2.0)],
                         ids=["sd-grey", "sd-colour-small", "ad-overflows-two-bytes", "sd-overflows"])
def test_squared_differences_and_overflow(ctx, oracle, cost, nch, scale):
    """Squared differences of one channel fit two bytes; costs that do not (values beyond 65534: the flag word says so) are
    aggregated from the fp32 volume."""
    assert run_case(ctx, oracle, 300, 200, -100, 27, cost, nch, 8, 3, 1, 6.0, 60000.0, scale=scale) == (0, 0, 0, 0)


def test_config1_shape_runs_padded_with_two_byte_costs(ctx, oracle):
    """BASELINE config 1's shape: 700x500 RGB, -r -120 -R 30 (151 labels -> 192), -t ad, -O 4, TSGM 2; also truncated costs."""
    assert run_case(ctx, oracle, 700, 500, -120, 30, "ad", 3, 4, 2, 0, 24.0, 96.0) == (0, 0, 0, 0)
    assert run_case(ctx, oracle, 350, 250, -120, 30, "ad", 3, 8, 3, 1, 6.0, 60000.0, trunc=40.0) == (0, 0, 0, 0)


@pytest.mark.parametrize("kind", ["weights", "FH-TSGM2"])
def test_combinations_that_read_the_fp32_volume(ctx, oracle, kind):
    nx, ny = 333, 241
    if kind == "weights":
        rng = np.random.default_rng(3)
        w = np.where(rng.random((8, ny, nx)) < 0.4, np.float32(4.0), np.float32(1.0)).astype(np.float32)
        assert run_case(ctx, oracle, nx, ny, -100, 27, "ad", 3, 8, 3, 1, 6.0, 60000.0, w=w) == (0, 0, 0, 0)
    else:
        assert run_case(ctx, oracle, nx, ny, -100, 27, "ad", 3, 4, 2, 1, 6.0, 60.0) == (0, 0, 0, 0)


def test_batch_of_colour_pairs(ctx, oracle):
    nx, ny, dmin, dmax = 320, 200, -127, 0
    cvs, Cas = [], []
    for b in range(4):
        u, v, _ = synth.stereo_pair(nx, ny, -90, 0, seed=60 + b, nch=3)
        du, dv = ctx.upload_image(u), ctx.upload_image(v)
        cvs.append(ctx.costvolume_dev(du, dv, dmin, dmax, "none", "ad", float("inf"), 3))
        Cas.append(oracle.costvolume(u, v, dmin, dmax, "none", "ad", np.inf, 3))
    _, outs, outcs = ctx.aggregate_batch_dev(cvs, 24.0, 96.0, 8, 3, 0, 1, None, None)
    threads(oracle)
    try:
        for b in range(4):
            Sa, oa, ca = oracle.mgm(Cas[b], dmin, 24.0, 96.0, 8, 3, 0, 1)
            assert ndiff(outcs[b].download()[0], ca) == 0 and ndiff(outs[b].download()[0], oa) == 0, b
    finally:
        oracle.set_threads(1)


# ---- compact-only filling (round 4): k_cost_diffx writes the compact copy alone; the fp32 volume only on demand --------------
def _fill_case(ctx, oracle, nx, ny, vnx, vny, dmin, dmax, pre, cost, nch, trunc, scale=1.0, seed=5):
    u, _, _ = synth.stereo_pair(nx, ny, dmin * 3 // 4, max(0, dmax * 3 // 4), seed=seed, nch=nch)
    _, v, _ = synth.stereo_pair(vnx, vny, dmin * 3 // 4, max(0, dmax * 3 // 4), seed=seed, nch=nch)
    u, v = (u * np.float32(scale)).astype(np.float32), (v * np.float32(scale)).astype(np.float32)
    return u, v, oracle.costvolume(u, v, dmin, dmax, pre, cost, trunc, 3)


@pytest.mark.parametrize("L", [64, 128, 192, 256, 384, 512, 768, 1024])
@pytest.mark.parametrize("kind", ["ad-grey", "ad-colour", "sd-grey", "ad-colour-trunc", "ad-fractional-trunc", "ad-sobelx", "ad-float-images"])
def test_compact_only_fill(ctx, oracle, L, kind):
    """k_cost_diffx; the volume read back (expanded from the compact copy, or written by the
    general kernel where a cost did not fit: fractional truncDist, float-valued images, colour at one byte per cost) equals the
    oracle's bit for bit, also where the right image is narrower and shorter than the left."""
    cost = "sd" if kind.startswith("sd") else "ad"
    nch = 3 if "colour" in kind else 1
    trunc = 40.0 if kind == "ad-colour-trunc" else (12.5 if kind == "ad-fractional-trunc" else np.inf)
    pre = "sobelx" if kind == "ad-sobelx" else "none"
    scale = 0.37 if kind == "ad-float-images" else 1.0
    nx, ny = 96, 14
    dmin = -(L - 1) + L // 4
    threads(oracle)
    try:
        for vnx, vny in ((nx, ny), (nx - 7, ny - 3), (nx + 9, ny)):
            u, v, Ca = _fill_case(ctx, oracle, nx, ny, vnx, vny, dmin, dmin + L - 1, pre, cost, nch, trunc, scale)
            du, dv = ctx.upload_image(u), ctx.upload_image(v)
            cv = ctx.costvolume_dev(du, dv, dmin, dmin + L - 1, pre, cost, float(trunc), 3)
            assert ndiff(cv.download(), Ca) == 0, (vnx, vny)
            for h in (du, dv, cv):
                h.free()
    finally:
        oracle.set_threads(1)


def test_compact_only_fill_refills_and_aggregates(ctx, oracle):
    """One volume refilled in turn with integer-valued, float-valued and again integer-valued pairs (the float-valued filling
    sends the later ones to the general kernel), aggregated each time; and truncDist = NaN (every cost NaN)."""
    nx, ny, dmin, dmax = 128, 40, -127, 0
    cv = None
    threads(oracle)
    try:
        for k, scale in enumerate((1.0, 0.37, 1.0)):
            u, v, Ca = _fill_case(ctx, oracle, nx, ny, nx, ny, dmin, dmax, "none", "ad", 3, np.inf, scale, seed=11 + k)
            du, dv = ctx.upload_image(u), ctx.upload_image(v)
            cv = ctx.costvolume_dev(du, dv, dmin, dmax, "none", "ad", float("inf"), 3, into=cv)
            S, o, kk = ctx.aggregate_dev(cv, 24.0, 96.0, 8, 3, 0, 1, None, "vfit", want_S=True)
            Sa, oa, ca = oracle.mgm(Ca, dmin, 24.0, 96.0, 8, 3, 0, 1)
            ra, rca = oracle.refine(Sa, dmin, "vfit", oa, ca)
            assert (ndiff(cv.download(), Ca), ndiff(S.download(), Sa), ndiff(o.download()[0], ra), ndiff(kk.download()[0], rca)) == (0, 0, 0, 0), k
            for h in (du, dv, S, o, kk):
                h.free()
        u, v, Ca = _fill_case(ctx, oracle, nx, ny, nx, ny, dmin, dmax, "none", "ad", 1, np.nan)
        du, dv = ctx.upload_image(u), ctx.upload_image(v)
        cv2 = ctx.costvolume_dev(du, dv, dmin, dmax, "none", "ad", float("nan"), 3)
        assert ndiff(cv2.download(), Ca) == 0
        for h in (du, dv, cv, cv2):
            h.free()
    finally:
        oracle.set_threads(1)


@pytest.mark.parametrize("L", [7, 100, 151, 300, 600, 1000])
@pytest.mark.parametrize("kind", ["census", "census-trunc", "ad-grey", "ad-colour", "sd-grey", "ad-float-images"])
def test_padded_compact_only_fill(ctx, oracle, L, kind):
    """Label counts that the pass kernels run padded (7 -> 64 ... 1000 -> 1024): K2 writes the padded compact copy itself
    (mgm_cv::p8), the aggregation reads it as it is, the fp32 volume exists only once somebody asks for it.  Aggregated
    first, read back second -- and the other way round on a second volume -- against the oracle; widths that take the
    four-pixel kernels and one that does not; a right image narrower and shorter than the left."""
    cost = "census" if kind.startswith("census") else ("sd" if kind.startswith("sd") else "ad")
    nch = 3 if "colour" in kind else 1
    trunc = 7.0 if kind == "census-trunc" else np.inf
    scale = 0.37 if kind == "ad-float-images" else 1.0
    dmin = -(L - 1) + L // 4
    threads(oracle)
    try:
        for k, (nx, ny, vnx, vny) in enumerate(((96, 14, 96, 14), (61, 9, 61, 9), (96, 14, 89, 11))):
            u, v, Ca = _fill_case(ctx, oracle, nx, ny, vnx, vny, dmin, dmin + L - 1, "none", cost, nch, trunc, scale, seed=30 + k)
            du, dv = ctx.upload_image(u), ctx.upload_image(v)
            cv = ctx.costvolume_dev(du, dv, dmin, dmin + L - 1, "none", cost, float(trunc), 3)
            if k == 1:
                assert ndiff(cv.download(), Ca) == 0, (nx, vnx)
            FH, P1, P2 = ((0, 8.0, 32.0), (1, 2.0, 20000.0), (1, 1.5, 9.0))[k]
            S, o, kk = ctx.aggregate_dev(cv, P1, P2, 8, 3, FH, 1, None, "vfit", want_S=True)
            Sa, oa, ca = oracle.mgm(Ca, dmin, P1, P2, 8, 3, FH, 1)
            ra, rca = oracle.refine(Sa, dmin, "vfit", oa, ca)
            assert (ndiff(S.download(), Sa), ndiff(o.download()[0], ra), ndiff(kk.download()[0], rca)) == (0, 0, 0), (nx, vnx, FH)
            assert ndiff(cv.download(), Ca) == 0, (nx, vnx)
            for h in (du, dv, cv, S, o, kk):
                h.free()
    finally:
        oracle.set_threads(1)


@pytest.mark.parametrize("L", [128, 151])
def test_grey_pair_with_a_difference_of_255_takes_two_bytes(ctx, oracle, L):
    """One byte per cost ends at 254: a grey pair that holds a 0 opposite a 255 is filled again with two bytes per cost (not with
    fp32 costs), also in a padded layout, and so are the refills of that volume."""
    nx, ny, dmin = 96, 20, -(L - 1)
    threads(oracle)
    try:
        cv = None
        for k in range(2):
            u, v, _ = synth.stereo_pair(nx, ny, dmin * 3 // 4, 0, seed=70 + k)
            u[0, 3, 40:44], v[0, 3, :] = 0.0, 255.0
            Ca = oracle.costvolume(u, v, dmin, dmin + L - 1, "none", "ad", np.inf, 3)
            assert Ca[np.isfinite(Ca)].max() == 255.0
            du, dv = ctx.upload_image(u), ctx.upload_image(v)
            cv = ctx.costvolume_dev(du, dv, dmin, dmin + L - 1, "none", "ad", float("inf"), 3, into=cv)
            S, o, kk = ctx.aggregate_dev(cv, 6.0, 60000.0, 8, 3, 1, 1, None, "vfit", want_S=True)
            Sa, oa, ca = oracle.mgm(Ca, dmin, 6.0, 60000.0, 8, 3, 1, 1)
            ra, rca = oracle.refine(Sa, dmin, "vfit", oa, ca)
            assert (ndiff(cv.download(), Ca), ndiff(S.download(), Sa), ndiff(o.download()[0], ra), ndiff(kk.download()[0], rca)) == (0, 0, 0, 0), k
            for h in (du, dv, S, o, kk):
                h.free()
        cv.free()
    finally:
        oracle.set_threads(1)


@pytest.mark.parametrize("L,top", [(128, 700.0), (256, 65534.0), (768, 700.0), (128, 70000.0)])
def test_uploaded_volume_of_whole_numbers_beyond_one_byte(ctx, oracle, L, top):
    """A volume computed elsewhere and uploaded (the drop-in path for the reference's own cost volumes): whole-number costs up
    to 65534 are aggregated from a two-byte copy where the pass kernels read one (up to 512 labels), anything else from the
    fp32 volume -- same results either way."""
    nx, ny = 333, 60
    C = np.rint(synth.raw_volume(nx, ny, L, seed=L + int(top) % 97, maxcost=60, inf_frac=0.02)).astype(np.float32)
    C[5, 7, 3], C[9, 100, L // 2] = np.float32(top), np.float32(255.0)
    cv = ctx.upload_volume(C, -L // 2)
    threads(oracle)
    try:
        for FH, P1, P2 in ((0, 8.0, 32.0), (1, 2.0, 20000.0)):
            S, o, kk = ctx.aggregate_dev(cv, P1, P2, 8, 3, FH, 1, None, None, want_S=True)
            Sa, oa, ca = oracle.mgm(C, -L // 2, P1, P2, 8, 3, FH, 1)
            assert (ndiff(S.download(), Sa), ndiff(o.download()[0], oa), ndiff(kk.download()[0], ca)) == (0, 0, 0), FH
            for h in (S, o, kk):
                h.free()
    finally:
        oracle.set_threads(1)
    cv.free()
