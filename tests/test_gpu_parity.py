"""Parity of the HIP path (through the C ABI of libmgm_hip.so) with the CPU oracle and
with the golden vectors of the compiled reference.  Bit-exact for volumes, labels and
costs; V-fit refined disparities are compared bit-exactly as well (tolerance allowed by
the task: 1e-5).  Run on a real MI355X:  python -m pytest tests -m gpu
"""
import numpy as np
import pytest

from helpers import golden_cases, labels_equal, load_golden, ndiff
from mgm_amd import synth

pytestmark = pytest.mark.gpu

VFIT_TOL = 1e-5  # stated tolerance for the floating-point refinement; the tests assert 0 differing bits


def test_device_selftest_exact_division_by_three(ctx):
    assert ctx.selftest_div3() == 0  # all 2^32 inputs


# ---- golden vectors of the reference --------------------------------------------
@pytest.mark.parametrize("name", golden_cases("cv"))
def test_golden_costvolume(ctx, name):
    g = load_golden(name)
    u, v = ctx.upload_image(g["u"]), ctx.upload_image(g["v"])
    cv = ctx.costvolume_dev(u, v, int(g["dmin"]), int(g["dmax"]), str(g["prefilter"]), str(g["distance"]),
                            float(g["truncDist"]), int(g["census_win"]))
    assert ndiff(cv.download(), g["C"]) == 0
    for h in (cv, u, v):
        h.free()


@pytest.mark.parametrize("name", golden_cases("weights"))
def test_golden_weights(ctx, name):
    g = load_golden(name)
    u = ctx.upload_image(g["u"])
    w = ctx.weights_dev(u, float(g["aP"]), float(g["aThresh"]))
    assert ndiff(w.download(), g["w8"]) == 0
    u.free(), w.free()


@pytest.mark.parametrize("name", golden_cases("agg"))
def test_golden_aggregation(ctx, name):
    g = load_golden(name)
    dmin = int(g["dmin"])
    cv = ctx.upload_volume(g["C"], dmin)
    args = (float(g["P1"]), float(g["P2"]), int(g["NDIR"]), int(g["MGM"]), int(g["FH"]), int(g["FIX"]), g.get("w8"))
    S, out, outc = ctx.aggregate(cv, *args, None, want_S=True)
    assert ndiff(S.download(), g["S"]) == 0
    assert ndiff(outc, g["outcost"]) == 0
    assert labels_equal(out, g["out"], outc)
    # stand-alone refinement on the materialised S, then the fused form
    ro, rc = ctx.refine(S, "vfit", g["out"], g["outcost"])
    assert ndiff(ro, g["out_vfit"]) == 0 and ndiff(rc, g["outcost_vfit"]) == 0
    _, fo, fc = ctx.aggregate(cv, *args, "vfit", want_S=False)
    # (a pixel without a finite S has the reference's UNINITIALISED label, mgm_core.cc:594, and what the refinement makes of
    # it -- label and cost -- is as undefined: the fused form leaves NaN / +INF there.  Only the NaN goldens have such pixels.)
    fin = np.isfinite(g["outcost"])
    if not name.startswith("agg_nan24_"):  # every other golden is finite everywhere: the whole maps, bit for bit
        assert fin.all(), name
        assert ndiff(fo, g["out_vfit"]) == 0 and ndiff(fc, g["outcost_vfit"]) == 0
    assert ndiff(fo[fin], g["out_vfit"][fin]) == 0 and ndiff(fc[fin], g["outcost_vfit"][fin]) == 0
    assert not np.isfinite(fc[~fin]).any()
    assert np.nanmax(np.abs(fo[fin] - g["out_vfit"][fin]), initial=0) <= VFIT_TOL
    # the other refinements of the reference's table (refine.h; the cubic evaluates in double): as a second kernel
    # on S, both stand-alone and behind mgm_aggregate
    for meth in ("parabola", "cubic", "parabolaOCV"):
        ro, rc = ctx.refine(S, meth, g["out"], g["outcost"])
        assert ndiff(ro[fin], g["out_" + meth][fin]) == 0 and ndiff(rc, g["outcost_" + meth]) == 0, meth
        _, fo, fc = ctx.aggregate(cv, *args, meth, want_S=False)
        assert ndiff(fo[fin], g["out_" + meth][fin]) == 0 and ndiff(fc[fin], g["outcost_" + meth][fin]) == 0, meth
    S.free(), cv.free()


@pytest.mark.parametrize("meth", ["parabola", "cubic", "parabolaOCV"])
def test_refinements_vs_oracle_on_a_real_volume(ctx, oracle, meth):
    nx, ny, dmin, dmax = 120, 50, -63, 64
    u, v, _ = synth.stereo_pair(nx, ny, -30, 30)
    C = oracle.costvolume(u, v, dmin, dmax, "none", "census", np.inf, 5)
    So, oo, co = oracle.mgm(C, dmin, 8.0, 32.0, 8, 3, 0, 1)
    ro, rc = oracle.refine(So, dmin, meth, oo, co)
    cv = ctx.upload_volume(C, dmin)
    _, fo, fc = ctx.aggregate(cv, 8.0, 32.0, 8, 3, 0, 1, None, meth, want_S=False)
    fin = np.isfinite(co)
    assert ndiff(fo[fin], ro[fin]) == 0 and ndiff(fc, rc) == 0
    cv.free()


# ---- seeded sweeps against the oracle ----------------------------------------------
SHAPES = [(40, 37, 12), (70, 35, 64), (35, 70, 100), (50, 40, 128), (130, 70, 151), (45, 33, 192), (64, 48, 256),
          (48, 40, 300), (40, 36, 384), (40, 36, 512)]
MODES = [(8, 3, 0, 8.0, 32.0), (4, 2, 0, 8.0, 32.0), (8, 4, 0, 8.0, 32.0), (8, 1, 0, 8.0, 32.0),
         (8, 3, 1, 2.0, 20000.0), (4, 2, 1, 2.0, 9.0), (8, 4, 1, 1.5, np.inf), (3, 1, 1, 2.0, 9.0)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%dx%d" % s)
def test_aggregation_modes_vs_oracle(ctx, oracle, shape):
    nx, ny, L = shape
    C = synth.raw_volume(nx, ny, L, inf_frac=0.03)
    dmin = -5
    cv = ctx.upload_volume(C, dmin)
    rng = np.random.default_rng(7)
    for (NDIR, MGM, FH, P1, P2) in MODES:
        for wmode in (0, 1):
            w8 = None
            if wmode:
                w8 = np.where(rng.random((8, ny, nx)) < 0.5, 0.3 if FH else 4.0, 1.0).astype(np.float32)
            Sa, oa, ca, lra = oracle.mgm(C, dmin, P1, P2, NDIR, MGM, FH, 1, w8, dump_lr=True)
            Sb, ob, cb = ctx.aggregate(cv, P1, P2, NDIR, MGM, FH, 1, w8, None, want_S=True)
            tag = (shape, NDIR, MGM, FH, wmode)
            for p in range(NDIR):
                assert ndiff(lra[p], ctx.debug_lr(cv, p)) == 0, tag + ("Lr", p)
            assert ndiff(Sa, Sb.download()) == 0, tag
            assert ndiff(ca, cb) == 0 and labels_equal(oa, ob, ca), tag
            fin = np.isfinite(ca)
            ora, cra = oracle.refine(Sa, dmin, "vfit", np.where(fin, oa, dmin), ca)
            _, fo, fc = ctx.aggregate(cv, P1, P2, NDIR, MGM, FH, 1, w8, "vfit", want_S=False)
            assert ndiff(ora[fin], fo[fin]) == 0 and ndiff(cra, fc) == 0, tag + ("vfit",)
            Sb.free()
    cv.free()


COSTS = [(1, (40, 23), (-7, 8), "none", "ad", 3, np.inf), (3, (40, 23), (-7, 8), "none", "ad", 3, 25.0),
         (1, (40, 23), (-7, 8), "none", "census", 3, np.inf), (1, (40, 23), (-7, 8), "none", "census", 5, np.inf),
         (3, (40, 23), (-30, 40), "none", "census", 3, np.inf), (3, (40, 23), (-30, 40), "none", "census", 5, np.inf),
         (1, (70, 23), (30, 100), "none", "census", 7, 20.0), (1, (40, 23), (-7, 8), "census", "ad", 3, np.inf),
         (3, (33, 17), (-70, 70), "none", "sd", 3, 50.0), (1, (1, 1), (0, 0), "none", "ad", 3, np.inf),
         (1, (5, 3), (-300, 211), "none", "census", 3, np.inf),
         # single-word census at label counts with a compact form: K2 writes bytes only, the fp32 volume is decoded on
         # demand (trunc +INF or a small integer); a fractional trunc or a two-word descriptor takes the general kernel
         (1, (70, 23), (-40, 23), "none", "census", 5, np.inf), (1, (70, 23), (-100, 27), "none", "census", 5, 7.0),
         (1, (70, 23), (-200, 55), "none", "census", 5, 7.5), (1, (300, 9), (-255, 0), "none", "census", 7, np.inf),
         (1, (300, 9), (-255, 0), "none", "census", 5, 0.0), (1, (90, 31), (0, 63), "none", "census", 3, np.inf),
         (1, (70, 23), (-100, 91), "none", "census", 5, np.inf), (1, (40, 9), (-200, 183), "none", "census", 3, 6.0),  # 192, 384 labels
         # sobelx / gblur prefilters (Neumann boundary, the reference's accumulation order) under AD and SD
         (1, (40, 23), (-7, 8), "sobelx", "ad", 3, np.inf), (3, (33, 17), (-20, 12), "sobelx", "sd", 3, 900.0),
         (1, (40, 23), (-7, 8), "gblur", "ad", 3, np.inf), (3, (33, 17), (-20, 12), "gblur", "sd", 3, 50.0),
         (1, (3, 2), (-2, 2), "gblur", "ad", 3, np.inf), (1, (70, 23), (-40, 23), "gblur", "ad", 3, 30.0),
         # clipped NCC (window sums in float, normalisation in double) and the Birchfield-Tomasi costs
         (1, (40, 23), (-7, 8), "none", "ncc", 3, np.inf), (3, (33, 17), (-20, 12), "none", "ncc", 5, 1.5),
         (1, (70, 23), (-40, 23), "gblur", "ncc", 7, np.inf), (1, (40, 23), (-7, 8), "none", "btad", 3, np.inf),
         (3, (33, 17), (-20, 12), "sobelx", "btsd", 3, 900.0), (1, (70, 23), (-40, 23), "none", "btsd", 3, 50.0),
         (1, (2, 2), (-1, 1), "none", "btad", 3, np.inf),
         # k_cost_ncc: widths that leave a partial workgroup tile, 7x7 windows, label counts off the wave width
         (1, (61, 19), (-60, 39), "none", "ncc", 7, np.inf), (3, (37, 11), (-33, 30), "none", "ncc", 3, 2.0),
         (1, (130, 9), (-255, 0), "none", "ncc", 5, np.inf), (2, (35, 8), (-3, 70), "none", "ncc", 5, np.inf),
         # k_bt_spans + k_cost_btx (label counts that are multiples of four, any width): several label turns per lane, a
         # colour pair, truncation, a count off the wave width
         (1, (64, 21), (-100, 27), "none", "btad", 3, np.inf), (3, (36, 11), (-300, 211), "none", "btsd", 3, 50.0),
         (1, (96, 9), (-40, 27), "none", "btad", 3, 2.5), (2, (8, 5), (-3, 0), "none", "btsd", 3, np.inf),
         # the same kernel writes the fp32 volumes of costs without a compact form: census over two / three descriptor words
         # (halves / thirds of bit counts), differences of blurred images, of float-valued ones after sobelx
         (1, (64, 21), (-100, 27), "none", "census", 7, np.inf), (3, (36, 11), (-60, 3), "none", "census", 5, 20.0),
         (1, (64, 21), (-40, 23), "gblur", "ad", 3, 30.0), (3, (36, 11), (-300, 211), "gblur", "sd", 3, np.inf),
         (1, (300, 8), (-255, 0), "none", "census", 7, 11.5)]


@pytest.mark.parametrize("case", COSTS, ids=lambda c: "%dch-%s-%s-w%d" % (c[0], c[3], c[4], c[5]))
def test_costvolume_vs_oracle(ctx, oracle, case):
    nch, (nx, ny), (dmin, dmax), pre, dist, win, td = case
    u, v, _ = synth.stereo_pair(nx, ny, max(dmin, -nx // 4), min(dmax, nx // 4), nch=nch)
    a = oracle.costvolume(u, v, dmin, dmax, pre, dist, td, win)
    # host-buffer entry point with per-pixel range images, like the reference's main()
    dminI = np.full((ny, nx), dmin, np.float32)
    dmaxI = np.full((ny, nx), dmax, np.float32)
    cv = ctx.costvolume(u, v, dminI, dmaxI, pre, dist, td, win)
    assert ndiff(a, cv.download()) == 0
    if dmax - dmin + 1 >= 64:  # ... and the volume aggregates to the oracle's result whichever copy the kernels read
        So, oo, co = oracle.mgm(a, dmin, 8.0, 32.0, 8, 3, 0, 1)
        S, o, c = ctx.aggregate(cv, 8.0, 32.0, 8, 3, 0, 1, None, None, want_S=True)
        assert ndiff(S.download(), So) == 0 and ndiff(o, oo) == 0 and ndiff(c, co) == 0
        S.free()
    cv.free()


@pytest.mark.parametrize("nch,win", [(1, 5), (3, 3), (1, 7)])
def test_ncc_volume_of_a_few_million_cells(ctx, oracle, nch, win):
    """The whole volume of k_cost_ncc (window statistics once per pixel, products per cell, the normalisation in double) at a
    size where every rounding case of the normalisation turns up, cell by cell against the oracle."""
    nx, ny, dmin, dmax = 400, 96, -127, 0
    u, v, _ = synth.stereo_pair(nx, ny, -90, 0, seed=17 + win, nch=nch)
    from oracle.oracle import usable_cpus
    oracle.set_threads(min(16, usable_cpus()))
    try:
        a = oracle.costvolume(u, v, dmin, dmax, "none", "ncc", np.inf, win)
    finally:
        oracle.set_threads(1)
    du, dv = ctx.upload_image(u), ctx.upload_image(v)
    cv = ctx.costvolume_dev(du, dv, dmin, dmax, "none", "ncc", float("inf"), win)
    assert ndiff(a, cv.download()) == 0
    for h in (du, dv, cv):
        h.free()


def test_ragged_costvolume(ctx, oracle):
    """-m/-M range images: the volume spans the hull of all ranges; a pixel owns only its own range (the rest reads +INF,
    dvec.cc:129) and the "no finite cost => zeros" rule looks at that range alone (mgm_costvolume.h:414-421)."""
    nx, ny = 60, 25
    u, v, gt = synth.stereo_pair(nx, ny, -14, 6)
    rng = np.random.default_rng(4)
    lo = (gt - rng.integers(1, 5, size=gt.shape)).astype(np.float32) + np.float32(0.3)
    hi = lo + rng.integers(1, 9, size=gt.shape).astype(np.float32)
    lo[0, :4], hi[0, :4] = 70, 75                     # every hypothesis of these pixels lies outside the right image
    cv = ctx.costvolume(u, v, lo, hi, "none", "census", np.inf, 5)
    _, _, dmin, dmax = cv.dims
    ilo, ihi = lo.astype(np.int32), hi.astype(np.int32)  # C's (int) conversion: towards zero
    assert dmin == ilo.min() and dmax == ihi.max()
    full = oracle.costvolume(u, v, dmin, dmax, "none", "census", np.inf, 5)
    # the oracle's own zero rule looks at the whole hull: recompute "finite" from the geometry
    o = np.arange(dmin, dmax + 1)[None, None, :]
    x = np.arange(nx)[None, :, None]
    inside = (x + o >= 0) & (x + o < nx)
    own = (o >= ilo[..., None]) & (o <= ihi[..., None])
    want = np.where(own & inside, full, np.float32(np.inf)).astype(np.float32)
    none_finite = ~(own & inside).any(axis=2)
    want[none_finite[..., None] & own] = 0
    assert ndiff(cv.download(), want) == 0
    cv.free()


def test_unsupported_and_invalid_inputs_fail_loudly(ctx):
    import mgm_amd
    u, v, _ = synth.stereo_pair(16, 8, -3, 3)
    du, dv = ctx.upload_image(u), ctx.upload_image(v)
    cv = ctx.costvolume_dev(du, dv, -3, 3)
    for bad in (dict(NDIR=9, MGM=2), dict(NDIR=4, MGM=5), dict(NDIR=0, MGM=1)):
        with pytest.raises(mgm_amd.MgmError) as e:
            ctx.aggregate(cv, 8.0, 32.0, bad["NDIR"], bad["MGM"])
        assert e.value.code == mgm_amd.MGM_ERR_INVALID
    # unknown cost / refinement names fall back silently, as in the reference
    cv2 = ctx.costvolume_dev(du, dv, -3, 3, "nope", "nope")
    assert ndiff(cv.download(), cv2.download()) == 0
    for h in (cv, cv2, du, dv):
        h.free()


def test_both_builds_of_the_pass_kernel_agree(oracle):
    """K3 exists twice (register-prefetch build, LDS-DMA build); they must agree bit for bit."""
    import os
    import mgm_amd
    C = synth.raw_volume(200, 120, 128, inf_frac=0.01)
    res = []
    for build in ("1", "0"):
        os.environ["MGM_HIP_PASS_BUILD"] = build
        c = mgm_amd.Context(0)
        cv = c.upload_volume(C, 0)
        S, o, oc = c.aggregate(cv, 8.0, 32.0, 8, 3, 0, 1, None, "vfit", want_S=True)
        res.append((S.download(), o, oc))
        c.close()
    os.environ.pop("MGM_HIP_PASS_BUILD")
    Sa, oa, ca = oracle.mgm(C, 0, 8.0, 32.0, 8, 3)
    assert ndiff(res[0][0], res[1][0]) == 0 and ndiff(res[0][1], res[1][1]) == 0
    assert ndiff(res[0][0], Sa) == 0


def test_uploaded_volume_with_nan_costs_takes_the_exact_kernel(ctx, oracle):
    """The fast scan-line kernels are compiled NaN-free; an uploaded volume is scanned once per filling and one that holds
    NaN costs is aggregated by the operand-order-faithful kernel (mgm_pass_exact.hip): S, labels and costs equal the
    oracle's -- which the NaN goldens pin on the reference -- bit for bit.  Both scan paths: a label count with a compact
    copy (k_compact) and one without (k_nanscan); the same volume refilled clean goes back to the fast kernels."""
    for L in (64, 50):
        C = synth.raw_volume(40, 21, L, seed=5, inf_frac=0.02)
        C[7, 11, 3] = np.nan
        C[12, 5, :] = np.nan
        for (NDIR, MGM, FH, P1, P2) in ((4, 2, 0, 8.0, 32.0), (8, 3, 1, 2.0, 20000.0), (8, 4, 0, 8.0, np.inf), (3, 2, 1, 2.0, 9.0)):
            cv = ctx.upload_volume(C, -3)
            ctx.timing(True)
            ctx.timing_reset()
            S, o, c = ctx.aggregate(cv, P1, P2, NDIR, MGM, FH, 1, None, None, want_S=True)
            names = {n for n, _ in ctx.timings()}
            ctx.timing(False)
            assert "k_pass_exact" in names and "k_pass2" not in names and "k_pass" not in names
            Sa, oa, ca = oracle.mgm(C, -3, P1, P2, NDIR, MGM, FH, 1)
            assert ndiff(S.download(), Sa) == 0 and ndiff(c, ca) == 0 and labels_equal(o, oa, ca), (L, NDIR, MGM, FH)
            S.free()
            cv.free()
        C2 = np.nan_to_num(C, nan=1.0, posinf=np.inf)
        cv = ctx.upload_volume(C2, 0)
        ctx.timing(True)
        ctx.timing_reset()
        ctx.aggregate(cv, 8.0, 32.0, 4, 2, 0, 1, None, None, want_S=False)
        assert "k_pass_exact" not in {n for n, _ in ctx.timings()}  # clean: the fast kernels
        ctx.timing(False)
        cv.free()


@pytest.mark.parametrize("L", [513, 600, 768, 1000, 1024, 1300, 1536, 2048, 2049, 2500, 4096, 8192, 8193, 12000, 20001])
@pytest.mark.parametrize("mode", [(8, 3, 0, 8.0, 32.0, False), (8, 3, 1, 2.0, 20000.0, False), (4, 2, 1, 2.0, 9.0, False),
                                  (8, 4, 0, 8.0, 32.0, True), (5, 1, 1, 1.5, 40.0, True)],
                         ids=["O8-T3", "O8-T3-FH", "O4-T2-FH", "O8-T4-w", "O5-T1-FH-w"])
def test_more_than_512_labels(ctx, oracle, L, mode):
    """The reference's Dvec has no label limit (dvec.cc:55-64); 513..1024 labels take the second pass-kernel build (round 4;
    weighted ones and 1025..2048 labels the first build with bands of four lines: 12 / 16 / 24 / 32 labels per lane) and the
    generic WTA instance, beyond 2048 the generic kernels (mgm_pass_exact.hip, k_wta_any; FH beyond 8192 labels -- round 4 --
    with its convolution arrays in global scratch instead of the LDS)."""
    NDIR, MGM, FH, P1, P2, weighted = mode
    nx, ny, dmin = (23, 19, -L // 3) if L <= 2048 else (13, 11, -L // 3)
    C = synth.raw_volume(nx, ny, L, seed=L, inf_frac=0.04)
    if L % 2:
        C = (C * np.float32(1.0 / 3.0)).astype(np.float32)
    w8 = None
    if weighted:
        rng = np.random.default_rng(L)
        w8 = np.where(rng.random((8, ny, nx)) < 0.4, np.float32(0.3), np.float32(1.0)).astype(np.float32)
    So, oo, co = oracle.mgm(C, dmin, P1, P2, NDIR, MGM, FH, 1, w8)
    ro, rc = oracle.refine(So, dmin, "vfit", np.where(np.isfinite(co), oo, dmin).astype(np.float32), co)
    cv = ctx.upload_volume(C, dmin)
    S, o, c = ctx.aggregate(cv, P1, P2, NDIR, MGM, FH, 1, w8, "vfit", want_S=True)
    assert ndiff(S.download(), So) == 0
    assert ndiff(c, rc) == 0
    fin = np.isfinite(co)
    assert ndiff(o[fin], ro[fin]) == 0
    S.free(), cv.free()


@pytest.mark.parametrize("L", [768, 1024, 900])
@pytest.mark.parametrize("mode", [(8, 3, 0, 8.0, 32.0), (8, 3, 1, 2.0, 20000.0), (4, 2, 1, 2.0, 9.0), (8, 4, 0, 8.0, 32.0)],
                         ids=["O8-T3", "O8-T3-FH", "O4-T2-FH", "O8-T4"])
def test_768_and_1024_labels_on_the_second_build(ctx, oracle, L, mode):
    """Round 4: 768 and 1024 labels (12 / 16 labels per lane; 513..1024 run padded to them) take the second pass-kernel build
    -- compact costs, deep rings, queues -- instead of the first build's bands of four lines.  Images with several bands per
    pass (7 lines per band) and enough work items for the queues; integer costs (compact) and, for the padded count, fp32."""
    from oracle.oracle import usable_cpus
    NDIR, MGM, FH, P1, P2 = mode
    nx, ny, dmin = 150, 101, -L // 2
    C = synth.raw_volume(nx, ny, L, seed=L + MGM, maxcost=40, inf_frac=0.03)
    if L == 900:
        C = (C * np.float32(0.5)).astype(np.float32)  # half-integers: no compact form
    oracle.set_threads(min(16, usable_cpus()))
    try:
        So, oo, co = oracle.mgm(C, dmin, P1, P2, NDIR, MGM, FH, 1)
        ro, rc = oracle.refine(So, dmin, "vfit", oo, co)
    finally:
        oracle.set_threads(1)
    cv = ctx.upload_volume(C, dmin)
    ctx.timing(True)
    ctx.timing_reset()
    S, o, c = ctx.aggregate(cv, P1, P2, NDIR, MGM, FH, 1, None, "vfit", want_S=True)
    names = {n for n, _ in ctx.timings()}
    ctx.timing(False)
    assert "k_pass2" in names and "k_pass" not in names
    assert ndiff(S.download(), So) == 0 and ndiff(c, rc) == 0 and ndiff(o, ro) == 0
    S.free(), cv.free()


def test_label_count_beyond_the_index_arithmetic_is_refused(ctx):
    """No label limit but the index arithmetic's (4 194 304; the reference's Dvec has none, dvec.cc:60)."""
    import mgm_amd
    with pytest.raises(mgm_amd.MgmError) as e:
        ctx.upload_volume(np.zeros((1, 1, (1 << 22) + 1), np.float32), 0)
    assert e.value.code == mgm_amd.MGM_ERR_UNSUPPORTED


def test_hand_off_slots_all_rewritten(ctx, oracle, monkeypatch):
    """MGM_HIP_CHECK_TAGS=1: after every pass launch the library scans the self-validating hand-off slots of the launched
    passes -- every word must carry the launch's tag (the invariant the protocol rests on: each slot written exactly once per
    launch of its pass).  Shapes that take every variant: several bands per pass, ragged last bands, two strips per line
    (one volume, passes 4-7), volumes sharing waves, padded label counts, one launch per pass into a shared region."""
    monkeypatch.setenv("MGM_HIP_CHECK_TAGS", "1")
    cases = [(200, 150, 256, 8, 3, 1, 2.0, 20000.0, 1), (200, 150, 128, 8, 3, 0, 8.0, 32.0, 1), (131, 77, 192, 8, 4, 0, 8.0, 32.0, 1),
             (90, 140, 128, 4, 2, 0, 8.0, 32.0, 4), (70, 50, 64, 8, 3, 1, 2.0, 9.0, 8), (150, 61, 100, 8, 3, 0, 8.0, 32.0, 2),
             (64, 200, 384, 5, 1, 1, 1.5, 40.0, 1)]
    for nx, ny, L, NDIR, MGM, FH, P1, P2, nb in cases:
        Cs = [synth.raw_volume(nx, ny, L, seed=100 + k, inf_frac=0.02) for k in range(nb)]
        cvs = [ctx.upload_volume(C, -3) for C in Cs]
        for rep in range(3):  # (the tag alternates: both values, and the second use of freshly cleared slots)
            _, outs, outcs = ctx.aggregate_batch_dev(cvs, P1, P2, NDIR, MGM, FH, 1, None, "vfit")
            ctx.synchronize()
            if rep == 2:
                S, o, c = oracle.mgm(Cs[-1], -3, P1, P2, NDIR, MGM, FH, 1)
                ro, rc = oracle.refine(S, -3, "vfit", np.where(np.isfinite(c), o, -3).astype(np.float32), c)
                fin = np.isfinite(c)
                assert ndiff(outcs[-1].download()[0], rc) == 0 and ndiff(outs[-1].download()[0][fin], ro[fin]) == 0
            for h in outs + outcs:
                h.free()
        for h in cvs:
            h.free()
    # one launch per pass, all passes' slots in one region (the overlapped multi-GPU schedule)
    C = synth.raw_volume(160, 130, 128, seed=7)
    cv = ctx.upload_volume(C, 0)
    for rep in range(3):
        for k in range(8):
            ctx.aggregate_passes_at_dev(cv, 8.0, 32.0, 3, 0, k, 1, k, 8, 8)
        ctx.synchronize()
    _, _, _, lr = oracle.mgm(C, 0, 8.0, 32.0, 8, 3, 0, 1, None, dump_lr=True)
    for k in range(8):
        assert ndiff(ctx.debug_lr(cv, k), lr[k]) == 0, k
    cv.free()
