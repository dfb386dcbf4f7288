"""Ragged volumes on their RANGE-PROPORTIONAL copies (mgm_pass_rel.hip, k_wta_rel; round 5) against the dense-hull kernels
of the same library, bit for bit, inside one process (MGM_HIP_REL is read at every call) -- the dense-hull path is what
tests/test_gpu_cli.py pins on the reference binary for ragged ranges, and those command lines now run through the new
kernels as well.  The timings tell which kernels really ran."""
import os

import numpy as np
import pytest

import mgm_amd
from helpers import ndiff
from mgm_amd import synth

pytestmark = pytest.mark.gpu


def ranges(gt, dmin, dmax, half, seed, jitter=3):
    rng = np.random.default_rng(seed)
    lo = gt - half + rng.integers(-jitter, jitter + 1, size=gt.shape)
    hi = gt + half + rng.integers(-jitter, jitter + 1, size=gt.shape)
    lo, hi = np.clip(lo, dmin, dmax), np.clip(hi, dmin, dmax)
    hi = np.maximum(hi, lo)
    return lo.astype(np.float32), hi.astype(np.float32)


CASES = [
    # nx, ny, dmin, dmax, half window, FH, MGM, NDIR, P1, P2, weights, refine, fix
    (96, 64, -40, 10, 6, 0, 3, 8, 8.0, 32.0, None, "vfit", 1),
    (96, 64, -40, 10, 6, 1, 3, 8, 2.0, 20000.0, None, "vfit", 1),
    (131, 77, -100, 20, 24, 1, 3, 8, 2.0, 20000.0, None, "vfit", 1),     # the bench's window (+-24 of a wide hull)
    (131, 77, -100, 20, 24, 0, 4, 8, 8.0, 32.0, None, "cubic", 1),
    (131, 77, -100, 20, 24, 0, 1, 4, 8.0, 32.0, None, "parabola", 0),    # no over-count fix: labels outside a range hold 0
    (96, 64, -40, 10, 10, 1, 4, 8, 1.5, 9.0, "three", "parabolaOCV", 1),  # free-form weights, small P2 (the cap binds)
    (96, 64, -40, 10, 10, 0, 3, 8, 8.0, 32.0, "three", None, 1),
    (96, 64, -40, 10, 10, 1, 1, 2, 2.0, 30.0, None, "vfit", 0),
    (64, 200, -70, 0, 27, 1, 3, 8, 2.0, 20000.0, None, "vfit", 1),       # tall image: several bands per pass; windows of up to 61 labels
    (300, 40, -30, 30, 3, 0, 3, 8, 8.0, 32.0, None, "vfit", 1),          # narrow windows, large shifts between neighbours
    # the weights compute_mgm_weights makes (1 and one other value) with FH: on the hull these must NOT take the two-valued
    # kernels, whose transforms are the producer's (found by the long random campaign: the hull side was wrong)
    (130, 42, -46, 25, 19, 1, 3, 8, 2.0, 20000.0, "image", "vfit", 1),
    (71, 83, -46, 4, 26, 1, 4, 8, 0.75, 20000.0, "image", "cubic", 0),
    (123, 92, -47, 20, 20, 0, 3, 8, 8.0, 32.0, "image", "parabola", 1),  # ... Hirschmueller: those they may take
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d_w%d_fh%d_t%d_o%d" % (c[0], c[1], c[4], c[5], c[6], c[7]))
def test_rel_matches_dense_hull(case):
    nx, ny, dmin, dmax, half, FH, MGM, NDIR, P1, P2, wkind, refine, fix = case
    u, v, gt = synth.stereo_pair(nx, ny, dmin * 3 // 4, max(0, dmax * 3 // 4), seed=11 + nx)
    lo, hi = ranges(gt, dmin, dmax, half, 5 + ny)
    os.environ["MGM_HIP_REL"] = "2"  # (2: unit-weight Hirschmueller volumes too -- the plan keeps those on the hull by default)
    with mgm_amd.Context(0) as ctx:
        cv = ctx.costvolume(u, v, lo, hi, "none", "census", float("inf"), 5)
        w8 = None
        if wkind == "three":
            rng = np.random.default_rng(3)
            w8 = ctx.upload_image(rng.choice(np.array([1.0, 2.5, 4.0], np.float32), size=(8, ny, nx), p=[0.6, 0.25, 0.15]))
        elif wkind == "image":
            w8 = ctx.weights_dev(ctx.upload_image(u), 4.0, 12.0)
        res = {}
        for mode in ("2", "0"):
            os.environ["MGM_HIP_REL"] = mode
            ctx.timing(True)
            ctx.timing_reset()
            _, o, c = ctx.aggregate_dev(cv, P1, P2, NDIR, MGM, FH, fix, w8, refine)
            names = [n for n, _ in ctx.timings()]
            ctx.timing(False)
            res[mode] = (o.download(), c.download(), names)
            # TSGM_ITER's second search in narrowed windows, on whatever Lr volumes that aggregation left
            wlo = ctx.upload_image(np.clip(gt - 3, dmin - 2, dmax)[None].astype(np.float32))
            whi = ctx.upload_image(np.clip(gt + 4, dmin, dmax + 3)[None].astype(np.float32))
            o2, c2 = ctx.wta_windowed_dev(cv, NDIR, fix, refine, wlo, whi)
            res[mode] += (o2.download(), c2.download())
            for h in (o, c, o2, c2, wlo, whi):
                h.free()
        os.environ.pop("MGM_HIP_REL", None)
    assert "k_pass_rel" in res["2"][2], res["2"][2]
    assert "k_pass_rel" not in res["0"][2], res["0"][2]
    for k, what in ((0, "disparity"), (1, "cost"), (3, "windowed disparity"), (4, "windowed cost")):
        a, b = res["2"][k], res["0"][k]
        # a pixel without a finite S: NaN label on both sides
        assert ndiff(a, b) == 0, (what, int(ndiff(a, b)))


def test_rel_is_not_taken_where_it_does_not_apply():
    nx, ny, dmin, dmax = 80, 48, -90, 10
    u, v, gt = synth.stereo_pair(nx, ny, -60, 0, seed=3)
    os.environ.pop("MGM_HIP_REL", None)
    with mgm_amd.Context(0) as ctx:
        # windows wider than 126 labels -> the dense hull (round 6: up to 126 labels fit the 128-slot form)
        lo = np.full(gt.shape, -200.0, np.float32)
        hi = np.full(gt.shape, -65.0, np.float32)  # 136 labels per pixel ...
        hi[3, 5] = -80.0                           # ... and one pixel with another range: a ragged volume
        cv = ctx.costvolume(u, v, lo, hi, "none", "census", float("inf"), 5)
        ctx.timing(True)
        ctx.aggregate_dev(cv, 8.0, 32.0, 8, 3, 0, 1, None, "vfit")
        assert "k_pass_rel" not in [n for n, _ in ctx.timings()]
        # P2 = +INF is the operand-order-faithful kernel's (all-INF slabs, INF - INF).  (Round 6: every
        # update function runs on the range-proportional kernels, update_cost2 and update_cost2_trunclinear included.)
        lo, hi = ranges(gt, dmin, dmax, 8, 9)
        cv2 = ctx.costvolume(u, v, lo, hi, "none", "census", float("inf"), 5)
        ctx.timing_reset()
        ctx.aggregate_dev(cv2, 2.0, float("inf"), 4, 2, 1, 1, None, "vfit")
        assert "k_pass_rel" not in [n for n, _ in ctx.timings()]
        ctx.timing_reset()
        S, _, _ = ctx.aggregate_dev(cv2, 8.0, 32.0, 4, 3, 0, 1, None, "vfit", want_S=True)  # (round 6: S wanted is served from the relative copy, k_rel_S)
        assert "k_rel_S" in [n for n, _ in ctx.timings()] and S is not None
        # unit weights, Hirschmueller, ONE volume whose hull exists (absolute differences: K2 writes the hull and the relative copy is
        # gathered from it): a tie in round 5 (the hull's queue kernels kept it), the range-proportional kernels' since round 6
        cv3 = ctx.costvolume(np.floor(u / 4), np.floor(v / 4), lo, hi, "none", "ad", float("inf"), 5)
        ctx.timing_reset()
        ctx.aggregate_dev(cv3, 8.0, 32.0, 4, 3, 0, 1, None, "vfit")
        assert "k_pass_rel" in [n for n, _ in ctx.timings()]
        ctx.timing_reset()
        ctx.aggregate_dev(cv3, 2.0, 30.0, 4, 3, 1, 1, None, "vfit")  # (FH on the same volume: the gathered copy is used)
        assert "k_pass_rel" in [n for n, _ in ctx.timings()]
        # ... but a single-word census volume only HAS the relative copy (k_cost_census_rel): the hull would have to be expanded first
        ctx.timing_reset()
        ctx.aggregate_dev(cv2, 8.0, 32.0, 4, 3, 0, 1, None, "vfit")
        assert "k_pass_rel" in [n for n, _ in ctx.timings()]
        assert "k_expand" not in [n for n, _ in ctx.timings()]
        # and whoever asks for that hull gets the one the general kernel writes
        hull = cv2.download()
        os.environ["MGM_HIP_REL"] = "0"
        ref = ctx.costvolume(u, v, lo, hi, "none", "census", float("inf"), 5).download()
        os.environ.pop("MGM_HIP_REL", None)
        assert hull.shape == ref.shape and np.array_equal(hull, ref)
        ctx.timing_reset()
        ctx.aggregate_dev(cv2, 2.0, 30.0, 4, 3, 1, 1, None, "vfit")  # FH: the range-proportional kernels
        assert "k_pass_rel" in [n for n, _ in ctx.timings()]


FUZZ_N = int(os.environ.get("MGM_FUZZ_N", "0"))
FUZZ_BASE = int(os.environ.get("MGM_FUZZ_BASE", "0"))


@pytest.mark.parametrize("seed", range(FUZZ_BASE, FUZZ_BASE + (FUZZ_N or 24)))
def test_rel_random_cases_match_dense_hull(seed):
    """Random shapes, windows, potentials, neighbour counts, weights, refinements: range-proportional kernels == dense hull."""
    rng = np.random.default_rng(991000 + seed)
    nx, ny = int(rng.integers(20, 140)), int(rng.integers(18, 100))
    dmin = -int(rng.integers(20, 120))
    dmax = int(rng.integers(0, 30))
    half = int(rng.integers(1, 29))
    FH = int(rng.integers(0, 2))
    MGM = int(rng.choice([1, 3, 4])) if seed % 3 else int(rng.choice([1, 2, 3, 4]))  # (round 6: TSGM = 2 on a third of the seeds; the others keep their cases)
    NDIR = int(rng.choice([1, 2, 4, 8]))
    P1 = float(rng.choice([0.75, 1.5, 2.0, 8.0]))
    P2 = float(rng.choice([9.0, 32.0, 40.0, 20000.0]))
    wkind = rng.choice(["none", "three", "image"])
    refine = rng.choice([None, "vfit", "parabola", "cubic", "parabolaOCV"])
    fix = int(rng.integers(0, 2))
    win = int(rng.choice([3, 5]))
    u, v, gt = synth.stereo_pair(nx, ny, dmin * 3 // 4, max(0, dmax * 3 // 4), seed=seed)
    lo, hi = ranges(gt, dmin, dmax, half, seed, jitter=int(rng.integers(0, 3)))
    trunc = float(rng.choice([float("inf"), float("inf"), 5.0, 12.0]))  # (drawn last: the earlier seeds keep their cases)
    os.environ["MGM_HIP_REL"] = "2"
    res = {}
    with mgm_amd.Context(0) as ctx:
        cv = ctx.costvolume(u, v, lo, hi, "none", "census", trunc, win)
        w8 = None
        if wkind == "three":
            w8 = ctx.upload_image(rng.choice(np.array([1.0, 2.5, 4.0], np.float32), size=(8, ny, nx), p=[0.6, 0.25, 0.15]))
        elif wkind == "image":
            w8 = ctx.weights_dev(ctx.upload_image(u), 4.0, 12.0)
        for mode in ("2", "0"):
            os.environ["MGM_HIP_REL"] = mode
            ctx.timing(True)
            ctx.timing_reset()
            _, o, c = ctx.aggregate_dev(cv, P1, P2, NDIR, MGM, FH, fix, w8, refine)
            names = [n for n, _ in ctx.timings()]
            ctx.timing(False)
            res[mode] = (o.download(), c.download(), names)
        os.environ.pop("MGM_HIP_REL", None)
    what = (seed, nx, ny, dmin, dmax, half, FH, MGM, NDIR, P1, P2, wkind, refine, fix, win, trunc)
    is_ragged = bool((lo != lo.flat[0]).any() or (hi != hi.flat[0]).any())  # (windows that cover a small hull everywhere: a uniform volume)
    fits = int((hi - lo).max()) + 1 <= 126
    assert ("k_pass_rel" in res["2"][2]) == (is_ragged and fits) and "k_pass_rel" not in res["0"][2], what
    assert ndiff(res["2"][0], res["0"][0]) == 0 and ndiff(res["2"][1], res["0"][1]) == 0, what


def test_rel_launches_of_different_walks_share_a_context():
    """FH launches walk their form-0 column passes with exchanged roles (fewer bands), Hirschmueller launches do not: the hand-off region's
    layout and the task table differ although shape, batch and workgroups per CU are the same -- alternating them on one context must not
    reuse either (found by the end-of-round suite: the cache keys named the switch, not what the launch did with it)."""
    nx, ny, dmin, dmax = 150, 60, -60, 0
    pairs = [synth.stereo_pair(nx, ny, -45, 0, seed=170 + k) for k in range(2)]
    res = {}
    for mode in ("1", "0"):
        os.environ["MGM_HIP_REL"] = mode
        try:
            with mgm_amd.Context(0) as ctx:
                cvs = []
                for k, (u, v, gt) in enumerate(pairs):
                    lo, hi = ranges(gt, dmin, dmax, 10, 130 + k)
                    lo[0, 0], hi[0, 1] = dmin, dmax  # (batched volumes share their hull)
                    cvs.append(ctx.costvolume(u, v, lo, hi, "none", "census", float("inf"), 5))
                out = []
                for FH in (1, 0, 1, 0):
                    ctx.timing(True)
                    ctx.timing_reset()
                    _, outs, outcs = ctx.aggregate_batch_dev(cvs, 2.0 if FH else 8.0, 30.0, 8, 3, FH, 1, None, "vfit")
                    names = [n for n, _ in ctx.timings()]
                    ctx.timing(False)
                    assert ("k_pass_rel" in names) == (mode == "1"), (mode, FH, names)
                    out.append(([o.download() for o in outs], [c.download() for c in outcs]))
                res[mode] = out
        finally:
            os.environ.pop("MGM_HIP_REL", None)
    for r in range(4):
        for k in range(2):
            assert ndiff(res["1"][r][0][k], res["0"][r][0][k]) == 0 and ndiff(res["1"][r][1][k], res["0"][r][1][k]) == 0, (r, k)


def test_rel_batch_honours_the_workspace_limit():
    """a batch of ragged volumes whose range-proportional Lr volumes exceed mgm_ctx_set_workspace_limit runs as several
    launches of k_pass_rel (as the dense path does) -- same maps as the unlimited launch."""
    nx, ny, dmin, dmax = 120, 70, -60, 0
    pairs = [synth.stereo_pair(nx, ny, -45, 0, seed=70 + k) for k in range(4)]
    os.environ["MGM_HIP_REL"] = "1"
    try:
        with mgm_amd.Context(0) as ctx:
            cvs = []
            for k, (u, v, gt) in enumerate(pairs):
                lo, hi = ranges(gt, dmin, dmax, 10, 30 + k)
                lo[0, 0], hi[0, 1] = dmin, dmax  # (batched volumes share their hull)
                cvs.append(ctx.costvolume(u, v, lo, hi, "none", "census", float("inf"), 5))
            res = []
            for limit in (0, int(1.3 * 4 * 8 * nx * ny * 64 * 1.08)):  # (room for one volume's eight Lr volumes, not for two)
                ctx.set_workspace_limit(limit)
                ctx.trim()
                ctx.timing(True)
                ctx.timing_reset()
                _, outs, outcs = ctx.aggregate_batch_dev(cvs, 2.0, 30.0, 8, 3, 1, 1, None, "vfit")
                names = [n for n, _ in ctx.timings()]
                ctx.timing(False)
                res.append(([o.download() for o in outs], [c.download() for c in outcs], names.count("k_pass_rel")))
            ctx.set_workspace_limit(0)
    finally:
        os.environ.pop("MGM_HIP_REL", None)
    assert res[0][2] == 1 and res[1][2] == 4, (res[0][2], res[1][2])
    for k in range(4):
        assert ndiff(res[0][0][k], res[1][0][k]) == 0 and ndiff(res[0][1][k], res[1][1][k]) == 0, k


@pytest.mark.parametrize("depth", [2, 3])
def test_pipelined_context_gathers_ragged_volumes(depth):
    """mgm_ctx_set_pipeline with RAGGED volumes: the deferred calls run as one launch of k_pass_rel over the gathered volumes; every
    pair gets the result of its own plain call (FH, weights; windows of 64 and of 128 slots)."""
    nx, ny, dmin, dmax = 110, 60, -150, 0
    for half, w8kind in ((10, None), (45, "image")):
        seeds = list(range(5))
        plain, piped = [], []
        for pipelined in (False, True):
            os.environ["MGM_HIP_REL"] = "1"
            try:
                with mgm_amd.Context(0) as ctx:
                    if pipelined:
                        ctx.set_pipeline(depth)
                    outs, keep = [], []
                    ctx.timing(True)
                    for s in seeds:
                        u, v, gt = synth.stereo_pair(nx, ny, -110, 0, seed=300 + s)
                        lo, hi = ranges(gt, dmin, dmax, half, s)
                        du, dv = ctx.upload_image(u), ctx.upload_image(v)  # (device-form calls: a host-buffer build would run what is deferred)
                        cv = ctx.costvolume_ranged_dev(du, dv, ctx.upload_image(lo[None]), ctx.upload_image(hi[None]), dmin, dmax, "none", "census",
                                                       float("inf"), 5)
                        w8 = ctx.weights_dev(du, 4.0, 12.0) if w8kind else None
                        o, c = ctx.new_image(nx, ny), ctx.new_image(nx, ny)
                        ctx.aggregate_dev(cv, 2.0, 30.0, 8, 3, 1, 1, w8, "vfit", out=o, outcost=c)
                        outs.append((o, c))
                        keep += [cv, w8]
                    ctx.synchronize()
                    names = [n for n, _ in ctx.timings()]
                    res = [(o.download(), c.download()) for o, c in outs]
                (piped if pipelined else plain).extend(res)
                if pipelined:
                    assert names.count("k_pass_rel") < len(seeds), names.count("k_pass_rel")  # (gathered launches)
                else:
                    assert names.count("k_pass_rel") == len(seeds)
            finally:
                os.environ.pop("MGM_HIP_REL", None)
        for k in range(len(seeds)):
            assert ndiff(plain[k][0], piped[k][0]) == 0 and ndiff(plain[k][1], piped[k][1]) == 0, (half, k)
