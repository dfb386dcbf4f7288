"""Pipelined contexts (mgm_ctx_set_pipeline): aggregation calls are deferred and gathered, `depth` of them run as ONE batched
launch.  Every result must be bit-identical to the plain call's, whatever the caller does with the objects in between
(refilling a volume that a deferred call still needs, downloading right after the call, post-processing, freeing, calls
that do not fit the waiting ones)."""
import numpy as np
import pytest

import mgm_amd
from helpers import ndiff
from mgm_amd import synth

pytestmark = pytest.mark.gpu

MODES = [
    # nx, ny, dmin, dmax, win, NDIR, MGM, FH, P1, P2
    (300, 200, -127, 0, 3, 4, 2, 0, 8.0, 32.0),      # 128 labels, Hirschmueller (queues, one band per CU)
    (320, 170, -255, 0, 5, 8, 3, 1, 2.0, 20000.0),   # 256 labels, FH
    (200, 150, -150, 0, 5, 8, 3, 0, 8.0, 32.0),      # 151 labels: the gathered batch runs padded
]


def plain_results(mode, seeds, nb):
    nx, ny, dmin, dmax, win, NDIR, MGM, FH, P1, P2 = mode
    res = []
    with mgm_amd.Context(0) as c:
        for s in seeds:
            cvs, ims = [], []
            for b in range(nb):
                u, v, _ = synth.stereo_pair(nx, ny, dmin * 3 // 4, 0, seed=s * 100 + b)
                du, dv = c.upload_image(u), c.upload_image(v)
                ims += [du, dv]
                cvs.append(c.costvolume_dev(du, dv, dmin, dmax, "none", "census", float("inf"), win))
            _, outs, outcs = c.aggregate_batch_dev(cvs, P1, P2, NDIR, MGM, FH, 1, None, "vfit")
            res.append([(o.download(), k.download()) for o, k in zip(outs, outcs)])
            for h in cvs + ims + outs + outcs:
                h.free()
    return res


@pytest.mark.parametrize("depth", [2, 3, 4])
@pytest.mark.parametrize("nb", [1, 2])
@pytest.mark.parametrize("mode", MODES)
def test_pipelined_stream_of_pairs_equals_plain_calls(mode, nb, depth):
    nx, ny, dmin, dmax, win, NDIR, MGM, FH, P1, P2 = mode
    seeds = list(range(1, 8))
    want = plain_results(mode, seeds, nb)
    with mgm_amd.Context(0) as c:
        c.set_pipeline(depth)
        sets = [{"cv": [None] * nb, "o": [c.new_image(nx, ny) for _ in range(nb)], "c": [c.new_image(nx, ny) for _ in range(nb)]}
                for _ in range(depth)]
        pending = []  # (seed index, set) whose results have not been looked at yet
        ims = []
        for k, s in enumerate(seeds):
            st = sets[k % depth]
            if len(pending) == depth:  # the set is about to be reused: its results are read first (download joins the streams)
                j, old = pending.pop(0)
                for b in range(nb):
                    assert ndiff(old["o"][b].download(), want[j][b][0]) == 0 and ndiff(old["c"][b].download(), want[j][b][1]) == 0, (j, b)
            for b in range(nb):
                u, v, _ = synth.stereo_pair(nx, ny, dmin * 3 // 4, 0, seed=s * 100 + b)
                du, dv = c.upload_image(u), c.upload_image(v)
                ims += [du, dv]
                st["cv"][b] = c.costvolume_dev(du, dv, dmin, dmax, "none", "census", float("inf"), win, into=st["cv"][b])
            c.aggregate_batch_dev(st["cv"], P1, P2, NDIR, MGM, FH, 1, None, "vfit", st["o"], st["c"])
            pending.append((k, st))
        c.synchronize()
        for j, old in pending:
            for b in range(nb):
                assert ndiff(old["o"][b].download(), want[j][b][0]) == 0 and ndiff(old["c"][b].download(), want[j][b][1]) == 0, (j, b)


def test_pipelined_context_with_a_careless_caller():
    """ONE volume and ONE pair of output images reused by every call of a depth-2 pipeline: the cost-volume build of call
    n+1 must first run the deferred call n that still needs the volume; a download, a median and a free right behind a call
    must see finished results; switching the pipeline off and on again keeps working."""
    nx, ny, dmin, dmax, win, NDIR, MGM, FH, P1, P2 = MODES[1]
    seeds = [11, 12, 13, 14, 15]
    want = plain_results(MODES[1], seeds, 1)
    with mgm_amd.Context(0) as c:
        c.set_pipeline(2)
        cv, o, k = None, c.new_image(nx, ny), c.new_image(nx, ny)
        for j, s in enumerate(seeds):
            u, v, _ = synth.stereo_pair(nx, ny, dmin * 3 // 4, 0, seed=s * 100)
            du, dv = c.upload_image(u), c.upload_image(v)
            cv = c.costvolume_dev(du, dv, dmin, dmax, "none", "census", float("inf"), win, into=cv)
            c.aggregate_dev(cv, P1, P2, NDIR, MGM, FH, 1, None, "vfit", out=o, outcost=k)
            if j % 2 == 0:
                got = o.download()  # right behind the call
                assert ndiff(got, want[j][0][0]) == 0 and ndiff(k.download(), want[j][0][1]) == 0, j
            elif j == 1:
                med = c.median_dev(o, 1)  # post-processing right behind the call: reads the finished map
                c.synchronize()
                assert ndiff(o.download(), want[j][0][0]) == 0
                med.free()
            if j == 2:
                c.set_pipeline(1)
                c.set_pipeline(2)
            du.free(), dv.free()  # (the images only feed the cost volume build)
        c.synchronize()
        assert ndiff(o.download(), want[-1][0][0]) == 0 and ndiff(k.download(), want[-1][0][1]) == 0
        # a call that wants S takes the unpipelined path of the same context
        S, o2, k2 = c.aggregate_dev(cv, P1, P2, NDIR, MGM, FH, 1, None, "vfit", want_S=True)
        assert ndiff(o2.download(), want[-1][0][0]) == 0
        S.free()
        cv.free()


def test_calls_that_do_not_fit_run_the_waiting_ones_first(oracle):
    """A pipelined context that is fed calls of two geometries, weighted and unweighted, in turn: nothing can be gathered,
    every call must still give its own result (against the oracle)."""
    from oracle.oracle import bits_equal
    with mgm_amd.Context(0) as c:
        c.set_pipeline(3)
        jobs = []
        for k, (nx, ny, L, weighted) in enumerate([(90, 40, 64, False), (70, 50, 128, False), (90, 40, 64, True), (90, 40, 64, False), (70, 50, 128, True)]):
            C = synth.raw_volume(nx, ny, L, seed=900 + k, inf_frac=0.02)
            cv = c.upload_volume(C, -L // 2)
            w = None
            if weighted:
                w = np.where(np.random.default_rng(k).random((8, ny, nx)) < 0.4, np.float32(4.0), np.float32(1.0)).astype(np.float32)
            dw = c.upload_image(w) if weighted else None
            _, o, kk = c.aggregate_dev(cv, 8.0, 32.0, 8, 3, 0, 1, dw, None)
            jobs.append((C, -L // 2, w, o, kk))
        c.synchronize()
        for C, dmin, w, o, kk in jobs:
            Sa, oa, ca = oracle.mgm(C, dmin, 8.0, 32.0, 8, 3, 0, 1, w)
            assert bits_equal(ca, kk.download()[0]) and bits_equal(oa, o.download()[0])


@pytest.mark.parametrize("MGM", [2, 3])
def test_gathered_calls_that_cannot_share_a_launch_run_one_by_one(oracle, MGM):
    """Weighted calls are gathered by the presence of weight planes, but whether planes count as weights is decided on
    their VALUES (all ones = unweighted).  With TSGM = 2 the two kinds are different update functions: the planner
    refuses the gathered launch and the calls run as the caller issued them; with TSGM = 3 they are the same function and
    one weighted launch serves all -- each call with its own result either way."""
    from oracle.oracle import bits_equal
    nx, ny, L = 90, 40, 64
    with mgm_amd.Context(0) as c:
        c.set_pipeline(3)
        jobs = []
        for k in range(3):
            C = synth.raw_volume(nx, ny, L, seed=950 + k, inf_frac=0.02)
            cv = c.upload_volume(C, -L // 2)
            w = np.ones((8, ny, nx), np.float32)
            if k == 1:
                w = np.where(np.random.default_rng(k).random((8, ny, nx)) < 0.4, np.float32(4.0), np.float32(1.0)).astype(np.float32)
            _, o, kk = c.aggregate_dev(cv, 8.0, 32.0, 8, MGM, 1, 1, c.upload_image(w), None)
            jobs.append((C, w, o, kk))
        c.synchronize()
        for C, w, o, kk in jobs:
            Sa, oa, ca = oracle.mgm(C, -L // 2, 8.0, 32.0, 8, MGM, 1, 1, w)
            assert bits_equal(ca, kk.download()[0]) and bits_equal(oa, o.download()[0])
