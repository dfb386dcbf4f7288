"""Several volumes in one pass launch (mgm_aggregate_batch_dev): every volume must get exactly the result
of its own mgm_aggregate_dev call -- and, for one case, of the oracle."""
import numpy as np
import pytest

from helpers import ndiff
from mgm_amd import synth

pytestmark = pytest.mark.gpu

MODES = [
    # nx, ny, L, NDIR, MGM, FH, P1, P2, weighted
    (97, 45, 256, 8, 3, 1, 2.0, 20000.0, False),   # second build, compact costs, FH
    (97, 45, 128, 8, 4, 0, 8.0, 32.0, False),      # second build, all four neighbours
    (60, 33, 128, 8, 3, 0, 8.0, 32.0, True),       # weighted Hirschmueller
    (60, 33, 64, 4, 2, 1, 2.0, 50.0, True),        # weighted FH, 4 directions
    (70, 41, 100, 8, 3, 0, 8.0, 32.0, False),      # label count the second build does not take: first build
    (66, 30, 192, 8, 3, 1, 2.0, 20000.0, False),   # 3 labels per lane: compact slabs leave 4 lanes of a DMA piece idle
    (83, 47, 64, 8, 3, 0, 8.0, 32.0, False),       # 64 labels: four volumes share a wave when the batch divides by 4
    (83, 47, 128, 4, 2, 0, 8.0, 32.0, False),      # 128 labels: two volumes per wave (the cfg2 mode)
    (50, 31, 100, 8, 4, 0, 8.0, 32.0, False),      # 100 labels padded to 128, then two volumes per wave
    (83, 47, 128, 8, 3, 1, 2.0, 20000.0, False),   # FH, two volumes per wave: lane-group min-convolution scans
    (61, 29, 64, 8, 4, 1, 1.5, 9.0, False),        # FH, four volumes per wave
    (50, 31, 100, 8, 1, 1, 2.0, 9.0, False),       # FH on 100 labels padded to 128 (padding mask per lane group)
    (40, 23, 128, 4, 2, 1, 2.0, 9.0, False),       # FH with TSGM = 2 keeps one volume per wave
]


@pytest.mark.parametrize("nb", [2, 3, 5, 8, 16])
@pytest.mark.parametrize("mode", MODES)
def test_batch_equals_single(ctx, mode, nb):
    nx, ny, L, NDIR, MGM, FH, P1, P2, weighted = mode
    cvs, w8s, ref = [], [], []
    for b in range(nb):
        C = np.rint(synth.raw_volume(nx, ny, L, seed=100 + b, inf_frac=0.03 * b))  # integers: the compact path
        cvs.append(ctx.upload_volume(C.astype(np.float32), -L // 2))
        if weighted:
            rng = np.random.default_rng(7 + b)
            w = np.where(rng.random((8, ny, nx)) < 0.4, np.float32(1.0 / 3.0), np.float32(1.0)).astype(np.float32)
            w8s.append(ctx.upload_image(w))
        _, o, c = ctx.aggregate_dev(cvs[b], P1, P2, NDIR, MGM, FH, 1, w8s[b] if weighted else None, "vfit")
        ref.append((o.download(), c.download()))
    _, outs, outcs = ctx.aggregate_batch_dev(cvs, P1, P2, NDIR, MGM, FH, 1, w8s if weighted else None, "vfit")
    for b in range(nb):
        assert ndiff(outs[b].download(), ref[b][0]) == 0, "volume %d: labels differ" % b
        assert ndiff(outcs[b].download(), ref[b][1]) == 0, "volume %d: costs differ" % b
    for cv in cvs:
        cv.free()


def test_batch_S_against_oracle(ctx, oracle):
    nx, ny, L, dmin = 80, 37, 128, -90
    P1, P2, NDIR, MGM, FH = 8.0, 32.0, 8, 3, 0
    Cs = [synth.raw_volume(nx, ny, L, seed=300 + b, inf_frac=0.02) for b in range(2)]
    cvs = [ctx.upload_volume(C, dmin) for C in Cs]
    S, outs, outcs = ctx.aggregate_batch_dev(cvs, P1, P2, NDIR, MGM, FH, 1, None, None, want_S=True)
    for b in range(2):
        So, oo, co = oracle.mgm(Cs[b], dmin, P1, P2, NDIR, MGM, FH, 1)
        assert ndiff(S[b].download(), So) == 0
        assert ndiff(outs[b].download().reshape(ny, nx), oo) == 0 and ndiff(outcs[b].download().reshape(ny, nx), co) == 0
    for x in cvs + S:
        x.free()


def test_batch_rejects_mismatched_geometry(ctx):
    import mgm_amd
    a = ctx.upload_volume(synth.raw_volume(40, 30, 64, seed=1), 0)
    b = ctx.upload_volume(synth.raw_volume(41, 30, 64, seed=2), 0)
    with pytest.raises(mgm_amd.MgmError) as e:
        ctx.aggregate_batch_dev([a, b], 8.0, 32.0, 8, 3)
    assert e.value.code == mgm_amd.MGM_ERR_INVALID
    a.free()
    b.free()


def test_batch_survives_a_real_out_of_memory(oracle):
    """The device-OOM fallback of mgm_aggregate_batch_dev with a REAL hipMalloc failure: ballast images leave room
    for about five volumes' workspace, a batch of eight is asked for -- the full reservation fails inside the runtime, the call
    must go on with half the batch (a stale hipErrorOutOfMemory must not surface from the retry's first launch as
    MGM_ERR_HIP) and give the same bits as the oracle; and the context stays usable afterwards."""
    import mgm_amd
    nx, ny, dmin, dmax, NDIR, MGM = 960, 540, -255, 0, 8, 3
    L = dmax - dmin + 1
    nb = 8
    per_vol = 4.0 * nx * ny * L * NDIR * 1.08
    c = mgm_amd.Context(0)
    ballast, dus, dvs, cvs = [], [], [], []
    try:
        for b in range(nb):
            u, v, _ = synth.stereo_pair(nx, ny, -190, 0, seed=777 + b)
            dus.append(c.upload_image(u))
            dvs.append(c.upload_image(v))
            cvs.append(c.costvolume_dev(dus[-1], dvs[-1], dmin, dmax, "none", "census", float("inf"), 5))
        c.synchronize()
        free = c.mem_info()[0]
        want_free = 5.2 * per_vol
        assert free > want_free + (1 << 30), "device too full for this test"
        left = free - want_free
        while left > (64 << 20):  # ballast in pieces of at most 16 GiB
            piece = int(min(left, 16 << 30))
            ballast.append(c.new_image(1 << 14, piece // 4 // (1 << 14), 1))
            left -= piece
        _, outs, outcs = c.aggregate_batch_dev(cvs, 2.0, 20000.0, NDIR, MGM, 1, 1, None, "vfit")
        c.synchronize()
        from oracle.oracle import usable_cpus
        oracle.set_threads(min(32, usable_cpus()))
        try:
            for b in (0, 3, 4, 7):  # (both halves of the batch)
                Ca = cvs[b].download()
                Sa, oa, ca = oracle.mgm(Ca, dmin, 2.0, 20000.0, NDIR, MGM, 1, 1)
                ra, rca = oracle.refine(Sa, dmin, "vfit", oa, ca)
                assert ndiff(rca, outcs[b].download()[0]) == 0 and ndiff(ra, outs[b].download()[0]) == 0, b
        finally:
            oracle.set_threads(1)
        # and the next call -- a plain one that fits -- must not inherit an error either
        _, o1, c1 = c.aggregate_dev(cvs[0], 2.0, 20000.0, NDIR, MGM, 1, 1, None, "vfit")
        assert ndiff(o1.download(), outs[0].download()) == 0
        ballast += [o1, c1] + list(outs) + list(outcs)
    finally:
        for h in ballast + dus + dvs + cvs:  # (images and volumes are the caller's: closing the context does not free them)
            h.free()
        c.close()
