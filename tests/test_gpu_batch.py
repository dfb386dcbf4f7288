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
