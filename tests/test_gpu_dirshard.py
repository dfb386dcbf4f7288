"""Direction sharding on ONE GPU: the ranks of a 2- and 4-way split are emulated one after the
other (same kernels, same slab bookkeeping as mgm_amd/dist.py, device copies instead of RCCL) and
the result must equal the unsharded aggregation bit for bit."""
import numpy as np
import pytest
import torch

from helpers import ndiff
from mgm_amd import dist as mdist
from mgm_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 4, 3])
@pytest.mark.parametrize("mode", [(8, 3, 0, 8.0, 32.0), (8, 3, 1, 2.0, 20000.0), (4, 2, 0, 8.0, 32.0)])
def test_sharded_equals_unsharded(ctx, world, mode):
    NDIR, MGM, FH, P1, P2 = mode
    nx, ny, L, dmin = 150, 61, 128, -100
    C = synth.raw_volume(nx, ny, L, seed=21, inf_frac=0.02)
    cv = ctx.upload_volume(C, dmin)
    _, o_ref, c_ref = ctx.aggregate(cv, P1, P2, NDIR, MGM, FH, 1, None, "vfit", want_S=False)
    slabs = mdist.row_slabs(ny, world)
    recv = [torch.empty((NDIR, n, nx, L), dtype=torch.float32, device="cuda") for _, n in slabs]
    for rank in range(world):  # "rank" runs its passes; its Lr slabs are delivered to every rank's buffer
        first, count = mdist.passes_of_rank(NDIR, world, rank)
        if not count:
            continue
        ctx.aggregate_passes_dev(cv, P1, P2, MGM, FH, first, count)
        ctx.synchronize()
        for k in range(count):
            vol = mdist.device_view(ctx.lr_device_ptr(k), (ny, nx, L))
            for g, (r0, n) in enumerate(slabs):
                recv[g][first + k].copy_(vol[r0:r0 + n])
        torch.cuda.synchronize()
    out = torch.empty((ny, nx), dtype=torch.float32, device="cuda")
    outc = torch.empty_like(out)
    for g, (r0, n) in enumerate(slabs):
        if n:
            ctx.wta_rows_dev(cv, r0, n, recv[g].data_ptr(), NDIR, 1, "vfit", out[r0:].data_ptr(), outc[r0:].data_ptr())
    ctx.synchronize()
    assert ndiff(out.cpu().numpy(), o_ref) == 0 and ndiff(outc.cpu().numpy(), c_ref) == 0
    cv.free()


@pytest.mark.parametrize("overlap", [False, True])
def test_per_process_path_with_one_rank(ctx, oracle, overlap):
    """mgm_amd/dist.py's aggregate_direction_sharded end to end on the one GPU of the box: a process group of ONE RCCL rank
    forced down the sharded path -- the library's stream as torch's current stream, the agreement, the rounds of the
    exchange (own slabs only), one launch per pass when overlapped (mgm_aggregate_passes_at_dev), the row-slab WTA, the
    all-gather, the time-boxed wait and the stage timers -- against the oracle."""
    import socket
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        nx, ny, dmin, dmax = 152, 61, -100, 27
        u, v, _ = synth.stereo_pair(nx, ny, -60, 10, seed=11)
        du, dv = ctx.upload_image(u), ctx.upload_image(v)
        cv = ctx.costvolume_dev(du, dv, dmin, dmax, "none", "census", float("inf"), 5)
        C = oracle.costvolume(u, v, dmin, dmax, "none", "census", np.inf, 5)
        for (NDIR, MGM, FH, P1, P2) in ((8, 3, 0, 8.0, 32.0), (8, 3, 1, 2.0, 20000.0), (4, 2, 0, 8.0, 32.0)):
            S, o, c = oracle.mgm(C, dmin, P1, P2, NDIR, MGM, FH, 1)
            ro, rc = oracle.refine(S, dmin, "vfit", o, c)
            stats = {}
            for rep in range(2):
                go, gc = mdist.aggregate_direction_sharded(ctx, cv, P1, P2, NDIR, MGM, FH, 1, "vfit", dist, timeout_s=60.0, overlap=overlap,
                                                           stats=stats, force_sharded=True)
                assert ndiff(go.cpu().numpy(), ro) == 0 and ndiff(gc.cpu().numpy(), rc) == 0, (NDIR, MGM, FH, overlap, rep)
            assert stats["steps"] == 2 and stats["passes_ms"] > 0 and stats["wta_ms"] > 0
        for h in (cv, du, dv):
            h.free()
    finally:
        dist.destroy_process_group()
