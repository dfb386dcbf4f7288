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
