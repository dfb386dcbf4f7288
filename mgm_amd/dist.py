"""One volume sharded by DIRECTION over the GPUs of a node (BASELINE cfg4).

Every rank builds the full cost volume from the images (cheaper than broadcasting 13 GB), runs
its contiguous share of the passes with ``mgm_aggregate_passes_dev`` and keeps their Lr volumes.
The per-pixel sum over directions must be taken in PASS ORDER in fp32 (a different order flips
argmins), so an all-reduce is out: instead the image rows are cut into one slab per
rank and the ranks exchange slabs with grouped point-to-point transfers (``batch_isend_irecv`` =
ncclGroupStart/ncclSend/ncclRecv/ncclGroupEnd on RCCL; every GPU talks to every other GPU at
once, one peer per xGMI link).  Each rank then owns all directions of its rows and finishes
them locally with ``mgm_wta_rows_dev`` (ordered sum, over-count fix, WTA, V-fit).

The exchange code is backend-agnostic (torch tensors): the CPU test runs it over gloo with
host tensors, the GPU path runs it over RCCL with zero-copy views of the workspace.
"""
import numpy as np


def row_slabs(ny, world):
    """Contiguous row ranges, one per rank: [(row0, nrows)], sizes differing by at most one."""
    base, extra = divmod(ny, world)
    out, r0 = [], 0
    for g in range(world):
        n = base + (1 if g < extra else 0)
        out.append((r0, n))
        r0 += n
    return out


def passes_of_rank(NDIR, world, rank):
    """Contiguous block of passes of `rank`: (first, count); blocks differ by at most one pass."""
    base, extra = divmod(NDIR, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def owner_of_pass(p, NDIR, world):
    for g in range(world):
        f, n = passes_of_rank(NDIR, world, g)
        if f <= p < f + n:
            return g
    raise ValueError(p)


def exchange_lr(lr_local, NDIR, ny, dist, group=None, like=None):
    """lr_local: list of [ny, nx, L] tensors, the Lr volumes of this rank's passes (in pass order).
    Returns a [NDIR, my_rows, nx, L] tensor with every pass's slab of this rank's rows.
    A rank beyond the NDIR-th runs no pass (world > NDIR) but still owns rows: it passes `like`, any tensor
    whose last two dimensions, dtype and device are those of the Lr volumes."""
    import torch
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    first, count = passes_of_rank(NDIR, world, rank)
    assert len(lr_local) == count
    slabs = row_slabs(ny, world)
    r0, nr = slabs[rank]
    ref = lr_local[0] if count else like
    if ref is None:
        raise ValueError("exchange_lr: a rank without passes must say what the volumes look like (like=)")
    nx, L = ref.shape[-2], ref.shape[-1]
    recv = torch.empty((NDIR, nr, nx, L), dtype=ref.dtype, device=ref.device)
    ops = []
    for p in range(NDIR):  # receives, in pass order per peer
        o = owner_of_pass(p, NDIR, world)
        if o == rank:
            recv[p].copy_(lr_local[p - first][r0:r0 + nr])
        elif nr:
            ops.append(dist.P2POp(dist.irecv, recv[p], o, group))
    for k in range(count):  # sends, in pass order per peer
        for g in range(world):
            g0, gn = slabs[g]
            if g != rank and gn:
                ops.append(dist.P2POp(dist.isend, lr_local[k][g0:g0 + gn], g, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return recv


def ordered_sum_numpy(slabs, C_rows, fix_overcount):
    """CPU restatement of the per-slab ordered sum (mgm_core.cc:582-599) used by the gloo test."""
    S = np.zeros_like(slabs[0])
    for p in range(len(slabs)):
        S = S + slabs[p]
    if fix_overcount == 1:
        S = S - np.float32(len(slabs) - 1) * C_rows
    return S


class _DevMem:
    """Zero-copy view of raw device memory for torch.as_tensor."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def device_view(ptr, shape):
    import torch
    return torch.as_tensor(_DevMem(ptr, shape), device="cuda")


def aggregate_direction_sharded(ctx, cv, P1, P2, NDIR, MGM, use_fh, fix_overcount, refine, dist, group=None, w8=None):
    """The whole aggregation of ONE volume across the ranks of `group`.  Returns (out, outcost) as torch
    tensors [ny, nx] on every rank (all-gathered rows)."""
    import torch
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nx, ny, dmin, dmax = cv.dims
    L = dmax - dmin + 1
    if world == 1:  # nothing to shard: the plain call (which may also pad odd label counts and pick its occupancy)
        _, o, c = ctx.aggregate_dev(cv, P1, P2, NDIR, MGM, use_fh, fix_overcount, w8, refine)
        ctx.synchronize()
        views = [device_view(ctx.lib.mgm_img_device_ptr(im.h), (ny, nx)).clone() for im in (o, c)]
        o.free(), c.free()
        return views[0], views[1]
    first, count = passes_of_rank(NDIR, world, rank)
    if count:  # (world > NDIR: the ranks beyond the NDIR-th run no pass; they still sum and search their rows)
        ctx.aggregate_passes_dev(cv, P1, P2, MGM, use_fh, first, count, w8)
    ctx.synchronize()  # Lr volumes complete before RCCL reads them (different streams)
    lr_local = [device_view(ctx.lr_device_ptr(k), (ny, nx, L)) for k in range(count)]
    recv = exchange_lr(lr_local, NDIR, ny, dist, group, like=torch.empty((0, nx, L), dtype=torch.float32, device="cuda"))
    torch.cuda.synchronize()
    slabs = row_slabs(ny, world)
    r0, nr = slabs[rank]
    out = torch.empty((max(nr, 1), nx), dtype=torch.float32, device="cuda")
    outc = torch.empty_like(out)
    if nr:
        ctx.wta_rows_dev(cv, r0, nr, recv.data_ptr(), NDIR, fix_overcount, refine, out.data_ptr(), outc.data_ptr())
    ctx.synchronize()
    maxr = max(n for _, n in slabs)
    pad = lambda t: torch.cat([t[:nr], t.new_zeros((maxr - nr, nx))]) if nr < maxr else t[:nr]
    go = [torch.empty((maxr, nx), dtype=torch.float32, device="cuda") for _ in range(world)]
    gc = [torch.empty_like(go[0]) for _ in range(world)]
    dist.all_gather(go, pad(out).contiguous(), group)
    dist.all_gather(gc, pad(outc).contiguous(), group)
    full_o = torch.cat([go[g][:slabs[g][1]] for g in range(world)])
    full_c = torch.cat([gc[g][:slabs[g][1]] for g in range(world)])
    return full_o, full_c
