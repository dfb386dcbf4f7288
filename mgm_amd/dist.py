"""One volume sharded by DIRECTION over the GPUs of a node (BASELINE cfg4), one process per GPU.

Every rank builds the full cost volume from the images (cheaper than broadcasting 13 GB), runs
its contiguous share of the passes with ``mgm_aggregate_passes[_at]_dev`` and keeps their Lr volumes.
The per-pixel sum over directions must be taken in PASS ORDER in fp32 (a different order flips
argmins), so an all-reduce is out: instead the image rows are cut into one slab per
rank and the ranks exchange slabs with grouped point-to-point transfers (``batch_isend_irecv`` =
ncclGroupStart/ncclSend/ncclRecv/ncclGroupEnd on RCCL; every GPU talks to every other GPU at
once, one peer per xGMI link).  Each rank then owns all directions of its rows and finishes
them locally with ``mgm_wta_rows_dev`` (ordered sum, over-count fix, WTA, V-fit).  The CPU analogue in
the reference is mgm_naive_parallelism (mgm_core.cc:710-805): passes in parallel on private Lr volumes,
then accumulated -- there in thread-finish order, here in pass order.

How a step is ordered (no host synchronisation inside it): the library's stream is made torch's current
stream (``torch.cuda.ExternalStream``), so RCCL's transfers wait for the pass kernels through stream
events and the row-slab WTA waits for the transfers the same way.  The exchange is cut into ROUNDS --
round k carries the k-th pass of every rank -- and with ``overlap=True`` a rank launches its passes one
per launch and posts round k right behind pass k, so that the slabs of pass k travel while pass k+1 runs.

What keeps a failure from becoming a hang:
  * AGREE, THEN EXCHANGE: whether its pass launch succeeded is something every rank knows when the launch
    returns; the ranks MIN-reduce that flag (on the control group, gloo if the caller has one: CPU only)
    BEFORE any transfer is posted, so either all of them enter the exchange or none does, and every rank
    reaches that small collective whatever happened to its launch;
  * TIME-BOXED: the end of the step is an event polled against a deadline (`timeout_s`); a peer that never
    delivers raises :class:`ExchangeTimeout` instead of blocking in ``synchronize`` forever.  After a
    time-out the process group is unusable (a transfer is still pending on the device): abort it
    (:func:`abort_group`) and tear the process down.

The exchange code is backend-agnostic (torch tensors): the CPU tests run it over gloo with host tensors,
the GPU path runs it over RCCL with zero-copy views of the workspace.
"""
import time
from datetime import timedelta

import numpy as np


class ExchangeError(RuntimeError):
    """Some rank could not run its passes: no rank entered the exchange."""


class ExchangeTimeout(RuntimeError):
    """The slab exchange (or the step around it) did not complete in time on this rank."""


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


def n_rounds(NDIR, world):
    """Rounds of the exchange: the largest number of passes any rank runs."""
    return max(passes_of_rank(NDIR, world, g)[1] for g in range(world))


_auto_ctrl = {}


def control_group(dist, group=None):
    """A CPU (gloo) group over the ranks of the DEFAULT group, made once per process, for agree().  Only for
    group=None: new_group is a collective of the whole default group, which the ranks of a sub-group cannot call alone."""
    if group is not None:
        return None
    if "g" not in _auto_ctrl:
        from datetime import timedelta
        _auto_ctrl["g"] = dist.new_group(backend="gloo", timeout=timedelta(seconds=900))
    return _auto_ctrl["g"]


def agree(ok, dist, group=None, ctrl=None, device=None):
    """True iff `ok` on EVERY rank.  Every rank must call it (it is a collective): the point is that a rank whose
    own work failed still gets here, so the others learn of it instead of waiting for its slabs.

    With a CPU control group (`ctrl`, gloo) nothing touches the device.  WITHOUT one the flag is a device tensor reduced on
    `group` (RCCL) and read back with .item(): the collective is ordered behind whatever the current stream holds -- the
    pass kernel just launched -- so the host blocks until that pass has finished.  That is harmless when all passes are
    launched before the exchange, and it costs the pass-by-pass schedule (overlap=True) most of its overlap:
    aggregate_direction_sharded therefore makes itself a control group in that case (control_group)."""
    import torch
    g = ctrl if ctrl is not None else group
    on_cpu = ctrl is not None or device is None or torch.device(device).type == "cpu"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if on_cpu else device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=g)
    return bool(int(t.item()) == 1)


def abort_group(dist, group=None):
    """Best effort: abort the communicator of a wedged process group so that teardown does not block on it."""
    try:
        pg = group if group is not None else dist.distributed_c10d._get_default_group()
        for name in ("abort", "_abort"):
            f = getattr(pg, name, None)
            if f is not None:
                f()
                return True
        b = pg._get_backend(__import__("torch").device("cuda"))
        for name in ("abort", "_abort"):
            f = getattr(b, name, None)
            if f is not None:
                f()
                return True
    except Exception:  # noqa: BLE001 -- nothing more can be done for a dead communicator
        pass
    return False


class SlabExchange:
    """The ordered all-to-all of Lr row slabs, round by round.

        ex = SlabExchange(NDIR, ny, nx, L, dist, ...)
        for k in range(ex.rounds):
            ex.post(k, lr_k)        # lr_k: [ny, nx, L] volume of this rank's k-th pass, or None if it has fewer passes
        recv = ex.finish()          # [NDIR, my_rows, nx, L]: every pass's slab of this rank's rows, in pass order

    Between any two ranks the messages are posted in round order on both sides, so they match without tags."""

    def __init__(self, NDIR, ny, nx, L, dist, group=None, dtype=None, device="cpu", timeout_s=None):
        import torch
        self.dist, self.group, self.NDIR, self.ny = dist, group, NDIR, ny
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.slabs = row_slabs(ny, self.world)
        self.r0, self.nr = self.slabs[self.rank]
        self.first, self.count = passes_of_rank(NDIR, self.world, self.rank)
        self.rounds = n_rounds(NDIR, self.world)
        self.device = torch.device(device)
        self.recv = torch.empty((NDIR, self.nr, nx, L), dtype=dtype or torch.float32, device=self.device)
        self.reqs = []
        self.deadline = (time.monotonic() + timeout_s) if timeout_s else None
        self.posted = 0

    def post(self, k, vol):
        dist = self.dist
        assert k == self.posted and k < self.rounds, "rounds are posted in order"
        assert (vol is not None) == (k < self.count), "a volume for each of this rank's passes, None beyond them"
        self.posted += 1
        ops = []
        for g in range(self.world):  # receives: the k-th pass of every rank that has one
            f, n = passes_of_rank(self.NDIR, self.world, g)
            if k >= n:
                continue
            p = f + k
            if g == self.rank:
                if self.nr:
                    self.recv[p].copy_(vol[self.r0:self.r0 + self.nr])
            elif self.nr:
                ops.append(dist.P2POp(dist.irecv, self.recv[p], g, self.group))
        if vol is not None:  # sends: every other rank's rows of my k-th pass
            for g in range(self.world):
                g0, gn = self.slabs[g]
                if g != self.rank and gn:
                    ops.append(dist.P2POp(dist.isend, vol[g0:g0 + gn], g, self.group))
        if ops:
            self.reqs += dist.batch_isend_irecv(ops)

    def _remaining(self):
        if self.deadline is None:
            return None
        left = self.deadline - time.monotonic()
        if left <= 0:
            raise ExchangeTimeout("slab exchange: rank %d of %d ran out of time" % (self.rank, self.world))
        return left

    def finish(self, block=True):
        """Wait for every transfer.  Device tensors: the CURRENT stream is ordered behind the transfers; with `block`
        the host also waits, polling an event against the deadline.  Host tensors (gloo): each request is waited for
        with what is left of the deadline."""
        import torch
        assert self.posted == self.rounds, "every round must be posted (ranks without a pass post None)"
        if self.device.type == "cuda":
            for req in self.reqs:
                req.wait()  # (RCCL: enqueues a stream wait, does not block the host)
            if block:
                ev = torch.cuda.Event()
                ev.record()
                poll_event(ev, self.deadline, "slab exchange: rank %d of %d" % (self.rank, self.world))
        else:
            for req in self.reqs:
                left = self._remaining()
                try:
                    if left is None:
                        req.wait()
                    else:
                        req.wait(timedelta(seconds=left))
                except RuntimeError as e:
                    if "ime" in str(e) and "out" in str(e):  # gloo: "Timed out waiting ..."
                        raise ExchangeTimeout(str(e)) from e
                    raise
        self.reqs = []
        return self.recv


def poll_event(ev, deadline, what="step"):
    """Host-side wait for a torch.cuda.Event that gives up at `deadline` (time.monotonic() seconds; None = never)."""
    while not ev.query():
        if deadline is not None and time.monotonic() > deadline:
            raise ExchangeTimeout("%s: not complete after the time allowed" % what)
        time.sleep(0.0002)


def exchange_lr(lr_local, NDIR, ny, dist, group=None, like=None, timeout_s=None):
    """lr_local: list of [ny, nx, L] tensors, the Lr volumes of this rank's passes (in pass order).
    Returns a [NDIR, my_rows, nx, L] tensor with every pass's slab of this rank's rows.
    A rank beyond the NDIR-th runs no pass (world > NDIR) but still owns rows: it passes `like`, any tensor
    whose last two dimensions, dtype and device are those of the Lr volumes."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    first, count = passes_of_rank(NDIR, world, rank)
    assert len(lr_local) == count
    ref = lr_local[0] if count else like
    if ref is None:
        raise ValueError("exchange_lr: a rank without passes must say what the volumes look like (like=)")
    ex = SlabExchange(NDIR, ny, ref.shape[-2], ref.shape[-1], dist, group, ref.dtype, ref.device, timeout_s)
    for k in range(ex.rounds):
        ex.post(k, lr_local[k] if k < count else None)
    return ex.finish()


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


def aggregate_direction_sharded(ctx, cv, P1, P2, NDIR, MGM, use_fh, fix_overcount, refine, dist, group=None, w8=None,
                                ctrl=None, timeout_s=120.0, overlap=False, stats=None, force_sharded=False):
    """The whole aggregation of ONE volume across the ranks of `group`.  Returns (out, outcost) as torch
    tensors [ny, nx] on every rank (all-gathered rows).

    ctrl: a CPU (gloo) group of the same ranks for the agreement step (default: `group` itself, device tensors -- the host
    then waits for the launched passes in agree(); with overlap=True and the default group a gloo group is made once).
    overlap: launch this rank's passes one per launch and post each pass's slabs right behind it.
    stats: a dict that receives this rank's stage times in ms (passes / exchange / wta / gather), torch events on the
    library's stream.  Raises ExchangeError (no rank exchanged anything; the group is still usable) or ExchangeTimeout
    (the group is wedged).  force_sharded: take the sharded path also with ONE rank (tests: every stage but the transfers)."""
    import torch
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nx, ny, dmin, dmax = cv.dims
    L = dmax - dmin + 1
    if world == 1 and not force_sharded:  # nothing to shard: the plain call (which may also pad odd label counts and pick its occupancy)
        _, o, c = ctx.aggregate_dev(cv, P1, P2, NDIR, MGM, use_fh, fix_overcount, w8, refine)
        ctx.synchronize()
        views = [device_view(ctx.lib.mgm_img_device_ptr(im.h), (ny, nx)).clone() for im in (o, c)]
        o.free(), c.free()
        return views[0], views[1]
    if overlap and ctrl is None:
        # the per-round agreement must not wait for the pass it follows (agree): CPU control group, made once.  A caller
        # on a sub-group has to bring its own; without one the schedule still works, pass k+1 just starts after pass k.
        ctrl = control_group(dist, group)
    deadline = (time.monotonic() + timeout_s) if timeout_s else None
    first, count = passes_of_rank(NDIR, world, rank)
    rounds = n_rounds(NDIR, world)
    slabs = row_slabs(ny, world)
    r0, nr = slabs[rank]
    maxr = max(n for _, n in slabs)
    lib_stream = torch.cuda.ExternalStream(ctx.stream_ptr())
    marks = []

    def mark():
        if stats is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)

    def launch(fn):  # a launch error is known when the call returns: remember it, and go on to the agreement
        try:
            fn()
            return None
        except Exception as e:  # noqa: BLE001 -- reported after every rank has been told
            return e

    with torch.cuda.stream(lib_stream):
        mark()
        err = None
        ex = None
        if not overlap:
            if count:  # (world > NDIR: the ranks beyond the NDIR-th run no pass; they still sum and search their rows)
                err = launch(lambda: ctx.aggregate_passes_dev(cv, P1, P2, MGM, use_fh, first, count, w8))
            mark()
            if not agree(err is None, dist, group, ctrl, "cuda"):
                raise ExchangeError("rank %d: %s" % (rank, err) if err is not None else "another rank could not run its passes")
            ex = SlabExchange(NDIR, ny, nx, L, dist, group, torch.float32, "cuda", None)
            ex.deadline = deadline
            for k in range(rounds):
                ex.post(k, device_view(ctx.lr_device_ptr(k), (ny, nx, L)) if k < count else None)
        else:
            for k in range(rounds):
                if k < count and err is None:
                    err = launch(lambda: ctx.aggregate_passes_at_dev(cv, P1, P2, MGM, use_fh, first + k, 1, k, count, NDIR, w8))
                if k == rounds - 1:
                    mark()
                if not agree(err is None, dist, group, ctrl, "cuda"):
                    if ex is not None:  # earlier rounds are in flight: let them land before anybody frees a buffer
                        ex.posted = ex.rounds
                        ex.finish()
                    raise ExchangeError("rank %d: %s" % (rank, err) if err is not None else "another rank could not run its passes")
                if ex is None:
                    ex = SlabExchange(NDIR, ny, nx, L, dist, group, torch.float32, "cuda", None)
                    ex.deadline = deadline
                ex.post(k, device_view(ctx.lr_device_ptr(k), (ny, nx, L)) if k < count else None)
        recv = ex.finish(block=False)
        mark()
        out = torch.zeros((maxr, nx), dtype=torch.float32, device="cuda")  # (padded to the longest slab for the gather)
        outc = torch.zeros_like(out)
        if nr:
            ctx.wta_rows_dev(cv, r0, nr, recv.data_ptr(), NDIR, fix_overcount, refine, out.data_ptr(), outc.data_ptr())
        mark()
        go = [torch.empty((maxr, nx), dtype=torch.float32, device="cuda") for _ in range(world)]
        gc = [torch.empty_like(go[0]) for _ in range(world)]
        dist.all_gather(go, out, group)
        dist.all_gather(gc, outc, group)
        full_o = torch.cat([go[g][:slabs[g][1]] for g in range(world)])
        full_c = torch.cat([gc[g][:slabs[g][1]] for g in range(world)])
        mark()
        done = torch.cuda.Event()
        done.record()
        poll_event(done, deadline, "direction-sharded step: rank %d of %d" % (rank, world))
    ctx.synchronize()  # the stream is idle: this only looks at the pass kernel's watchdog word
    if stats is not None:
        names = ["passes_ms", "exchange_ms", "wta_ms", "gather_ms"]
        for i, n in enumerate(names):
            stats[n] = stats.get(n, 0.0) + marks[i].elapsed_time(marks[i + 1])
        stats["steps"] = stats.get("steps", 0) + 1
    return full_o, full_c


# ---- self-diagnosis of the first run on a real node (round 4) ----------------------------------------------------------------
def link_probe(dist, group=None, device="cuda", mbytes=256, reps=3, timeout_s=60.0):
    """Point-to-point rates of the exchange's own transport (batch_isend_irecv: grouped ncclSend/ncclRecv on RCCL, TCP on
    gloo), in the exchange's own pattern: for every distance d = 1 .. world-1 ALL ranks send `mbytes` MB to rank+d and receive
    from rank-d at the same time -- every link busy, one peer per link, as in a round of the slab exchange.
    Returns {"gbps": [[rate of rank r at distance d for d in 1..world-1] for r in ranks], "min", "max", "mbytes"} on every
    rank (the rates are all-gathered); DESIGN.md section 6 ASSUMES 153 GB/s per xGMI link -- this measures it."""
    import torch
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world < 2:
        return {"gbps": [], "min": None, "max": None, "mbytes": mbytes}
    n = mbytes * (1 << 20) // 4
    src = torch.ones(n, dtype=torch.float32, device=device)
    dst = torch.empty(n, dtype=torch.float32, device=device)
    on_gpu = torch.device(device).type == "cuda"
    mine = []
    deadline = time.monotonic() + timeout_s
    for d in range(1, world):
        to, frm = (rank + d) % world, (rank - d) % world
        to_g = dist.get_global_rank(group, to) if group is not None else to
        frm_g = dist.get_global_rank(group, frm) if group is not None else frm
        best = 0.0
        for it in range(reps + 1):  # (the first repetition sets the connection up)
            ops = [dist.P2POp(dist.isend, src, to_g, group), dist.P2POp(dist.irecv, dst, frm_g, group)]
            if on_gpu:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            works = dist.batch_isend_irecv(ops)
            if on_gpu:
                done = torch.cuda.Event()
                done.record()
                poll_event(done, deadline, "link probe, distance %d" % d)
            else:
                for w in works:
                    w.wait(timedelta(seconds=max(1.0, deadline - time.monotonic())))
            dt = time.perf_counter() - t0
            if it:
                best = max(best, mbytes * (1 << 20) / dt / 1e9)
        mine.append(best)
    t = torch.tensor(mine, dtype=torch.float64, device=device)
    allr = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(allr, t, group)
    rates = [[float(x) for x in a.cpu()] for a in allr]
    flat = [x for a in rates for x in a]
    return {"gbps": rates, "min": min(flat), "max": max(flat), "mbytes": mbytes,
            "what": "GB/s per direction and rank; all ranks send to rank+d and receive from rank-d at once (d = 1 .. world-1)"}


def peer_copy_probe(n_devices, mbytes=256):
    """The same question for the one-process transport (hipMemcpyPeerAsync behind mgm_multi): device-to-device copy rates
    i -> j for all ordered pairs, one copy at a time, and with all devices copying to their neighbour at distance d at once.
    Needs a process that sees all `n_devices` GPUs.  Returns {"pairs_gbps": [[...]], "all_at_once_gbps": [per d], ...}."""
    import torch
    n = mbytes * (1 << 20) // 4
    bufs = [(torch.ones(n, dtype=torch.float32, device="cuda:%d" % i), torch.empty(n, dtype=torch.float32, device="cuda:%d" % i))
            for i in range(n_devices)]
    pairs = [[0.0] * n_devices for _ in range(n_devices)]
    for i in range(n_devices):
        for j in range(n_devices):
            if i == j:
                continue
            for it in range(2):
                torch.cuda.synchronize(i), torch.cuda.synchronize(j)
                t0 = time.perf_counter()
                bufs[j][1].copy_(bufs[i][0], non_blocking=True)
                torch.cuda.synchronize(i), torch.cuda.synchronize(j)
                dt = time.perf_counter() - t0
            pairs[i][j] = mbytes * (1 << 20) / dt / 1e9
    at_once = []
    for d in range(1, n_devices):
        for i in range(n_devices):
            torch.cuda.synchronize(i)
        t0 = time.perf_counter()
        for i in range(n_devices):
            bufs[(i + d) % n_devices][1].copy_(bufs[i][0], non_blocking=True)
        for i in range(n_devices):
            torch.cuda.synchronize(i)
        at_once.append(mbytes * (1 << 20) / (time.perf_counter() - t0) / 1e9)
    off = [pairs[i][j] for i in range(n_devices) for j in range(n_devices) if i != j]
    return {"pairs_gbps": pairs, "all_at_once_gbps_per_link": at_once, "min": min(off) if off else None, "max": max(off) if off else None,
            "mbytes": mbytes, "what": "device-to-device copies i -> j one at a time (pairs_gbps[i][j]), and every device to its neighbour at "
                                      "distance d at once (rate of each link)"}


def sharding_model(world, NDIR, lr_volume_gb, k3_ms_of_passes, wta_ms_full, k2_ms, link_gbps):
    """DESIGN.md section 6's model of one direction-sharded step, with whatever has been MEASURED put in: k3_ms_of_passes[c] =
    pass-kernel time of a rank that runs c passes in one launch (single-GPU measurements), wta_ms_full = ordered sum + WTA
    over all rows on one GPU, k2_ms = the cost volume, link_gbps = rate of one link with all links busy.
    Each rank sends (NDIR/world) passes x (1/world) of the rows to each peer: the exchange time is that over one link."""
    c = -(-NDIR // world)
    per_link_gb = c * lr_volume_gb / world
    ex = per_link_gb / link_gbps * 1e3 if world > 1 else 0.0
    k3 = k3_ms_of_passes.get(c)
    tot = None if k3 is None else k2_ms + k3 + ex + wta_ms_full / world
    return {"world": world, "passes_per_rank": c, "gb_per_link": per_link_gb, "exchange_ms": ex, "k3_ms": k3, "wta_ms": wta_ms_full / world,
            "total_ms": tot, "link_gbps": link_gbps}
