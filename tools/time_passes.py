"""Development aid (runs on the GPU box): the single-GPU ingredients of the direction-sharding model of DESIGN.md section 6,
measured at BASELINE config 4 (4096x4096x192, census 5x5, 8 directions, TSGM 3): K3 for every block of passes a rank of
an n-GPU split runs (n = 1, 2, 4, 8) -- in one launch, and one launch per pass (the overlapped schedule) --, and the row-slab
WTA on ny/n rows.  Only the xGMI rate of the table is an assumption."""
import os, sys, time
import numpy as np
import torch
torch.cuda.init()  # (torch's HIP runtime first: the library then shares it -- the other order leaves torch without devices)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mgm_amd
from mgm_amd import synth, dist as mdist
import bench

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"]
ctx = mgm_amd.Context(0)
u, v, _ = bench.pair_of(w)
du, dv = ctx.upload_image(u), ctx.upload_image(v)
cv = ctx.costvolume_dev(du, dv, w["dmin"], w["dmax"], "none", "census", float("inf"), w["win"])
NDIR, ny, nx, L = w["NDIR"], w["ny"], w["nx"], bench.labels_of(w)

def k3(first, count, per_pass=False, reps=3):
    best = 1e9
    for _ in range(reps):
        ctx.synchronize()
        ctx.timing(True); ctx.timing_reset()
        if per_pass:
            for k in range(count):
                ctx.aggregate_passes_at_dev(cv, w["P1"], w["P2"], w["MGM"], w["FH"], first + k, 1, k, count, NDIR)
        else:
            ctx.aggregate_passes_dev(cv, w["P1"], w["P2"], w["MGM"], w["FH"], first, count)
        ctx.synchronize()
        t = sum(ms for n, ms in ctx.timings() if n.startswith("k_pass"))
        ctx.timing(False)
        best = min(best, t)
    return best

for n in (1, 2, 4, 8):
    blocks = [mdist.passes_of_rank(NDIR, n, r) for r in range(n)]
    one = [k3(f, c) for f, c in blocks]
    per = [k3(f, c, True) for f, c in blocks] if NDIR // n > 1 else one
    print("n=%d  K3 per rank, one launch: max %.2f ms (%s) | one launch per pass: max %.2f ms (%s)" % (
        n, max(one), " ".join("%.1f" % t for t in one), max(per), " ".join("%.1f" % t for t in per)), flush=True)
# row-slab WTA: all passes of ny/n rows (the slabs are read from the workspace of an 8-pass run: same traffic)
ctx.aggregate_passes_dev(cv, w["P1"], w["P2"], w["MGM"], w["FH"], 0, NDIR)
ctx.synchronize()
for n in (1, 2, 4, 8):
    nr = ny // n
    recv = torch.empty((NDIR, nr, nx, L), dtype=torch.float32, device="cuda")
    for p in range(NDIR):
        recv[p].copy_(mdist.device_view(ctx.lr_device_ptr(p), (ny, nx, L))[:nr])
    out = torch.empty((nr, nx), dtype=torch.float32, device="cuda"); outc = torch.empty_like(out)
    torch.cuda.synchronize()
    ctx.timing(True); ctx.timing_reset()
    for _ in range(3):
        ctx.wta_rows_dev(cv, 0, nr, recv.data_ptr(), NDIR, 1, "vfit", out.data_ptr(), outc.data_ptr())
    ctx.synchronize()
    t = min(ms for nme, ms in ctx.timings() if nme == "k_wta")
    ctx.timing(False)
    print("n=%d  k_wta on %d rows: %.2f ms; slabs received per rank: %.2f GB, per link: %.2f GB" % (
        n, nr, t, (NDIR - NDIR // n) * nr * nx * L * 4 / 1e9 if n > 1 else 0, (NDIR // n) * nr * nx * L * 4 / 1e9 if n > 1 else 0), flush=True)
    del recv
ctx.close()
