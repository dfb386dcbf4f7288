"""Development aid (GPU box): K3 of every single pass of a workload, and of all of them in one launch, under the current
environment (MGM_HIP_XCDQ, MGM_HIP_XCDQ_K, MGM_HIP_STRIPS ...).  python tools/time_each_pass.py cfg3"""
import os, sys
import torch
torch.cuda.init()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mgm_amd
import bench

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
ctx = mgm_amd.Context(0)
u, v, _ = bench.pair_of(w)
cv = ctx.costvolume_dev(ctx.upload_image(u), ctx.upload_image(v), w["dmin"], w["dmax"], "none", "census", float("inf"), w["win"])

def k3(first, count, reps=5):
    best = 1e9
    for _ in range(reps):
        ctx.synchronize(); ctx.timing(True); ctx.timing_reset()
        ctx.aggregate_passes_dev(cv, w["P1"], w["P2"], w["MGM"], w["FH"], first, count)
        ctx.synchronize()
        best = min(best, sum(ms for n, ms in ctx.timings() if n.startswith("k_pass")))
        ctx.timing(False)
    return best

print(" ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("MGM_HIP_")) or "(defaults)")
print("single passes:", " ".join("%.2f" % k3(q, 1) for q in range(w["NDIR"])), "| pairs:", " ".join("%.2f" % k3(q, 2) for q in range(0, w["NDIR"], 2)), "| all: %.2f ms" % k3(0, w["NDIR"]))
