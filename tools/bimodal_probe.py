"""Development aid (GPU box): is the 15.8 / 18.6 ms bimodality of the 16-volume 128-label launch a property of the PROCESS
(where the workspace landed) or of the launch?  Re-allocates the workspace several times in one process and prints the
pass-kernel time of each allocation next to the workspace's address."""
import sys
import numpy as np
sys.path.insert(0, ".")
import bench
import mgm_amd

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
import os
with mgm_amd.Context(0) as c:
    c.set_placement_tries(int(os.environ.get("PLACE_TRIES", "0")))  # mgm_ctx_set_placement_tries: keep the fastest of n placements
    ims = []
    for b in range(B):
        u, v, _ = bench.pair_of(w, b)
        ims.append((c.upload_image(u), c.upload_image(v)))
    cvs = [c.costvolume_dev(a, b_, w["dmin"], w["dmax"], "none", "census", float("inf"), w["win"]) for a, b_ in ims]
    outs = outcs = None
    for rep in range(8):
        c.timing(False)
        for it in range(2):
            _, outs, outcs = c.aggregate_batch_dev(cvs, w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, None, "vfit", outs, outcs)
        c.synchronize()
        c.timing(True)
        c.timing_reset()
        for it in range(6):
            _, outs, outcs = c.aggregate_batch_dev(cvs, w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, None, "vfit", outs, outcs)
        c.synchronize()
        t = [ms for n, ms in c.timings() if n.startswith("k_pass")]
        probes = [c.probe_workspace(n) for n in (1, 4, 8, 16)]
        print("allocation %d: lr at 0x%x  K3 min %.2f  median %.2f  max %.2f ms   store-pattern probe (1/4/8/16 streams) %s GB/s" %
              (rep, c.lr_device_ptr(0) or 0, min(t), float(np.median(t)), max(t), " ".join("%.0f" % g for g in probes)), flush=True)
        ballast = c.new_image(1024, 1024 * (1 + rep * 3))  # shift where the next allocation lands
        c.trim()
