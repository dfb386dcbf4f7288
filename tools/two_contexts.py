"""Development aid (GPU box): what would overlapping the k_wta of one pair with the pass kernel of the next be worth?  Two
CONTEXTS on one device (each its own stream and workspace), each running single-pair steps of a workload: one after the
other, then at the same time from two threads.  python tools/two_contexts.py cfg3 [steps]"""
import os, sys, threading, time
import torch
torch.cuda.init()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mgm_amd
import bench

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nx, ny = w["nx"], w["ny"]


class Lane:
    def __init__(self, seed):
        self.ctx = mgm_amd.Context(0)
        u, v, _ = bench.pair_of(w, seed)
        self.du, self.dv = self.ctx.upload_image(u), self.ctx.upload_image(v)
        self.out, self.outc = self.ctx.new_image(nx, ny), self.ctx.new_image(nx, ny)
        self.cv = None

    def run(self, n):
        c = self.ctx
        for _ in range(n):
            self.cv = c.costvolume_dev(self.du, self.dv, w["dmin"], w["dmax"], "none", "census", float("inf"), w["win"], into=self.cv)
            c.aggregate_batch_dev([self.cv], w["P1"], w["P2"], w["NDIR"], w["MGM"], w["FH"], 1, None, "vfit", [self.out], [self.outc], want_S=False)
        c.synchronize()


a, b = Lane(0), Lane(1)
a.run(3); b.run(3)
for rep in range(3):
    t0 = time.perf_counter(); a.run(steps); t1 = time.perf_counter(); b.run(steps); t2 = time.perf_counter()
    ta, tb = threading.Thread(target=a.run, args=(steps,)), threading.Thread(target=b.run, args=(steps,))
    t3 = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join(); t4 = time.perf_counter()
    print("one context: %.1f / %.1f volumes/s; two at once: %.1f volumes/s" % (steps / (t1 - t0), steps / (t2 - t1), 2 * steps / (t4 - t3)), flush=True)
