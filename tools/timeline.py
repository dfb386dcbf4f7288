"""Development aid: turn the per-work-item time stamps of the queue kernels (k_pass2, XCDQ, built with -DMGM_P2_TIMELINE=1;
MGM_HIP_TIMELINE=<file> makes the library dump them after every pass launch) into the answer to "which resource binds a
single launch": how busy the compute units are, how long bands wait for their predecessors, how long the tail is.

    python tools/timeline.py dump.txt [--csv out.csv]

The dump (mgm_plan.hip) holds one line per work item:
    item <ticket> <pass> <band> <strip> <queue> <start_us> <end_us> <wait_us> <waits> <hw_id> <xcc_id> <steps> <spins>
(<wait_us> is 0 in -DMGM_P2_TIMELINE=1 builds -- this compiler rejects clock reads inside the poll loop of the queue kernels --
and the time waited is then estimated as spins x the poll period, fitted: duration = a*steps + b*spins over all items)
and a header line `launch <nx> <ny> <L> <nitems> <wg_per_cu> <kernel_us>`.
"""
import sys
from collections import defaultdict

import numpy as np


def load(path):
    launches = []
    cur = None
    for line in open(path):
        f = line.split()
        if not f:
            continue
        if f[0] == "launch":
            cur = {"hdr": f[1:], "items": []}
            launches.append(cur)
        elif f[0] == "item" and cur is not None:
            cur["items"].append([float(x) for x in f[1:]])
    return launches


def analyse(l, out=sys.stdout):
    hdr = l["hdr"]
    it = np.array(l["items"])
    if it.size == 0:
        print("empty launch", file=out)
        return
    pas, band, strip, queue = it[:, 1].astype(int), it[:, 2].astype(int), it[:, 3].astype(int), it[:, 4].astype(int)
    t0, t1, wait, nwait = it[:, 5], it[:, 6], it[:, 7], it[:, 8]
    hw, xcc, steps = it[:, 9].astype(np.int64), it[:, 10].astype(int), it[:, 11]
    spins = it[:, 12] if it.shape[1] > 12 else np.zeros(len(it))
    if wait.sum() == 0 and spins.sum() > 0:
        A = np.stack([steps, spins], 1)
        coef, *_ = np.linalg.lstsq(A, t1 - t0, rcond=None)
        print("  fitted: item duration = %.4f us x steps + %.4f us x polls (residual rms %.1f us)" %
              (coef[0], coef[1], float(np.sqrt(np.mean((A @ coef - (t1 - t0)) ** 2)))), file=out)
        wait = np.minimum(spins * max(coef[1], 0.0), t1 - t0)
    span = t1.max() - t0.min()
    t1 = t1 - t0.min()
    t0 = t0 - t0.min()
    # a compute unit = (xcc, se, sh, cu) of the HW_ID register (gfx9 layout: cu_id [11:8], sh_id [12], se_id [15:13])
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    cus = np.unique(cu)
    print("launch %s: %d work items on %d compute units (%d XCDs), span %.1f us" % (" ".join(hdr), len(it), len(cus), len(np.unique(xcc)), span), file=out)
    busy = np.array([np.sum((t1 - t0)[cu == c]) for c in cus])
    waitc = np.array([np.sum(wait[cu == c]) for c in cus])
    last = np.array([np.max(t1[cu == c]) for c in cus])
    first = np.array([np.min(t0[cu == c]) for c in cus])
    n_cu = max(len(cus), 1)
    tot = span * n_cu
    work = np.sum(t1 - t0) - np.sum(wait)
    print("  CU-time budget (span x CUs)                 %10.0f us  100.0 %%" % tot, file=out)
    print("  running band steps (in an item, not waiting) %9.0f us  %5.1f %%" % (work, 100 * work / tot), file=out)
    print("  waiting for the predecessor band's slabs      %8.0f us  %5.1f %%   (chain-bound share)" % (np.sum(wait), 100 * np.sum(wait) / tot), file=out)
    gaps = tot - np.sum(t1 - t0)
    tail = np.sum(span - last)
    ramp = np.sum(first)
    print("  no item on the CU: start-up ramp              %8.0f us  %5.1f %%" % (ramp, 100 * ramp / tot), file=out)
    print("  no item on the CU: tail after its last item   %8.0f us  %5.1f %%   (quantisation / imbalance)" % (tail, 100 * tail / tot), file=out)
    print("  no item on the CU: between items              %8.0f us  %5.1f %%" % (gaps - tail - ramp, 100 * (gaps - tail - ramp) / tot), file=out)
    st = np.sum(steps)
    if st > 0:
        print("  band-steps %.0f; per step while running: %.3f us; incl. waits: %.3f us; ideal at this step on %d CUs: %.1f us" %
              (st, work / st, np.sum(t1 - t0) / st, n_cu, work / n_cu), file=out)
    print("  items per CU: min %d / median %d / max %d; busy time per CU: min %.0f / median %.0f / max %.0f us" %
          (min(np.sum(cu == c) for c in cus), int(np.median([np.sum(cu == c) for c in cus])), max(np.sum(cu == c) for c in cus),
           busy.min(), np.median(busy), busy.max()), file=out)
    print("  per pass: bands, first start, last end, chain = sum over bands of (end - predecessor's end) lower bound, waits", file=out)
    for p in np.unique(pas):
        m = pas == p
        nb = band[m].max() + 1
        # the pass's critical chain as executed: the last band's end; and how much of the span bands of this pass spent waiting
        print("    pass %d: %3d bands x %d strips, active %.0f .. %.0f us, mean item %.0f us, mean wait %.0f us (%.0f %% of item), waits/item %.1f" %
              (p, nb, strip[m].max() + 1, t0[m].min(), t1[m].max(), np.mean((t1 - t0)[m]), np.mean(wait[m]),
               100 * np.sum(wait[m]) / max(np.sum((t1 - t0)[m]), 1e-9), np.mean(nwait[m])), file=out)
    # utilisation over time: share of CUs inside an item / inside an item and not (on average) waiting, in 20 slices
    nb = 20
    edges = np.linspace(0, span, nb + 1)
    print("  time slices (us): CUs with an item [%], thereof waiting [%] (item wait spread evenly over the item)", file=out)
    for k in range(nb):
        a, b = edges[k], edges[k + 1]
        ov = np.clip(np.minimum(t1, b) - np.maximum(t0, a), 0, None)
        dur = np.maximum(t1 - t0, 1e-9)
        print("    %6.0f-%6.0f  %5.1f  %5.1f" % (a, b, 100 * ov.sum() / ((b - a) * n_cu), 100 * (ov * wait / dur).sum() / max(ov.sum(), 1e-9)), file=out)


def main():
    ls = load(sys.argv[1])
    if not ls:
        print("no launches in", sys.argv[1])
        return 1
    which = ls[-1:] if "--all" not in sys.argv else ls
    for l in which:
        analyse(l)
    if "--csv" in sys.argv:
        path = sys.argv[sys.argv.index("--csv") + 1]
        with open(path, "w") as f:
            f.write("ticket,pass,band,strip,queue,start_us,end_us,wait_us,waits,hw_id,xcc_id,steps\n")
            for r in ls[-1]["items"]:
                f.write(",".join("%g" % x for x in r) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
