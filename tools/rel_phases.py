"""Development aid: sums the `phases` lines of a MGM_HIP_TIMELINE dump of the range-proportional kernels (a
-DMGM_REL_PHASES=1 build: RELDEFS=-DMGM_REL_PHASES=1 bash tools/sweep_build.sh relph "") -- clocks per step each wave spent
computing / publishing / at the step barrier, and the loader issuing / waiting for its DMAs / at the barrier.
    python tools/rel_phases.py dump.txt"""
import sys

import numpy as np

items, ph = {}, {}
for line in open(sys.argv[1]):
    f = line.split()
    if not f:
        continue
    if f[0] == "launch":
        items, ph = {}, {}  # (the last launch of the file)
    elif f[0] == "item":
        items[int(f[1])] = float(f[12])  # steps
    elif f[0] == "phases":
        ph[int(f[1])] = [float(x) for x in f[2:]]
steps = sum(items[k] for k in ph)
a = np.array([ph[k] for k in sorted(ph)])
tot = a.sum(0) / steps
names = ["compute", "publish", "barrier"]
print("clocks per step (s_memtime ticks), mean over %d work items, %d steps each" % (len(ph), int(steps / max(len(ph), 1))))
for w in range(4):
    print("  compute wave %d: " % w + ", ".join("%s %.0f" % (names[k], tot[3 * w + k]) for k in range(3)) + "  | sum %.0f" % tot[3 * w:3 * w + 3].sum())
if a.shape[1] > 15 and tot[15] > 0:
    print("  FH: sweeps of the scans per step in wave 0: %.2f (6 = three min-convolutions whose first guess holds)" % tot[15])
print("  loader wave:    issue %.0f, wait for DMA %.0f, barrier %.0f  | sum %.0f" % (tot[12], tot[13], tot[14], tot[12:15].sum()))
