import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import mgm_amd
from mgm_amd import synth
seed = 18
rng = np.random.default_rng(9000 + seed)
nb = int(rng.integers(1, 17))
nx, ny = int(rng.integers(1, 40)), int(rng.integers(1, 30))
L = int(rng.choice([5, 31, 64, 65, 100, 127, 128, 129, 192, 200, 256, 300, 384, 512]))
if nb * nx * ny * L > 400000:
    L = int(rng.choice([31, 64, 100, 128])); nb = min(nb, 4)
NDIR = int(rng.integers(1, 9)); MGM = int(rng.integers(1, 5)); FH = int(rng.integers(0, 2))
P1, P2 = [(8.0, 32.0), (2.0, 9.0), (1.5, 20000.0), (2.0, np.inf), (0.5, 3.25)][int(rng.integers(0, 5))]
fix = int(rng.integers(0, 2)); dmin = int(rng.integers(-300, 300))
allint = rng.random() < 0.7; weighted = rng.random() < 0.25
refine = [None, "vfit", "cubic"][int(rng.integers(0, 3))]
Cs = []
for b in range(nb):
    fr = float(rng.choice([0.0, 0.05, 0.4]))
    C = synth.raw_volume(nx, ny, L, seed=seed * 17 + b, inf_frac=fr)
    third = (not allint) and rng.random() < 0.5
    if third: C = (C * np.float32(1.0 / 3.0)).astype(np.float32)
    print("vol", b, "inf_frac", fr, "third", third)
    Cs.append(C)
def run(sel, P2=P2, NDIR=NDIR):
    ctx = mgm_amd.Context(0)
    cvs = [ctx.upload_volume(Cs[b], dmin) for b in sel]
    try:
        ctx.aggregate_batch_dev(cvs, P1, P2, NDIR, MGM, FH, fix, None, None)
        ctx.synchronize()
        print(sel, P2, NDIR, "ok", flush=True)
    except mgm_amd.MgmError as e:
        print(sel, P2, NDIR, "FAILED", e, flush=True)
    ctx.close()
run(list(range(nb)))
for b in range(nb): run([b])
run(list(range(nb)), P2=32.0)
for q in (1,2,3): run(list(range(nb)), NDIR=q)
