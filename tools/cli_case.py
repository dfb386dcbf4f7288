"""Development aid: re-run one seed of tests/test_gpu_cli.py::test_cli_random_options_match_reference and show where the
outputs of mgm_amd/bin/mgm and the reference CLI differ.  python tools/cli_case.py SEED"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from mgm_amd import synth
seed = int(sys.argv[1])
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(77000 + seed)
nch = int(rng.choice([1, 3]))
nx, ny = int(rng.integers(8, 90)), int(rng.integers(6, 60))
dmin = int(rng.integers(-20, 1)); dmax = dmin + int(rng.integers(2, 40))
if rng.random() < 0.15:
    dmin = int(rng.integers(-200, 1)); dmax = dmin + int(rng.choice([62, 63, 64, 100, 126, 127, 128, 150, 191, 192, 255, 256, 300]))
u, v, gt = synth.stereo_pair(nx, ny, max(dmin, -16), min(dmax, 8) if min(dmax, 8) > max(dmin, -16) else max(dmin, -16) + 1, seed=int(rng.integers(0, 1000)), nch=nch)
np.save(tmp + "/u.npy", np.ascontiguousarray(u.transpose(1, 2, 0)) if nch > 1 else u[0])
np.save(tmp + "/v.npy", np.ascontiguousarray(v.transpose(1, 2, 0)) if nch > 1 else v[0])
fh = int(rng.integers(0, 2))
P1, P2 = [(8, 32), (2, 9), (1.5, 700), (4, 20000), (0.5, 3.25)][int(rng.integers(0, 5))]
args = ["-r", str(dmin), "-R", str(dmax), "-O", str(int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 4, 8]))), "-P1", str(P1), "-P2", str(P2),
        "-t", str(rng.choice(["ad", "sd", "census", "ncc", "btad", "btsd"])), "-p", str(rng.choice(["none", "none", "census", "sobelx", "gblur"])),
        "-s", str(rng.choice(["none", "vfit", "parabola", "cubic", "parabolaOCV"]))]
if rng.random() < 0.3: args += ["-aP2", str(rng.choice([4, 0.3])), "-aThresh", str(rng.choice([5, 12]))]
if rng.random() < 0.3: args += ["-truncDist", str(rng.choice([63, 300, 20]))]
env = dict(TSGM=str(int(rng.integers(1, 5))), TSGM_ITER=str(int(rng.choice([1, 1, 2, 3]))), TSGM_FIX_OVERCOUNT=str(int(rng.integers(0, 2))),
           USE_TRUNCATED_LINEAR_POTENTIALS=str(fh), MEDIAN=str(int(rng.choice([0, 0, 1, 2]))), TESTLRRL=str(int(rng.integers(0, 2))),
           TESTLRRL_TAU=str(rng.choice([1.0, 0.5, 2.5])), CENSUS_NCC_WIN=str(int(rng.choice([3, 5, 7]))))
win = int(env["CENSUS_NCC_WIN"])
if args[args.index("-p") + 1] == "census" and args[args.index("-t") + 1] != "census" and nch * (win * win - 1) > 24:
    env["TESTLRRL"], env["MEDIAN"], env["TSGM_ITER"] = "0", "0", "1"  # (NaN costs: as tests/test_gpu_cli.py)
lo = hi = None
if rng.random() < 0.3:
    lo = np.floor(rng.integers(dmin - 3, dmax, size=(ny, nx))).astype(np.float32) + rng.random((ny, nx)).astype(np.float32)
    hi = lo + rng.integers(0, 14, size=(ny, nx)).astype(np.float32)
    lo[rng.random((ny, nx)) < 0.02] = np.nan
    hi[rng.random((ny, nx)) < 0.02] = np.inf
    np.save(tmp + "/lo.npy", lo); np.save(tmp + "/hi.npy", hi)
    args += ["-m", tmp + "/lo.npy", "-M", tmp + "/hi.npy"]
print(nch, nx, ny, " ".join(args), env)
outs = {}
for tag, exe in (("ref", ROOT + "/oracle/_ref/mgm"), ("ours", ROOT + "/mgm_amd/bin/mgm")):
    d = tmp + "/" + tag; os.mkdir(d)
    cmd = [exe] + args + ["-l", d + "/nolr.npy", tmp + "/u.npy", tmp + "/v.npy", d + "/disp.npy", d + "/cost.npy", d + "/back.npy"]
    r = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="2", **env), capture_output=True, text=True)
    print(tag, r.returncode, r.stderr[:200])
    outs[tag] = {f: np.load(d + "/" + f) for f in sorted(os.listdir(d))}
for f in outs["ref"]:
    a, b = outs["ref"][f], outs["ours"][f]
    a, b = a.reshape(ny, nx, -1), b.reshape(ny, nx, -1)
    bad = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
    print(f, "differing words:", int(bad.sum()))
    for y, x, c in zip(*np.nonzero(bad)):
        print("   y,x,c", y, x, c, "ref", a[y, x, c], "ours", b[y, x, c], "" if lo is None else ("lo %r hi %r" % (lo[y, x], hi[y, x])),
              "cost ref/ours", outs["ref"]["cost.npy"].reshape(ny, nx)[y, x], outs["ours"]["cost.npy"].reshape(ny, nx)[y, x])
