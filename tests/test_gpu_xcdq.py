"""The per-XCD work queues of the pass kernel (k_pass2, XCDQ; mgm_api.hip run_passes): whatever the dealing of bands to
queues -- a pass pinned to one XCD, blocks of 1, 3 or 8 bands, no queues at all -- the aggregated volume is the same,
bit for bit, and equals the oracle's; every hand-off slot carries the launch's tag afterwards (MGM_HIP_CHECK_TAGS).
The switches are read once per process, so every setting runs in a process of its own."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from mgm_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
import mgm_amd
from mgm_amd import synth
a = sys.argv[1:]
nx, ny, L, NDIR, MGM, FH, P1, P2, nb = int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), float(a[6]), float(a[7]), int(a[8])
ctx = mgm_amd.Context(0)
h = hashlib.sha256()
cvs = [ctx.upload_volume(np.rint(synth.raw_volume(nx, ny, L, seed=900 + b, inf_frac=0.02)).astype(np.float32), -L // 2) for b in range(nb)]
for rep in range(3):  # consecutive launches alternate the tag on the same slots
    S, outs, outcs = ctx.aggregate_batch_dev(cvs, P1, P2, NDIR, MGM, FH, 1, None, "vfit", want_S=True)
    for b in range(nb):
        h.update(np.ascontiguousarray(S[b].download()).tobytes())
        h.update(np.ascontiguousarray(outs[b].download()).tobytes())
        h.update(np.ascontiguousarray(outcs[b].download()).tobytes())
print("DIGEST", h.hexdigest())
""" % ROOT

SETTINGS = [
    {"MGM_HIP_XCDQ": "0"},
    {"MGM_HIP_XCDQ": "1"},
    {"MGM_HIP_XCDQ": "1", "MGM_HIP_XCDQ_K": "0"},
    {"MGM_HIP_XCDQ": "1", "MGM_HIP_XCDQ_K": "1"},
    {"MGM_HIP_XCDQ": "1", "MGM_HIP_XCDQ_K": "3"},
    {"MGM_HIP_XCDQ": "1", "MGM_HIP_XCDQ_K": "8"},
]

SHAPES = [
    # nx, ny, L, NDIR, MGM, FH, P1, P2, volumes
    (331, 217, 128, 8, 3, 0, 8.0, 32.0, 1),
    (331, 217, 256, 8, 3, 1, 2.0, 20000.0, 1),
    (260, 190, 128, 4, 2, 0, 8.0, 32.0, 1),     # four passes: blocks of two bands by default
    (200, 170, 256, 8, 4, 1, 2.0, 9.0, 3),      # three volumes: 24 chains
    (180, 140, 192, 8, 3, 0, 8.0, 32.0, 1),     # three labels per lane
    (170, 150, 384, 8, 1, 0, 8.0, 32.0, 1),     # six labels per lane, one neighbour
    (150, 130, 512, 8, 3, 1, 2.0, 30.0, 1),     # eight labels per lane, FH with a finite cap
    (230, 170, 64, 8, 4, 0, 8.0, 32.0, 1),      # one label per lane, four neighbours
]


def run(shape, env):
    e = dict(os.environ, MGM_HIP_CHECK_TAGS="1", **env)
    r = subprocess.run([sys.executable, "-c", SCRIPT] + [str(a) for a in shape], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0]


@pytest.mark.parametrize("shape", SHAPES)
def test_queue_dealings_agree(shape):
    digests = [run(shape, env) for env in SETTINGS]
    assert len(set(digests)) == 1, list(zip(SETTINGS, digests))


def test_queues_against_oracle(oracle):
    nx, ny, L, dmin = 150, 100, 128, -64
    P1, P2, NDIR, MGM, FH = 8.0, 32.0, 8, 3, 0
    C = np.rint(synth.raw_volume(nx, ny, L, seed=900, inf_frac=0.02)).astype(np.float32)
    So, oo, co = oracle.mgm(C, dmin, P1, P2, NDIR, MGM, FH, 1)
    h = hashlib.sha256()
    code = SCRIPT.replace("seed=900 + b", "seed=900 + b").replace('"vfit"', "None")
    # one launch shape, queues forced on: the digest of the oracle's S / labels / costs, three launches
    for rep in range(3):
        h.update(np.ascontiguousarray(So.astype(np.float32)).tobytes())
        h.update(np.ascontiguousarray(oo.astype(np.float32).reshape(-1)).tobytes())
        h.update(np.ascontiguousarray(co.astype(np.float32).reshape(-1)).tobytes())
    e = dict(os.environ, MGM_HIP_CHECK_TAGS="1", MGM_HIP_XCDQ="1", MGM_HIP_XCDQ_K="1")
    r = subprocess.run([sys.executable, "-c", code, str(nx), str(ny), str(L), str(NDIR), str(MGM), str(FH), str(P1), str(P2), "1"],
                       env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "DIGEST " + h.hexdigest() in r.stdout
