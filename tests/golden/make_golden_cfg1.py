"""BASELINE.json config 1, whole: the reference's own example pair (data/fountain23-im{L,R}.png, 700x500 RGB) through
the reference's own command line

    TSGM=2 mgm -r -120 -R 30 -t ad -O 4 imL imR disp cost        (Makefile-style run; 151 labels, L->R and R->L, LR check)

Runs only in the build container (needs /root/reference and oracle/_ref/mgm, the reference CLI compiled by
oracle/Makefile).  Writes tests/golden/cfg1_fountain23.npz: the two images as uint8 (what the PNGs hold), the
reference's stdout, and its disparity and cost maps -- data only.  The SHA-256 fingerprints of the two maps are checked
against the ones the survey measured on the default serial build.

    python tests/golden/make_golden_cfg1.py
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "mgm")
ARGS = "-r -120 -R 30 -t ad -O 4"
ENV = dict(TSGM="2")
SURVEY_SHA16 = {"disp": "d19abe7251c39b0e", "cost": "cf19d4580c1608b6"}


def main():
    from PIL import Image
    ims = {s: np.asarray(Image.open("/root/reference/data/fountain23-im%s.png" % s))[:, :, :3].copy() for s in "LR"}
    assert ims["L"].dtype == np.uint8 and ims["L"].shape == (500, 700, 3)
    with tempfile.TemporaryDirectory() as d:
        for s in "LR":
            np.save(os.path.join(d, s + ".npy"), ims[s].astype(np.float32))  # iio reads (H, W, C) float32 .npy natively
        cmd = [REF] + ARGS.split() + [os.path.join(d, "L.npy"), os.path.join(d, "R.npy"), os.path.join(d, "disp.npy"), os.path.join(d, "cost.npy")]
        r = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="8", **ENV), capture_output=True, text=True, check=True)
        disp, cost = np.load(os.path.join(d, "disp.npy")), np.load(os.path.join(d, "cost.npy"))
    disp, cost = disp.reshape(500, 700), cost.reshape(500, 700)
    for name, a in (("disp", disp), ("cost", cost)):
        sha = hashlib.sha256(np.ascontiguousarray(a, "<f4").tobytes()).hexdigest()[:16]
        print(name, sha, "(survey: %s)" % SURVEY_SHA16[name], "NaN:", int(np.isnan(a).sum()))
        assert sha == SURVEY_SHA16[name], "the reference build here does not reproduce the survey's fingerprint"
    np.savez_compressed(os.path.join(HERE, "cfg1_fountain23.npz"), uL=ims["L"], uR=ims["R"], disp=disp, cost=cost,
                        stdout=np.array(r.stdout), args=np.array(ARGS), env=np.array(repr(ENV)))
    print("wrote cfg1_fountain23.npz", os.path.getsize(os.path.join(HERE, "cfg1_fountain23.npz")), "bytes")


if __name__ == "__main__":
    sys.exit(main())
