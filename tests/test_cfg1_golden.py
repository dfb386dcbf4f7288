"""BASELINE.json config 1 whole -- the reference's own data/fountain23 pair (700x500 RGB, 151 labels) through
`TSGM=2 mgm -r -120 -R 30 -t ad -O 4` -- against the golden maps the REFERENCE'S OWN command line produced
(tests/golden/make_golden_cfg1.py; SHA-256 fingerprints equal to SURVEY.md section 4).

CPU: the oracle restatement (cost volume + mgm() for both runs) plus a numpy restatement of main()'s left-right check.
GPU: the `mgm` host program over libmgm_hip.so on the same files, stdout and both maps bit for bit."""
import ast
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ndiff

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = os.path.join(ROOT, "mgm_amd", "bin", "mgm")


@pytest.fixture(scope="module")
def cfg1():
    d = np.load(os.path.join(GOLDEN, "cfg1_fountain23.npz"))
    return {k: d[k] for k in d.files}


def leftright(d, other, tau):
    """leftright_test (mgm.cc:68-91): keep d[x,y] if Lx = round(x + d) lies inside `other` and |Lx + other[Lx,y] - x| <= tau."""
    ny, nx = d.shape
    x = np.arange(nx, dtype=np.float32)[None, :].repeat(ny, 0)
    with np.errstate(invalid="ignore"):
        Lx = np.round(x + d)  # (round half away from zero in C; x + d is an integer here: -s none)
        ok = (Lx >= 0) & (Lx < other.shape[1])
        Li = np.where(ok, Lx, 0).astype(np.int64)
        back = np.take_along_axis(other, Li, axis=1)
        ok &= np.abs(Lx + back - x) <= tau
    return np.where(ok, d, np.float32(np.nan)).astype(np.float32)


def test_oracle_reproduces_config1(oracle, cfg1):
    uL = np.ascontiguousarray(cfg1["uL"].astype(np.float32).transpose(2, 0, 1))
    uR = np.ascontiguousarray(cfg1["uR"].astype(np.float32).transpose(2, 0, 1))
    oracle.set_threads(min(8, len(os.sched_getaffinity(0))))
    try:
        P1, P2 = 8.0 * 3, 32.0 * 3  # main() scales the penalties by the channel count (mgm.cc:356-357)
        CL = oracle.costvolume(uL, uR, -120, 30, "none", "ad", np.inf, 3)
        _, oL, cL = oracle.mgm(CL, -120, P1, P2, 4, 2, 0, 1)
        CR = oracle.costvolume(uR, uL, -30, 120, "none", "ad", np.inf, 3)  # the right-to-left run (mgm.cc:404-414)
        _, oR, _ = oracle.mgm(CR, -30, P1, P2, 4, 2, 0, 1)
    finally:
        oracle.set_threads(1)
    disp = leftright(oL, oR, 1.0)
    assert ndiff(cL, cfg1["cost"]) == 0
    assert ndiff(disp, cfg1["disp"]) == 0
    assert int(np.isnan(disp).sum()) == 45588


@pytest.mark.gpu
def test_cli_reproduces_config1(cfg1, tmp_path):
    for s in "LR":
        np.save(tmp_path / (s + ".npy"), cfg1["u" + s].astype(np.float32))
    cmd = [OURS] + str(cfg1["args"]).split() + [str(tmp_path / "L.npy"), str(tmp_path / "R.npy"), str(tmp_path / "disp.npy"), str(tmp_path / "cost.npy")]
    r = subprocess.run(cmd, env=dict(os.environ, **ast.literal_eval(str(cfg1["env"]))), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert r.stdout == str(cfg1["stdout"])
    assert ndiff(np.load(tmp_path / "disp.npy").reshape(500, 700), cfg1["disp"]) == 0
    assert ndiff(np.load(tmp_path / "cost.npy").reshape(500, 700), cfg1["cost"]) == 0
