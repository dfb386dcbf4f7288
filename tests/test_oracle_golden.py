"""The CPU oracle (oracle/mgm_oracle.c) against the golden vectors produced by the
compiled reference (tests/golden/make_golden.py).  Bit-exact."""
import numpy as np
import pytest

from helpers import golden_cases, labels_equal, load_golden, ndiff


@pytest.mark.parametrize("name", golden_cases("cv"))
def test_costvolume(oracle, name):
    g = load_golden(name)
    C = oracle.costvolume(g["u"], g["v"], int(g["dmin"]), int(g["dmax"]), str(g["prefilter"]), str(g["distance"]),
                          float(g["truncDist"]), int(g["census_win"]))
    assert ndiff(C, g["C"]) == 0


@pytest.mark.parametrize("name", golden_cases("weights"))
def test_weights(oracle, name):
    g = load_golden(name)
    assert ndiff(oracle.weights(g["u"], float(g["aP"]), float(g["aThresh"])), g["w8"]) == 0


@pytest.mark.parametrize("name", golden_cases("agg"))
def test_aggregation_and_refinement(oracle, name):
    g = load_golden(name)
    S, out, outc = oracle.mgm(g["C"], int(g["dmin"]), float(g["P1"]), float(g["P2"]), int(g["NDIR"]), int(g["MGM"]),
                              int(g["FH"]), int(g["FIX"]), g.get("w8"))
    assert ndiff(S, g["S"]) == 0
    assert ndiff(outc, g["outcost"]) == 0
    assert labels_equal(out, g["out"], outc)
    for meth in ("vfit", "parabola", "cubic", "parabolaOCV"):
        ro, rc = oracle.refine(g["S"], int(g["dmin"]), meth, g["out"], g["outcost"])
        assert ndiff(ro, g["out_" + meth]) == 0, meth
        assert ndiff(rc, g["outcost_" + meth]) == 0, meth


def test_name_tables_fall_back_silently(oracle):
    # mgm_costvolume.h:184-190, 201-207; mgm_refine.h:28-35: unknown names select entry 0
    L = oracle.lib
    assert L.orc_distance_index(b"census") == 2 and L.orc_distance_index(b"nope") == 0
    assert L.orc_prefilter_index(b"sobelx") == 2 and L.orc_prefilter_index(b"sobel_x") == 0
    assert L.orc_refinement_index(b"vfit") == 1 and L.orc_refinement_index(b"bogus") == 0


def test_threads_do_not_change_results(oracle):
    from mgm_amd import synth
    C = synth.raw_volume(40, 30, 16, inf_frac=0.02)
    a = oracle.mgm(C, 0, 8.0, 32.0, 8, 3)
    oracle.set_threads(4)
    b = oracle.mgm(C, 0, 8.0, 32.0, 8, 3)
    oracle.set_threads(1)
    assert ndiff(a[0], b[0]) == 0 and np.array_equal(a[1], b[1])
