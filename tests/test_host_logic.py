"""Host-side plumbing that needs no device: synthetic inputs, the exact x/3 sequence
(checked on the CPU with C fmaf over a strided sample of all floats), CPU counting."""
import ctypes

import numpy as np

from mgm_amd import synth
from oracle.oracle import usable_cpus


def test_synth_is_deterministic_and_in_range():
    a = synth.stereo_pair(64, 40, -12, 3)
    b = synth.stereo_pair(64, 40, -12, 3)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    u, v, gt = a
    assert u.dtype == np.float32 and u.shape == (1, 40, 64) and gt.min() >= -12 and gt.max() <= 3
    assert np.array_equal(u, np.rint(u)) and u.min() >= 0 and u.max() <= 255
    C = synth.raw_volume(20, 10, 16, inf_frac=0.2)
    assert np.isfinite(C).any(axis=2).all()


def test_div3_sequence_matches_ieee_division(oracle):
    """q = fma(fma(-3, x*c, x), c, x*c), c = RN(1/3): equal to x/3 for every finite float
    but -0 (the kernels finish with v_div_fixup_f32, which repairs -0 and +-inf)."""
    f = oracle.lib.orc_check_div3
    f.restype = ctypes.c_ulonglong
    f.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_ulonglong]
    assert f(0, 1 << 32, 61) == 0            # strided sample of all bit patterns
    assert f(0x3f000000, 0x3f000000 + (1 << 22), 1) == 0   # one binade exhaustively
    assert f(0, 1 << 16, 1) == 0             # subnormals near zero


def test_usable_cpus():
    assert 1 <= usable_cpus() <= 64


def test_multi_plan_partitions_passes_and_rows():
    """mgm_multi_plan (C ABI, no device needed): contiguous blocks covering every pass and row exactly once, sizes
    differing by at most one -- and the same partition as the Python launcher's (mgm_amd/dist.py)."""
    import mgm_amd
    from mgm_amd import dist as mdist
    for n in (1, 2, 3, 4, 8):
        for NDIR in (1, 3, 4, 8):
            for ny in (5, 13, 1080, 4096):
                plan = mgm_amd.multi_plan(n, NDIR, ny)
                assert [p for f, c, _, _ in plan for p in range(f, f + c)] == list(range(NDIR))
                assert [r for _, _, r0, nr in plan for r in range(r0, r0 + nr)] == list(range(ny))
                assert max(c for _, c, _, _ in plan) - min(c for _, c, _, _ in plan) <= 1
                assert [(f, c) for f, c, _, _ in plan] == [mdist.passes_of_rank(NDIR, n, k) for k in range(n)]
                assert [(r0, nr) for _, _, r0, nr in plan] == mdist.row_slabs(ny, n)


def test_kernel_hash_ignores_comments_only(tmp_path):
    """bench.kernel_source_hash (the guard of profiles/*_traffic.json): comments and line breaks do not count, code does."""
    import bench
    assert bench.strip_comments('a = "x//y"; // c\n/* k */ b = \'"\';\n\n  f(1 /* one */, 2);') == 'a = "x//y"; b = \'"\'; f(1 , 2);'
    d = tmp_path / "csrc"
    d.mkdir()
    (d / "k.hip").write_text("// one\n__global__ void k(int *p) { *p = 1; }  /* tail */\n")
    (d / "notes.txt").write_text("not a source")
    h0 = bench.kernel_source_hash(str(d))
    (d / "k.hip").write_text("// another comment\n\n__global__ void k(int *p)\n{\n    *p = 1;  // set\n}\n")
    assert bench.kernel_source_hash(str(d)) == h0
    (d / "k.hip").write_text("__global__ void k(int *p) { *p = 2; }\n")
    assert bench.kernel_source_hash(str(d)) != h0


def test_roofline_of_range_proportional_launch_prices_owned_cells():
    """bench.roofline_of: a k_pass_rel launch is priced on the labels the pixels OWN (55 of a hull of 256 for cfg3r), so its
    fraction cannot pass 1 where the hull-priced figure would; a k_pass2 launch of the same workload is priced on the hull."""
    import bench
    w = bench.WORKLOADS["cfg3r"]
    own = (2 * w["ragged"] + 1) / float(bench.labels_of(w))
    rel = bench.roofline_of(w, 4, {"k_cost": 0.16, "k_pass_rel": 8.0, "k_wta": 0.76}, "cfg3r")
    hull = bench.roofline_of(w, 4, {"k_cost": 1.2, "k_pass2": 8.0, "k_wta": 0.76}, "cfg3r")
    assert abs(rel["existing_cells_over_hull_cells"] - own) < 1e-12
    assert abs(rel["frac"] - rel["frac_range_proportional"]) < 1e-12
    assert abs(rel["frac_dense_hull_equivalent"] * own - rel["frac"]) < 1e-9
    assert rel["frac"] < 1.0 < rel["frac_dense_hull_equivalent"]
    assert abs(hull["frac"] - hull["frac_dense_hull_equivalent"]) < 1e-12 and hull["frac_range_proportional"] < hull["frac"]
    assert rel["algorithmic_bytes_per_launch"] == 12.0 * w["NDIR"] * w["nx"] * w["ny"] * (2 * w["ragged"] + 1) * 4
