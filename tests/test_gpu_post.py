"""mgm_post.hip (median, left-right check, range update) on exactly the maps the NaN-faithful path produces: NaN labels
where a pixel had no finite S, +INF costs, isolated and in patches.  The command-line tests cannot cover this -- the
reference's own label there is uninitialised memory (mgm_core.cc:594) -- so the kernels are compared on DEFINED inputs with
the numpy restatement (oracle/post.py) and, where it travelled, with the compiled reference itself
(oracle/_ref/libmgm_refpost.so: median_filter img_tools.h:203-238, leftright_test mgm.cc:68-91, update_dmin_dmax 120-158)."""
import numpy as np
import pytest

from helpers import ndiff
from test_oracle_vs_ref import nan_label_map

pytestmark = pytest.mark.gpu


def refpost():
    from oracle.oracle import RefPost
    return RefPost() if RefPost.available() else None


@pytest.mark.parametrize("radius", [1, 2, 3, 7, 8, 9])
def test_median_with_nan_labels(ctx, radius):
    from oracle import post
    rp = refpost()
    for seed in range(3):
        nch = 1 + seed % 2
        m = np.stack([nan_label_map(37, 45, 300 * radius + 10 * seed + c, frac_inf=0.03 if seed else 0.0) for c in range(nch)])
        d = ctx.upload_image(m)
        got = ctx.median_dev(d, radius)
        g = got.download()
        assert ndiff(g, post.median(m, radius)) == 0, (radius, seed)
        if rp is not None:
            assert ndiff(g, rp.median(m, radius)) == 0, (radius, seed, "reference")
        d.free(), got.free()


def test_median_of_an_all_nan_map_and_of_nan_costs(ctx):
    from oracle import post
    m = np.full((1, 16, 20), np.nan, np.float32)
    m[0, 3, 4] = 5.0
    for r in (1, 8):
        d = ctx.upload_image(m)
        got = ctx.median_dev(d, r)
        assert ndiff(got.download(), post.median(m, r)) == 0
        d.free(), got.free()


def test_median_window_cap(ctx):
    """Beyond the bounded work of the radix-selection kernel the call is refused, not left to run for hours."""
    import mgm_amd
    d = ctx.new_image(1920, 1080)
    o = ctx.new_image(1920, 1080)
    with pytest.raises(mgm_amd.MgmError) as e:
        ctx.median_dev(d, 1024, out=o)
    assert e.value.code == mgm_amd.MGM_ERR_UNSUPPORTED
    d.free(), o.free()


def test_leftright_with_nan_labels(ctx):
    from oracle import post
    rp = refpost()
    for seed, (tau, rnx) in enumerate([(1.0, 45), (0.5, 45), (2.0, 39), (1.0, 60), (0.0, 45)]):
        for integer in (False, True):
            d = nan_label_map(29, 45, 9000 + seed, frac_inf=0.02, subpixel=not integer)
            o = -nan_label_map(29, rnx, 9100 + seed, frac_inf=0.02)
            if integer:
                o = np.rint(o)
            dd, do = ctx.upload_image(d[None]), ctx.upload_image(o[None])
            got = ctx.leftright_dev(dd, do, tau)
            g = got.download()[0]
            assert ndiff(g, post.leftright(d, o, tau)) == 0, (seed, integer)
            if rp is not None:
                assert ndiff(g, rp.leftright(d, o, tau)) == 0, (seed, integer, "reference")
            for h in (dd, do, got):
                h.free()


def test_update_ranges_with_nan_labels(ctx):
    """update_dmin_dmax + remove_nonfinite_values_Img (mgm.cc:120-158, 386-388) on a map with NaN labels, vs the reference."""
    rp = refpost()
    if rp is None:
        pytest.skip("oracle/_ref/libmgm_refpost.so did not travel to this box")
    for seed in range(4):
        m = nan_label_map(33, 41, 9500 + seed, frac_inf=0.02 if seed % 2 else 0.0)
        lo = np.full((33, 41), -30.0, np.float32)
        hi = np.full((33, 41), 12.0, np.float32)
        want_lo, want_hi = rp.update_ranges(m, lo, hi, 3, 2)
        dm, dl, dh = ctx.upload_image(m[None]), ctx.upload_image(lo[None]), ctx.upload_image(hi[None])
        ctx.update_ranges_dev(dm, dl, dh, 3, 2)
        assert ndiff(dl.download()[0], want_lo) == 0 and ndiff(dh.download()[0], want_hi) == 0, seed
        for h in (dm, dl, dh):
            h.free()
