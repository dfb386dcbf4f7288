"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs oracle/_ref, i.e. /root/reference).  The
compiled reference (gfacciol/mgm, built by oracle/Makefile from the sources where
they lie) is called through oracle/ref_harness.cc; its inputs and outputs are
stored as compressed .npz files.  Nothing but data is written: arrays of
inputs, parameters and the reference's outputs.

    python tests/golden/make_golden.py            # regenerate everything

CENSUS_NCC_WIN is an environment parameter the reference caches on first use
(smartparameter.h:26-50), so every census window runs in its own process.
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def fountain_crop(x0=250, y0=180, w=96, h=40):
    """A crop of the reference's own example pair (data/fountain23-im{L,R}.png), as planar float RGB."""
    from PIL import Image
    out = []
    for side in "LR":
        im = np.asarray(Image.open("/root/reference/data/fountain23-im%s.png" % side)).astype(np.float32)
        out.append(np.ascontiguousarray(im[y0:y0 + h, x0:x0 + w, :3].transpose(2, 0, 1)))
    return out


def worker(win):
    from mgm_amd import synth
    from oracle.oracle import Reference
    ref = Reference()
    assert ref.census_win() == win
    rng = np.random.default_rng(1234 + win)
    cases = {}

    # ---- cost volumes (images -> C) ----
    uL, uR = fountain_crop()
    gray = [a.mean(axis=0, keepdims=True).round().astype(np.float32) for a in (uL, uR)]
    s_u, s_v, _ = synth.stereo_pair(64, 24, -10, 6, seed=7)
    cv_cases = [
        ("fountain_rgb", uL, uR, -20, 12, "none", "ad", np.inf),
        ("fountain_rgb_trunc", uL, uR, -20, 12, "none", "ad", 30.0),
        ("fountain_gray_census", gray[0], gray[1], -20, 12, "none", "census", np.inf),
        ("fountain_rgb_census", uL, uR, -20, 12, "none", "census", np.inf),
        ("synth_census_allinvalid", s_u, s_v, 70, 80, "none", "census", np.inf),   # every q outside => zeros
        ("synth_sd", s_u, s_v, -10, 6, "none", "sd", 400.0),
        ("synth_pcensus_ad", s_u, s_v, -10, 6, "census", "ad", np.inf),            # -p census keeps the AD cost
        ("synth_sobelx_ad", s_u, s_v, -10, 6, "sobelx", "ad", np.inf),             # prefilters of the non-census costs
        ("fountain_rgb_sobelx_sd", uL, uR, -20, 12, "sobelx", "sd", 9000.0),
        ("synth_gblur_ad", s_u, s_v, -10, 6, "gblur", "ad", 40.0),
        ("fountain_rgb_gblur_sd", uL, uR, -20, 12, "gblur", "sd", np.inf),
        ("fountain_rgb_ncc", uL, uR, -20, 12, "none", "ncc", np.inf),               # CENSUS_NCC_WIN is the NCC window too
        ("synth_ncc_trunc", s_u, s_v, -10, 6, "none", "ncc", 0.5),
        ("synth_btad", s_u, s_v, -10, 6, "none", "btad", np.inf),
        ("fountain_rgb_btsd", uL, uR, -20, 12, "none", "btsd", 2000.0),
    ]
    if win != 3:
        cv_cases = [c for c in cv_cases if "census" in c[0] or "_ncc" in c[0]]
    for name, u, v, dmin, dmax, pre, dist, td in cv_cases:
        if (pre == "census" or dist == "census") and (u.shape[0] * (win * win - 1)) % 8:
            continue
        C = ref.costvolume(u, v, dmin, dmax, pre, dist, td)
        cases["cv_%s_w%d" % (name, win)] = dict(kind="cv", u=u, v=v, dmin=dmin, dmax=dmax, prefilter=pre,
                                                 distance=dist, truncDist=td, census_win=win, C=C)
    if win == 3:
        # ---- weights ----
        for aP, aT in [(4.0, 5.0), (0.3, 12.0)]:
            cases["w_fountain_%g_%g" % (aP, aT)] = dict(kind="weights", u=uL, aP=aP, aThresh=aT,
                                                          w8=ref.weights(uL, aP, aT))
        # ---- aggregation (C -> S, out, outcost) + refinement ----
        C0 = ref.costvolume(gray[0], gray[1], -20, 12, "none", "census", np.inf)   # 33 labels
        Cr = synth.raw_volume(37, 21, 64, seed=5, inf_frac=0.04)
        Cr2 = synth.raw_volume(29, 33, 20, seed=6, inf_frac=0.02)
        wts = ref.weights(uL, 4.0, 5.0)
        wr = np.where(rng.random((8, 21, 37)) < 0.5, 0.3, 1.0).astype(np.float32)
        agg = [
            ("census_h_o4_t2", C0, -20, None, 8.0, 32.0, 4, 2, 0, 1),
            ("census_h_o8_t3", C0, -20, None, 8.0, 32.0, 8, 3, 0, 1),
            ("census_h_o8_t4", C0, -20, None, 8.0, 32.0, 8, 4, 0, 1),
            ("census_fh_o8_t3", C0, -20, None, 2.0, 20000.0, 8, 3, 1, 1),
            ("census_fh_o4_t2", C0, -20, None, 2.0, 9.0, 4, 2, 1, 1),
            ("census_h_o8_t3_w", C0, -20, wts, 8.0, 32.0, 8, 3, 0, 1),
            ("raw64_h_o8_t1", Cr, 0, None, 8.0, 32.0, 8, 1, 0, 1),
            ("raw64_h_o8_t3_nofix", Cr, 0, None, 8.0, 32.0, 8, 3, 0, 0),
            ("raw64_fh_o8_t4_w", Cr, 0, wr, 1.5, np.inf, 8, 4, 1, 1),
            ("raw20_h_o2_t2", Cr2, -7, None, 1.3, 7.7, 2, 2, 0, 1),
            ("raw20_fh_o3_t1", Cr2, -7, None, 2.0, 9.0, 3, 1, 1, 1),
        ]
        # ---- NaN costs: what the reference makes of them turns on the operand order of its minima (`a < b ? a : b`,
        # fmin3, the strict `<` scan of Dvec::get_minvalue) -- the library's operand-order-faithful kernel
        # (mgm_pass_exact.hip) and the oracle are pinned on these.  All four update functions, every TSGM.
        Cn = synth.raw_volume(31, 23, 24, seed=11, inf_frac=0.03)
        Cn[rng.random(Cn.shape) < 0.01] = np.nan
        Cn[5, 7, :] = np.nan          # a pixel without any comparable cost
        Cn[9, 3:6, 0] = -0.0          # (and signed zeros: of equal minima the scan keeps the first)
        wn = np.where(rng.random((8, 23, 31)) < 0.5, 4.0, 1.0).astype(np.float32)
        agg += [
            ("nan24_h_o4_t2", Cn, -9, None, 8.0, 32.0, 4, 2, 0, 1),
            ("nan24_h_o8_t3", Cn, -9, None, 8.0, 32.0, 8, 3, 0, 1),
            ("nan24_h_o8_t4_w", Cn, -9, wn, 8.0, 32.0, 8, 4, 0, 1),
            ("nan24_h_o8_t1_p2inf", Cn, -9, None, 8.0, np.inf, 8, 1, 0, 0),
            ("nan24_fh_o4_t2", Cn, -9, None, 2.0, 9.0, 4, 2, 1, 1),
            ("nan24_fh_o8_t3", Cn, -9, None, 2.0, 20000.0, 8, 3, 1, 1),
            ("nan24_fh_o8_t4_w", Cn, -9, wn, 1.5, np.inf, 8, 4, 1, 1),
        ]
        for name, C, dmin, w8, P1, P2, NDIR, MGM, FH, FIX in agg:
            S, out, outc = ref.mgm(C, dmin, P1, P2, NDIR, MGM, FH, FIX, w8)
            d = dict(kind="agg", C=C, dmin=dmin, P1=P1, P2=P2, NDIR=NDIR, MGM=MGM, FH=FH, FIX=FIX, S=S, out=out,
                     outcost=outc)
            if w8 is not None:
                d["w8"] = w8
            for meth in ("vfit", "parabola", "cubic", "parabolaOCV"):
                ro, rc = ref.refine(S, dmin, meth, out, outc)
                d["out_" + meth] = ro
                d["outcost_" + meth] = rc
            cases["agg_" + name] = d
    only = os.environ.get("MGM_GOLDEN_ONLY")  # add fixtures without rewriting the (byte-wise timestamped) old ones
    n = 0
    for name, d in cases.items():
        if only and only not in name:
            continue
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        n += 1
    print("window %d: wrote %d cases" % (win, n))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker(int(sys.argv[1]))
    else:
        for f in os.listdir(HERE):
            if f.endswith(".npz") and not os.environ.get("MGM_GOLDEN_ONLY"):
                os.remove(os.path.join(HERE, f))
        for win in (3, 5, 7):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), str(win)],
                                  env=dict(os.environ, CENSUS_NCC_WIN=str(win)))
        files = sorted(f for f in os.listdir(HERE) if f.endswith(".npz"))
        json.dump(files, open(os.path.join(HERE, "INDEX.json"), "w"), indent=1)
        print(len(files), "fixtures,", sum(os.path.getsize(os.path.join(HERE, f)) for f in files) // 1024, "KiB")
