"""Development aid (CPU): emulate the carry guess of fh_scan (mgm_pass_common.h) in numpy float32 on the
per-pass Lr slabs of a real census volume and report how often the first sweep rejects it."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from oracle.oracle import Oracle
from mgm_amd import synth

f32 = np.float32
INF = f32(np.inf)
LPL = 4


def seq_scan(M, P1, fwd):  # exact sequential recurrence over the 256 labels, M: [N,256]
    F = M.copy()
    rng = range(1, F.shape[1]) if fwd else range(F.shape[1] - 2, -1, -1)
    d = -1 if fwd else 1
    for o in rng:
        F[:, o] = np.minimum(F[:, o], F[:, o + d] + P1)
    return F


def shift_rows(c, k, fwd):  # DPP row shift by k inside rows of 16 lanes; lanes without a source: None mask
    N = c.shape[0]
    x = c.reshape(N, 4, 16)
    out = np.full_like(x, np.nan)
    if fwd:
        out[:, :, k:] = x[:, :, :-k]
    else:
        out[:, :, :-k] = x[:, :, k:]
    return out.reshape(N, 64)


def guess(a, P1, fwd):
    R = f32(LPL) * P1
    c = a.copy()
    for k in (1, 2, 4, 8):
        t = c + f32(k) * R
        s = shift_rows(t, k, fwd)
        c = np.where(np.isnan(s), c, np.minimum(c, s))
    lane = np.arange(64)
    li, row = lane & 15, lane >> 4
    if fwd:
        offA = np.where(row >= 1, (li + 1).astype(f32) * R, INF).astype(f32)
        src = c[:, np.maximum(row * 16 - 1, 0)]
        c = np.minimum(c, src + offA)
        offB = np.where(row >= 2, (li + 1 + np.where(row == 3, 16, 0)).astype(f32) * R, INF).astype(f32)
        c = np.minimum(c, c[:, [31]] + offB)
    else:
        r16 = f32(16) * R
        t3 = c[:, 48]
        t2 = np.minimum(c[:, 32], t3 + r16)
        t1 = np.minimum(c[:, 16], t2 + r16)
        tp = np.where(row == 0, t1[:, None], np.where(row == 1, t2[:, None], t3[:, None]))
        offD = np.where(row <= 2, (16 - li).astype(f32) * R, INF).astype(f32)
        c = np.minimum(c, tp + offD)
    return c


def check(M, P1, fwd):
    N = M.shape[0]
    X = M.reshape(N, 64, LPL)
    order = range(LPL) if fwd else range(LPL - 1, -1, -1)
    a = None
    for q in order:
        a = X[:, :, q].copy() if a is None else np.minimum(X[:, :, q], a + P1)
    c = guess(a, P1, fwd)
    F = seq_scan(M, P1, fwd).reshape(N, 64, LPL)
    true_c = F[:, :, LPL - 1 if fwd else 0]
    bad = ~((c == true_c) | (np.isinf(c) & np.isinf(true_c)))
    return bad, c, true_c, F.reshape(N, -1)


def main():
    nx, ny, L = 384, 40, 256
    u, v, _ = synth.stereo_pair(nx, ny, -(L - 1) * 3 // 4, 0)
    o = Oracle(threads=8)
    Cv = o.costvolume(u, v, -(L - 1), 0, "none", "census", np.inf, 5)
    P1, P2 = f32(2.0), f32(20000.0)
    _, _, _, lr = o.mgm(Cv, -(L - 1), float(P1), float(P2), 8, 3, FH=1, FIX=1, dump_lr=True)
    M = lr.reshape(-1, L)
    with np.errstate(invalid="ignore"):
        badf, cf, tf, F = check(M, P1, True)
        badb, cb, tb, _ = check(F, P1, False)
    print("slabs", M.shape[0], "fwd rejected %.3f%%" % (100 * badf.any(1).mean()), "bwd rejected %.3f%%" % (100 * badb.any(1).mean()))
    for name, bad, c, t, src in (("fwd", badf, cf, tf, M), ("bwd", badb, cb, tb, F)):
        idx = np.nonzero(bad.any(1))[0]
        for i in idx[:3]:
            lanes = np.nonzero(bad[i])[0]
            print(name, "slab", i, "lanes", lanes[:8], "guess", c[i, lanes[:4]], "true", t[i, lanes[:4]])
            print("   n_inf", int(np.isinf(src[i]).sum()), "min", float(src[i][np.isfinite(src[i])].min()) if np.isfinite(src[i]).any() else None,
                  "values near", src[i].reshape(64, LPL)[max(lanes[0] - 2, 0):lanes[0] + 2].ravel())




def sweeps_needed(M, c, P1, fwd, limit=70):
    """plain fixed-point sweeps after a rejected guess: number of sweeps until nothing changes"""
    N = M.shape[0]
    X = M.reshape(N, 64, LPL)
    order = list(range(LPL)) if fwd else list(range(LPL - 1, -1, -1))
    n = np.zeros(N, int)
    active = np.ones(N, bool)
    for it in range(limit):
        cin = np.full_like(c, np.inf)
        if fwd:
            cin[:, 1:] = c[:, :-1]
        else:
            cin[:, :-1] = c[:, 1:]
        f = None
        for q in order:
            f = np.minimum(X[:, :, q], (cin if f is None else f) + P1)
        changed = ~((f == c) | (np.isinf(f) & np.isinf(c)))
        ch = changed.any(1)
        n[active & ch] += 1
        active &= ch
        c = f
        if not active.any():
            break
    return n


def hist():
    nx, ny, L = int(os.environ.get("NX", 384)), 40, 256
    u, v, _ = synth.stereo_pair(nx, ny, -(L - 1) * 3 // 4, 0)
    o = Oracle(threads=8)
    Cv = o.costvolume(u, v, -(L - 1), 0, "none", "census", np.inf, 5)
    P1, P2 = f32(2.0), f32(20000.0)
    _, _, _, lr = o.mgm(Cv, -(L - 1), float(P1), float(P2), 8, 3, FH=1, FIX=1, dump_lr=True)
    M = lr.reshape(-1, L)
    with np.errstate(invalid="ignore"):
        badf, cf, tf, F = check(M, P1, True)
        badb, cb, tb, _ = check(F, P1, False)
        for name, bad, c, src, fwd in (("fwd", badf, cf, M, True), ("bwd", badb, cb, F, False)):
            idx = np.nonzero(bad.any(1))[0]
            n = sweeps_needed(src[idx], c[idx], P1, fwd)
            print(name, "rejected %.3f%%" % (100.0 * len(idx) / M.shape[0]), "extra plain sweeps histogram:",
                  np.bincount(np.minimum(n, 20))[:21])




def repair(a, P1, fwd):
    """numpy restatement of fh_repair (f64 min-plus scan + successive rounding)"""
    N = a.shape[0]
    P1d, Rd = np.float64(P1), np.float64(LPL) * np.float64(P1)
    b = (a + P1).astype(np.float64)

    def shift(x, d):
        out = np.full_like(x, np.inf)
        if fwd:
            out[:, d:] = x[:, :-d]
        else:
            out[:, :-d] = x[:, d:]
        return out
    s = shift(b, 1) + Rd
    d = 1
    while d < 64:
        s = np.minimum(s, shift(s, d) + d * Rd)
        d *= 2
    s = s - P1d
    B = np.float64(2.0) ** (np.floor(np.log2(np.float64(P1))) + 1)
    for it in range(64):
        reach = np.isfinite(s) & (s >= B)
        if not reach.any():
            break
        magic = B * 805306368.0
        r = (s + magic) - magic
        s = np.where(reach, r, s)
        B = B + B
    return np.minimum(a, s.astype(f32))


def check_repair():
    nx, ny, L = int(os.environ.get("NX", 384)), 40, 256
    u, v, _ = synth.stereo_pair(nx, ny, -(L - 1) * 3 // 4, 0)
    o = Oracle(threads=8)
    Cv = o.costvolume(u, v, -(L - 1), 0, "none", "census", np.inf, 5)
    for P1 in (f32(2.0), f32(8.0), f32(3.0), f32(2.5)):
        _, _, _, lr = o.mgm(Cv, -(L - 1), float(P1), 20000.0, 8, 3, FH=1, FIX=1, dump_lr=True)
        M = lr.reshape(-1, L)
        with np.errstate(invalid="ignore"):
            for fwd in (True, False):
                src = M if fwd else seq_scan(M, P1, True)
                X = src.reshape(-1, 64, LPL)
                order = range(LPL) if fwd else range(LPL - 1, -1, -1)
                a = None
                for q in order:
                    a = X[:, :, q].copy() if a is None else np.minimum(X[:, :, q], a + P1)
                c = repair(a, P1, fwd)
                F = seq_scan(src, P1, fwd).reshape(-1, 64, LPL)
                t = F[:, :, LPL - 1 if fwd else 0]
                bad = ~((c == t) | (np.isinf(c) & np.isinf(t)))
                print("P1", P1, "fwd" if fwd else "bwd", "repair wrong on %.4f%% of slabs" % (100.0 * bad.any(1).mean()))


if __name__ == "__main__":
    check_repair() if os.environ.get("REPAIR") else (hist() if os.environ.get("HIST") else main())
