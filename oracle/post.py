"""numpy restatement of what the reference's main() does to the disparity maps right after the path -- TEST
INFRASTRUCTURE ONLY (see oracle/oracle.py for who may import this).  Plain loops: small images only.

Pinned on the compiled reference (oracle/_ref/libmgm_refpost.so) by tests/test_oracle_vs_ref.py, on maps with NaN labels and
+INF costs -- exactly the maps the NaN-faithful path produces."""
import numpy as np


def median(u, radius):
    """median_filter, img_tools.h:203-238: per channel, the (2r+1)^2 window clipped at the border, NaN samples dropped, the
    UPPER median v[n/2] of what is left (nth_element); an all-NaN window leaves the pixel as it is.  +-INF are ordinary
    samples (they order like any value)."""
    u = np.asarray(u, np.float32)
    u3 = u.reshape((-1,) + u.shape[-2:])
    out = u3.copy()
    nch, ny, nx = u3.shape
    for c in range(nch):
        for y in range(ny):
            for x in range(nx):
                w = u3[c, max(0, y - radius):y + radius + 1, max(0, x - radius):x + radius + 1].ravel()
                w = np.sort(w[~np.isnan(w)])
                if w.size:
                    out[c, y, x] = w[w.size // 2]
    return out.reshape(u.shape)


def c_round(v):
    """round() of <math.h>: half away from zero."""
    return np.sign(v) * np.floor(np.abs(v) + 0.5)


def leftright(d, other, tau):
    """leftright_test, mgm.cc:68-91: Lx = (int)round(x + d); outside `other` -> NaN; else Rx = Lx + other[Lx, y] and the
    label is dropped iff fabs(Rx - x) > tau.  Consequences for non-finite inputs, all as the compiled reference behaves on
    x86-64 (cvttsd2si of NaN / out-of-range = INT_MIN, which lies outside every image): a NaN or infinite d -> NaN; a NaN in
    `other` makes the comparison false -> the label is KEPT; an infinite value in `other` -> dropped."""
    d = np.asarray(d, np.float32)
    other = np.asarray(other, np.float32)
    ny, nx = d.shape
    out = np.full_like(d, np.nan)
    for y in range(ny):
        for x in range(nx):
            s = np.float32(x) + d[y, x]            # int + float -> float
            r = c_round(np.float64(s))
            if not np.isfinite(r) or r < -2147483648.0 or r > 2147483647.0:
                continue                           # INT_MIN: outside
            Lx = int(r)
            if 0 <= Lx < other.shape[1]:
                Rx = np.float32(Lx) + other[y, Lx]
                if not (abs(np.float64(np.float32(Rx - np.float32(x)))) > np.float64(np.float32(tau))):
                    out[y, x] = d[y, x]
    return out
