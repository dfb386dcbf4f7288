"""Deterministic synthetic stereo pairs for parity tests and bench.py.

``u`` is a 1-channel float32 image of integer grey levels 0..255 (smooth random
field + white noise); the ground-truth disparity is a piecewise-constant integer
map (a few fronto-parallel planes plus a slanted ramp) inside the search range;
``v(x+d) = u(x)`` forward-warped, holes filled with independent noise, plus a
+-2 grey-level perturbation.  Pure numpy, seeded: the same arrays everywhere.
"""
import numpy as np

SEED = 20150907


def _smooth_field(rng, ny, nx, cells=24):
    gy, gx = max(2, ny // cells + 2), max(2, nx // cells + 2)
    g = rng.random((gy, gx)).astype(np.float32)
    ys = np.linspace(0, gy - 1.001, ny, dtype=np.float32)
    xs = np.linspace(0, gx - 1.001, nx, dtype=np.float32)
    y0, x0 = ys.astype(np.int32), xs.astype(np.int32)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx).astype(np.float32)


def stereo_pair(nx, ny, dmin, dmax, seed=SEED, nch=1):
    """Returns (u, v, gt) with u, v float32 (nch, ny, nx) and gt int32 (ny, nx)."""
    rng = np.random.default_rng(seed)
    base = _smooth_field(rng, ny, nx) * 160.0 + rng.random((ny, nx)).astype(np.float32) * 95.0
    u = np.clip(np.rint(base), 0, 255).astype(np.float32)
    # disparity planes
    if dmax < dmin:
        dmin = dmax = 0
    span = dmax - dmin
    gt = np.full((ny, nx), dmin + span // 2, np.int32)
    nplanes = 4
    for k in range(nplanes):
        y0, y1 = sorted(rng.integers(0, ny, 2))
        x0, x1 = sorted(rng.integers(0, nx, 2))
        gt[y0:y1 + 1, x0:x1 + 1] = dmin + int(rng.integers(span // 8, max(span // 8, span - span // 8) + 1))
    # slanted ramp in a horizontal band
    y0 = ny // 3
    ramp = (dmin + span // 4 + (np.arange(nx) * (span // 2)) // max(1, nx - 1)).astype(np.int32)
    gt[y0:y0 + max(1, ny // 8), :] = ramp[None, :]
    gt = np.clip(gt, dmin, dmax)
    # forward warp
    v = rng.integers(0, 256, (ny, nx)).astype(np.float32)
    xs = np.arange(nx)[None, :] + gt
    ok = (xs >= 0) & (xs < nx)
    yy = np.broadcast_to(np.arange(ny)[:, None], (ny, nx))
    v[yy[ok], xs[ok]] = u[ok]
    v = np.clip(v + rng.integers(-2, 3, (ny, nx)).astype(np.float32), 0, 255).astype(np.float32)
    if nch == 1:
        return u[None], v[None], gt
    us, vs = [u], [v]
    for c in range(1, nch):
        g = np.float32(0.6 + 0.2 * c)
        us.append(np.clip(np.rint(u * g), 0, 255).astype(np.float32))
        vs.append(np.clip(np.rint(v * g), 0, 255).astype(np.float32))
    return np.stack(us), np.stack(vs), gt


def raw_volume(nx, ny, L, seed=SEED, maxcost=24, inf_frac=0.0):
    """Uniform random integer costs 0..maxcost as float32 [ny][nx][L] (isolates K3 from K1/K2)."""
    rng = np.random.default_rng(seed)
    C = rng.integers(0, maxcost + 1, (ny, nx, L)).astype(np.float32)
    if inf_frac > 0:
        C[rng.random((ny, nx, L)) < inf_frac] = np.inf
        C[..., 0][~np.isfinite(C).any(axis=2)] = 0.0  # every pixel keeps a finite hypothesis
    return C
