import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases(kind):
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        name = os.path.basename(f)[:-4]
        if name.startswith({"cv": "cv_", "agg": "agg_", "weights": "w_"}[kind]):
            out.append(name)
    return out


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: (d[k].item() if d[k].ndim == 0 else d[k]) for k in d.files}


def ndiff(a, b):
    """Number of float32 words that differ bitwise (NaN == NaN whatever the payload)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    na, nb = np.isnan(a), np.isnan(b)
    return int(np.sum((a.view(np.uint32) != b.view(np.uint32)) & ~(na & nb)))


def labels_equal(a, b, cost):
    """Integer labels must match wherever a finite minimum exists (elsewhere the reference is undefined)."""
    fin = np.isfinite(cost)
    return bool(np.array_equal(np.asarray(a)[fin], np.asarray(b)[fin]))
