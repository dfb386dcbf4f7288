"""ctypes bindings for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Two libraries live here:

* ``libmgm_oracle.so``  -- the plain-C restatement (oracle/mgm_oracle.c),
  class :class:`Oracle`.
* ``_ref/libmgm_ref.so`` -- the real reference (gfacciol/mgm) compiled from
  ``/root/reference`` behind oracle/ref_harness.cc, class :class:`Reference`.
  It exists only if it was built in the container that has the reference.

Nothing in the product (mgm_amd/, include/, src/) may import this module; only
tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg do.

Array conventions: images are numpy float32 ``(nch, ny, nx)`` (planar, the
reference's ``Img`` layout, img.h:35-51); volumes are ``(ny, nx, L)``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libmgm_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libmgm_ref.so")
REFPOST_SO = os.path.join(HERE, "_ref", "libmgm_refpost.so")
REF_MGM = os.path.join(HERE, "_ref", "mgm")
REF_MGM_O = os.path.join(HERE, "_ref", "mgm_o")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def int_ranges(dminI, dmaxI):
    """Range images -> the integer ranges the reference's Dvecs get: `init(int min, int max)` called with floats
    (mgm_costvolume.h:323, dvec.cc:55) truncates toward zero."""
    lo = np.trunc(np.asarray(dminI, np.float64)).astype(np.int32)
    hi = np.trunc(np.asarray(dmaxI, np.float64)).astype(np.int32)
    return np.ascontiguousarray(lo.reshape(lo.shape[-2:])), np.ascontiguousarray(hi.reshape(hi.shape[-2:]))


def build(force=False):
    """Compile the oracle (and, when /root/reference exists, oracle/_ref)."""
    if force or not os.path.exists(ORACLE_SO) or (
        os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "mgm_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", HERE, "libmgm_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference"):  # (make is incremental: a newer ref_harness.cc rebuilds _ref/libmgm_ref.so)
        subprocess.check_call(["make", "-C", HERE, "ref"] + (["-B"] if force else []), stdout=subprocess.DEVNULL)


def usable_cpus(cap=64):
    """CPUs this process may really use: affinity mask and cgroup quota, capped."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, cap))


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim == 2:
        a = a[None]
    assert a.ndim == 3
    return a


def _optp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """The C restatement.  One instance per process is enough (stateless)."""

    def __init__(self, threads=1):
        build()
        L = self.lib = C.CDLL(ORACLE_SO)
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_num_threads(threads)
        L.orc_census_nwords.argtypes = [C.c_int, C.c_int]
        L.orc_census.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _u32p]
        L.orc_distance_index.argtypes = [C.c_char_p]
        L.orc_prefilter_index.argtypes = [C.c_char_p]
        L.orc_refinement_index.argtypes = [C.c_char_p]
        L.orc_costvolume.argtypes = [_f32p, _f32p] + [C.c_int] * 7 + [C.c_int, C.c_int, C.c_float, C.c_int, _f32p]
        L.orc_weights.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _f32p]
        L.orc_mgm.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float,
                              C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p, C.c_void_p]
        L.orc_refine.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p]
        L.orc_costvolume_ranged.argtypes = ([_f32p, _f32p] + [C.c_int] * 7 + [_i32p, _i32p, C.c_int, C.c_int, C.c_float,
                                                                                 C.c_int, _f32p])
        L.orc_mgm_ranged.argtypes = ([_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float] + [C.c_int] * 4
                                     + [C.c_void_p, _f32p, _f32p, C.c_void_p, C.c_uint])
        L.orc_refine_ranged.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p, C.c_int, _f32p, _f32p]

    def set_threads(self, n):
        self.lib.orc_set_num_threads(int(n))

    def census(self, u, winradius):
        u = _img(u)
        nch, ny, nx = u.shape
        nw = self.lib.orc_census_nwords(nch, winradius)
        out = np.zeros((nw, ny, nx), np.uint32)
        r = self.lib.orc_census(u, nx, ny, nch, winradius, out)
        if r < 0:
            raise ValueError("census window/channels not byte aligned (%d)" % r)
        return out

    def costvolume(self, u, v, dmin, dmax, prefilter="none", distance="ad", truncDist=np.inf, census_win=3):
        u, v = _img(u), _img(v)
        nch, ny, nx = u.shape
        _, vny, vnx = v.shape
        Cv = np.empty((ny, nx, dmax - dmin + 1), np.float32)
        r = self.lib.orc_costvolume(u, v, nx, ny, nch, vnx, vny, dmin, dmax,
                                    self.lib.orc_prefilter_index(prefilter.encode()),
                                    self.lib.orc_distance_index(distance.encode()),
                                    truncDist, census_win, Cv)
        if r:
            raise NotImplementedError("oracle: cost mode not restated (%d)" % r)
        return Cv

    def weights(self, u, aP, aThresh):
        u = _img(u)
        nch, ny, nx = u.shape
        w = np.empty((8, ny, nx), np.float32)
        self.lib.orc_weights(u, nx, ny, nch, aP, aThresh, w)
        return w

    def mgm(self, Cv, dmin, P1, P2, NDIR, MGM, FH=0, FIX=1, w8=None, dump_lr=False):
        Cv = np.ascontiguousarray(Cv, np.float32)
        ny, nx, L = Cv.shape
        S = np.empty_like(Cv)
        out = np.empty((ny, nx), np.float32)
        outc = np.empty((ny, nx), np.float32)
        if w8 is not None:
            w8 = np.ascontiguousarray(w8, np.float32)
            assert w8.shape == (8, ny, nx)
        lr = np.empty((NDIR, ny, nx, L), np.float32) if dump_lr else None
        r = self.lib.orc_mgm(Cv, nx, ny, L, dmin, _optp(w8), P1, P2, NDIR, MGM, FH, FIX, S, out, outc, _optp(lr))
        if r:
            raise ValueError("orc_mgm failed (%d)" % r)
        return (S, out, outc, lr) if dump_lr else (S, out, outc)

    def refine(self, S, dmin, method, out, outcost):
        S = np.ascontiguousarray(S, np.float32)
        ny, nx, L = S.shape
        out = np.array(out, np.float32, copy=True)
        outcost = np.array(outcost, np.float32, copy=True)
        r = self.lib.orc_refine(S, nx, ny, L, dmin, self.lib.orc_refinement_index(method.encode()), out, outcost)
        if r < 0:
            raise ValueError("orc_refine failed (%d)" % r)
        return out, outcost


    # ---- ragged ranges (round 6): hull volumes (ny, nx, L) + integer range images (ny, nx) ----
    def costvolume_ranged(self, u, v, lo, hi, hmin, hmax, prefilter="none", distance="ad", truncDist=np.inf, census_win=3):
        """Cost volume on the hull [hmin, hmax]; +INF where a pixel does not own the label."""
        u, v = _img(u), _img(v)
        nch, ny, nx = u.shape
        _, vny, vnx = v.shape
        L = hmax - hmin + 1
        Cv = np.empty((ny, nx, L), np.float32)
        r = self.lib.orc_costvolume_ranged(u, v, nx, ny, nch, vnx, vny, hmin, L, np.ascontiguousarray(lo, np.int32),
                                           np.ascontiguousarray(hi, np.int32),
                                           self.lib.orc_prefilter_index(prefilter.encode()),
                                           self.lib.orc_distance_index(distance.encode()), truncDist, census_win, Cv)
        if r:
            raise ValueError("orc_costvolume_ranged failed (%d)" % r)
        return Cv

    def mgm_ranged(self, Cv, hmin, lo, hi, P1, P2, NDIR, MGM, FH=0, FIX=1, w8=None, srange=None, want_S=True,
                   dump_lr=False):
        """mgm() with range images.  srange = (slo, shi, shmin, shmax): the (narrowed) ranges mgm() is CALLED with while
        the cost volume keeps (lo, hi) -- main()'s TSGM_ITER loop.  dump_lr: True (every pass) or a tuple of pass numbers.
        Returns (S, out, outcost[, Lr])."""
        Cv = np.ascontiguousarray(Cv, np.float32)
        ny, nx, L = Cv.shape
        lo, hi = np.ascontiguousarray(lo, np.int32), np.ascontiguousarray(hi, np.int32)
        if srange is None:
            slo = shi = None
            shmin, sL = hmin, L
        else:
            slo, shi = np.ascontiguousarray(srange[0], np.int32), np.ascontiguousarray(srange[1], np.int32)
            shmin, sL = srange[2], srange[3] - srange[2] + 1
        S = np.empty((ny, nx, sL), np.float32) if want_S else None
        out = np.empty((ny, nx), np.float32)
        outc = np.empty((ny, nx), np.float32)
        if w8 is not None:
            w8 = np.ascontiguousarray(w8, np.float32)
            assert w8.shape == (8, ny, nx)
        passes = tuple(range(NDIR)) if dump_lr is True else tuple(sorted(set(dump_lr or ())))  # (one volume per DISTINCT pass, in pass order)
        lr = np.empty((len(passes), ny, nx, L), np.float32) if passes else None
        r = self.lib.orc_mgm_ranged(Cv, nx, ny, L, hmin, lo, hi, _optp(slo), _optp(shi), shmin, sL, _optp(w8), P1, P2,
                                    NDIR, MGM, FH, FIX, _optp(S), out, outc, _optp(lr), sum(1 << q for q in passes))
        if r:
            raise ValueError("orc_mgm_ranged failed (%d)" % r)
        return (S, out, outc, lr) if dump_lr else (S, out, outc)

    def refine_ranged(self, S, shmin, slo, shi, method, out, outcost):
        S = np.ascontiguousarray(S, np.float32)
        ny, nx, L = S.shape
        out = np.array(out, np.float32, copy=True)
        outcost = np.array(outcost, np.float32, copy=True)
        r = self.lib.orc_refine_ranged(S, nx, ny, L, shmin, np.ascontiguousarray(slo, np.int32),
                                       np.ascontiguousarray(shi, np.int32),
                                       self.lib.orc_refinement_index(method.encode()), out, outcost)
        if r < 0:
            raise ValueError("orc_refine_ranged failed (%d)" % r)
        return out, outcost


class Reference:
    """The compiled reference itself (only where oracle/_ref was built)."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self):
        if not os.path.exists(REF_SO):
            build()
        L = self.lib = C.CDLL(REF_SO)
        L.ref_census_win.restype = C.c_int
        L.ref_costvolume.argtypes = [_f32p, _f32p] + [C.c_int] * 7 + [C.c_char_p, C.c_char_p, C.c_float, _f32p]
        L.ref_census.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, C.c_int]
        L.ref_weights.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _f32p]
        L.ref_mgm.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, _f32p, _f32p]
        L.ref_refine.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, _f32p, _f32p]
        if hasattr(L, "ref_seconds"):
            L.ref_seconds.restype = C.c_double
        if hasattr(L, "ref_mgm_ranged"):
            L.ref_costvolume_ranged.argtypes = ([_f32p, _f32p] + [C.c_int] * 5 + [_f32p, _f32p, C.c_int, C.c_int, C.c_char_p,
                                                                                   C.c_char_p, C.c_float, C.c_float, _f32p])
            L.ref_mgm_ranged.argtypes = ([_f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float] + [C.c_int] * 4
                                         + [C.c_float, C.c_void_p, _f32p, _f32p])
            L.ref_refine_ranged.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int, C.c_char_p, _f32p, _f32p]

    def has_ranged(self):
        return hasattr(self.lib, "ref_mgm_ranged")

    # ---- ragged ranges: the reference takes the float range IMAGES (its Dvec constructor truncates them) ----
    def costvolume_ranged(self, u, v, dminI, dmaxI, hmin, hmax, prefilter="none", distance="ad", truncDist=np.inf,
                          fill=np.inf):
        u, v = _img(u), _img(v)
        nch, ny, nx = u.shape
        _, vny, vnx = v.shape
        L = hmax - hmin + 1
        Cv = np.empty((ny, nx, L), np.float32)
        r = self.lib.ref_costvolume_ranged(u, v, nx, ny, nch, vnx, vny, np.ascontiguousarray(dminI, np.float32),
                                           np.ascontiguousarray(dmaxI, np.float32), hmin, L, prefilter.encode(),
                                           distance.encode(), truncDist, fill, Cv)
        if r:
            raise ValueError("ref_costvolume_ranged: a range leaves the hull (%d)" % r)
        return Cv

    def mgm_ranged(self, Cv, hmin, dminI, dmaxI, P1, P2, NDIR, MGM, FH=0, FIX=1, w8=None, srange=None, want_S=True,
                   fill=np.inf):
        Cv = np.ascontiguousarray(Cv, np.float32)
        ny, nx, L = Cv.shape
        dminI, dmaxI = np.ascontiguousarray(dminI, np.float32), np.ascontiguousarray(dmaxI, np.float32)
        if srange is None:
            smin = smax = None
            shmin, sL = hmin, L
        else:
            smin, smax = np.ascontiguousarray(srange[0], np.float32), np.ascontiguousarray(srange[1], np.float32)
            shmin, sL = srange[2], srange[3] - srange[2] + 1
        S = np.empty((ny, nx, sL), np.float32) if want_S else None
        out = np.empty((ny, nx), np.float32)
        outc = np.empty((ny, nx), np.float32)
        if w8 is not None:
            w8 = np.ascontiguousarray(w8, np.float32)
        with _quiet_stdout():
            r = self.lib.ref_mgm_ranged(Cv, nx, ny, dminI, dmaxI, hmin, L, _optp(smin), _optp(smax), shmin, sL, _optp(w8),
                                        P1, P2, NDIR, MGM, FH, FIX, fill, _optp(S), out, outc)
        if r:
            raise ValueError("ref_mgm_ranged: a range leaves the hull (%d)" % r)
        return S, out, outc

    def refine_ranged(self, S, shmin, sminI, smaxI, method, out, outcost):
        S = np.ascontiguousarray(S, np.float32)
        ny, nx, sL = S.shape
        out = np.array(out, np.float32, copy=True)
        outcost = np.array(outcost, np.float32, copy=True)
        self.lib.ref_refine_ranged(S, nx, ny, np.ascontiguousarray(sminI, np.float32),
                                   np.ascontiguousarray(smaxI, np.float32), shmin, sL, method.encode(), out, outcost)
        return out, outcost

    def seconds(self):
        """Wall time of the reference function inside the last costvolume / mgm / refine call (not the container copies)."""
        return float(self.lib.ref_seconds())

    def census_win(self):
        """CENSUS_NCC_WIN as cached by the reference (one value per process)."""
        return self.lib.ref_census_win()

    def census(self, u, winradius):
        u = _img(u)
        nch, ny, nx = u.shape
        out = np.zeros((16, ny, nx), np.float32)
        nw = self.lib.ref_census(u, nx, ny, nch, winradius, out, 16)
        return out[:nw].view(np.uint32).copy()

    def costvolume(self, u, v, dmin, dmax, prefilter="none", distance="ad", truncDist=np.inf):
        u, v = _img(u), _img(v)
        nch, ny, nx = u.shape
        _, vny, vnx = v.shape
        Cv = np.empty((ny, nx, dmax - dmin + 1), np.float32)
        self.lib.ref_costvolume(u, v, nx, ny, nch, vnx, vny, dmin, dmax, prefilter.encode(), distance.encode(),
                                truncDist, Cv)
        return Cv

    def weights(self, u, aP, aThresh):
        u = _img(u)
        nch, ny, nx = u.shape
        w = np.empty((8, ny, nx), np.float32)
        self.lib.ref_weights(u, nx, ny, nch, aP, aThresh, w)
        return w

    def mgm(self, Cv, dmin, P1, P2, NDIR, MGM, FH=0, FIX=1, w8=None):
        Cv = np.ascontiguousarray(Cv, np.float32)
        ny, nx, L = Cv.shape
        S = np.empty_like(Cv)
        out = np.empty((ny, nx), np.float32)
        outc = np.empty((ny, nx), np.float32)
        if w8 is not None:
            w8 = np.ascontiguousarray(w8, np.float32)
        with _quiet_stdout():
            self.lib.ref_mgm(Cv, nx, ny, dmin, dmin + L - 1, _optp(w8), P1, P2, NDIR, MGM, FH, FIX,
                             S.ctypes.data_as(C.c_void_p), out, outc)
        return S, out, outc

    def refine(self, S, dmin, method, out, outcost):
        S = np.ascontiguousarray(S, np.float32)
        ny, nx, L = S.shape
        out = np.array(out, np.float32, copy=True)
        outcost = np.array(outcost, np.float32, copy=True)
        self.lib.ref_refine(S, nx, ny, dmin, dmin + L - 1, method.encode(), out, outcost)
        return out, outcost


class RefPost:
    """median_filter / leftright_test / update_dmin_dmax of the compiled reference (oracle/ref_post_harness.cc), where built."""

    @staticmethod
    def available():
        return os.path.exists(REFPOST_SO)

    def __init__(self):
        L = self.lib = C.CDLL(REFPOST_SO)
        L.refpost_median.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        L.refpost_leftright.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_float]
        L.refpost_update_ranges.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int]

    def median(self, u, radius):
        u = _img(u)
        nch, ny, nx = u.shape
        out = np.empty_like(u)
        self.lib.refpost_median(u, nx, ny, nch, radius, out)
        return out

    def leftright(self, d, other, tau):
        d = np.array(d, np.float32, copy=True).reshape(d.shape[-2], d.shape[-1])
        other = np.ascontiguousarray(other, np.float32).reshape(other.shape[-2], other.shape[-1])
        self.lib.refpost_leftright(d, d.shape[1], d.shape[0], other, other.shape[1], other.shape[0], tau)
        return d

    def update_ranges(self, outoff, lo, hi, slack=3, radius=2):
        o = np.ascontiguousarray(outoff, np.float32).reshape(outoff.shape[-2], outoff.shape[-1])
        lo, hi = np.array(lo, np.float32, copy=True).reshape(o.shape), np.array(hi, np.float32, copy=True).reshape(o.shape)
        self.lib.refpost_update_ranges(o, o.shape[1], o.shape[0], lo, hi, slack, radius)
        return lo, hi


class _quiet_stdout:
    """The reference prints pass digits with printf (mgm_core.cc:491)."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        self.null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self.null, 1)

    def __exit__(self, *a):
        C.CDLL(None).fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)
        os.close(self.null)


def bits_equal(a, b):
    """Bit-exact comparison of float32 arrays with NaN == NaN (any payload)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    return np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])
