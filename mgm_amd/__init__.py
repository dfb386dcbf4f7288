"""mgm_amd -- Python (ctypes) binding of libmgm_hip.so, the MI355X MGM stereo core.

This module is host-side plumbing for tests and benchmarks: every compute call
goes through the C ABI declared in ``include/mgm_hip.h`` and runs as HIP
kernels on a gfx950 device.  There is no CPU path here; if the shared library
is missing or no device is usable, calls raise :class:`MgmError`.

The function names and argument meanings mirror the reference's three entry
points (``allocate_and_fill_sgm_costvolume``, ``mgm``, ``subpixel_refinement_sgm``;
mgm.cc:32-59) so tests read like calls into the reference.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGM_HIP_LIB") or os.path.join(_HERE, "lib", "libmgm_hip.so")

# every symbol include/mgm_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "mgm_ctx_create", "mgm_ctx_destroy", "mgm_last_error", "mgm_ctx_synchronize", "mgm_ctx_stream", "mgm_version", "mgm_ctx_trim",
    "mgm_timing_enable", "mgm_timing_reset", "mgm_timing_count", "mgm_timing_get",
    "mgm_img_create", "mgm_img_upload", "mgm_img_download", "mgm_img_dims", "mgm_img_device_ptr", "mgm_img_free",
    "mgm_cv_create", "mgm_cv_upload", "mgm_cv_download", "mgm_cv_dims", "mgm_cv_device_ptr", "mgm_cv_free",
    "mgm_costvolume_build_dev", "mgm_costvolume_build", "mgm_weights_dev",
    "mgm_aggregate_dev", "mgm_aggregate", "mgm_debug_download_lr", "mgm_refine_dev", "mgm_refine",
    "mgm_selftest_div3", "mgm_aggregate_passes_dev", "mgm_lr_device_ptr", "mgm_wta_rows_dev",
    "mgm_aggregate_batch_dev", "mgm_median_dev", "mgm_leftright_dev", "mgm_backproject_dev",
    "mgm_wta_windowed_dev", "mgm_update_ranges_dev", "mgm_costvolume_build_ranged_dev",
    "mgm_multi_create", "mgm_multi_destroy", "mgm_multi_size", "mgm_multi_ctx", "mgm_multi_last_error", "mgm_multi_plan",
    "mgm_multi_aggregate", "mgm_multi_transport", "mgm_img_device", "mgm_cv_device", "mgm_aggregate_passes_at_dev",
    "mgm_ctx_set_workspace_limit", "mgm_ctx_mem_info", "mgm_ctx_set_pipeline", "mgm_img_update", "mgm_debug_probe_workspace", "mgm_ctx_set_placement_tries",
]

MGM_OK, MGM_ERR_INVALID, MGM_ERR_UNSUPPORTED, MGM_ERR_HIP, MGM_ERR_NOMEM, MGM_ERR_INTERNAL = range(6)


class MgmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mgm_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load_library():
    """dlopen libmgm_hip.so (building nothing: see mgm_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MgmError(MGM_ERR_HIP, "%s not found: run `python -m mgm_amd.build` (hipcc, gfx950)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, fp, i, f, cp = C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_char_p
    pp = C.POINTER(C.c_void_p)
    L.mgm_version.restype = cp
    L.mgm_last_error.restype = cp
    L.mgm_last_error.argtypes = [vp]
    L.mgm_ctx_create.argtypes = [i, pp]
    L.mgm_ctx_destroy.argtypes = [vp]
    L.mgm_ctx_synchronize.argtypes = [vp]
    L.mgm_ctx_trim.argtypes = [vp]
    L.mgm_ctx_stream.argtypes = [vp]
    L.mgm_ctx_stream.restype = vp
    L.mgm_timing_enable.argtypes = [vp, i]
    L.mgm_timing_reset.argtypes = [vp]
    L.mgm_timing_count.argtypes = [vp]
    L.mgm_timing_get.argtypes = [vp, i, C.POINTER(cp), C.POINTER(f)]
    L.mgm_img_create.argtypes = [vp, i, i, i, pp]
    L.mgm_img_upload.argtypes = [vp, fp, i, i, i, pp]
    L.mgm_img_download.argtypes = [vp, vp, fp]
    L.mgm_img_update.argtypes = [vp, vp, fp]
    L.mgm_debug_probe_workspace.argtypes = [vp, i, C.POINTER(f)]
    L.mgm_ctx_set_placement_tries.argtypes = [vp, i]
    L.mgm_img_dims.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.mgm_img_device_ptr.argtypes = [vp]
    L.mgm_img_device_ptr.restype = vp
    L.mgm_img_free.argtypes = [vp, vp]
    L.mgm_cv_create.argtypes = [vp, i, i, i, i, pp]
    L.mgm_cv_upload.argtypes = [vp, fp, i, i, i, i, pp]
    L.mgm_cv_download.argtypes = [vp, vp, fp]
    L.mgm_cv_dims.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.mgm_cv_device_ptr.argtypes = [vp]
    L.mgm_cv_device_ptr.restype = vp
    L.mgm_cv_free.argtypes = [vp, vp]
    L.mgm_costvolume_build_dev.argtypes = [vp, vp, vp, i, i, cp, cp, f, i, pp]
    L.mgm_costvolume_build.argtypes = [vp, fp, fp, i, i, i, i, i, fp, fp, cp, cp, f, i, pp]
    L.mgm_weights_dev.argtypes = [vp, vp, f, f, pp]
    L.mgm_aggregate_dev.argtypes = [vp, vp, vp, f, f, i, i, i, i, cp, vp, vp, pp]
    L.mgm_aggregate.argtypes = [vp, vp, fp, f, f, i, i, i, i, cp, fp, fp, pp]
    L.mgm_median_dev.argtypes = [vp, vp, i, vp]
    L.mgm_costvolume_build_ranged_dev.argtypes = [vp, vp, vp, vp, vp, i, i, cp, cp, f, i, pp]
    L.mgm_wta_windowed_dev.argtypes = [vp, vp, i, i, cp, vp, vp, vp, vp]
    L.mgm_update_ranges_dev.argtypes = [vp, vp, vp, vp, i, i]
    L.mgm_leftright_dev.argtypes = [vp, vp, vp, f, vp]
    L.mgm_backproject_dev.argtypes = [vp, vp, vp, vp, vp]
    L.mgm_aggregate_batch_dev.argtypes = [vp, i, pp, pp, f, f, i, i, i, i, cp, pp, pp, pp]
    L.mgm_debug_download_lr.argtypes = [vp, i, fp]
    L.mgm_refine_dev.argtypes = [vp, vp, cp, vp, vp]
    L.mgm_refine.argtypes = [vp, vp, cp, fp, fp]
    L.mgm_selftest_div3.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.mgm_aggregate_passes_dev.argtypes = [vp, vp, vp, f, f, i, i, i, i]
    L.mgm_lr_device_ptr.argtypes = [vp, i]
    L.mgm_lr_device_ptr.restype = vp
    L.mgm_wta_rows_dev.argtypes = [vp, vp, i, i, vp, i, i, cp, vp, vp]
    ip = C.POINTER(i)
    L.mgm_multi_create.argtypes = [ip, i, pp]
    L.mgm_multi_destroy.argtypes = [vp]
    L.mgm_multi_size.argtypes = [vp]
    L.mgm_multi_ctx.argtypes = [vp, i]
    L.mgm_multi_ctx.restype = vp
    L.mgm_multi_last_error.argtypes = [vp]  # (NULL: why the last mgm_multi_create failed)
    L.mgm_multi_last_error.restype = cp
    L.mgm_multi_plan.argtypes = [i, i, i, ip, ip, ip, ip]
    L.mgm_multi_aggregate.argtypes = [vp, pp, pp, f, f, i, i, i, i, cp, vp, vp]
    L.mgm_multi_transport.argtypes = [vp]
    L.mgm_multi_transport.restype = cp
    L.mgm_img_device.argtypes = [vp]
    L.mgm_cv_device.argtypes = [vp]
    L.mgm_aggregate_passes_at_dev.argtypes = [vp, vp, vp, f, f, i, i, i, i, i, i, i]
    L.mgm_ctx_set_workspace_limit.argtypes = [vp, C.c_ulonglong]
    L.mgm_ctx_set_pipeline.argtypes = [vp, i]
    L.mgm_ctx_mem_info.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Image:
    """Device-resident planar float image (the reference's ``struct Img``)."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    @property
    def shape(self):
        nx, ny, nch = C.c_int(), C.c_int(), C.c_int()
        self.ctx.lib.mgm_img_dims(self.h, nx, ny, nch)
        return (nch.value, ny.value, nx.value)

    def download(self):
        out = np.empty(self.shape, np.float32)
        self.ctx._chk(self.ctx.lib.mgm_img_download(self.ctx.h, self.h, _ptr(out)))
        return out

    def update(self, a):
        """Refill from a host array of the image's own shape (no allocation)."""
        a = np.ascontiguousarray(a, np.float32)
        if a.size != int(np.prod(self.shape)):
            raise ValueError("Image.update: %r does not match the image's shape %r" % (a.shape, self.shape))
        self.ctx._chk(self.ctx.lib.mgm_img_update(self.ctx.h, self.h, _ptr(a)))

    def free(self):
        if self.h:
            self.ctx.lib.mgm_img_free(self.ctx.h, self.h)
            self.h = None


class CostVolume:
    """Device-resident dense volume [ny][nx][L] (the reference's ``costvolume_t``)."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    @property
    def dims(self):
        nx, ny, dmin, dmax = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.ctx.lib.mgm_cv_dims(self.h, nx, ny, dmin, dmax)
        return nx.value, ny.value, dmin.value, dmax.value

    def download(self):
        nx, ny, dmin, dmax = self.dims
        out = np.empty((ny, nx, dmax - dmin + 1), np.float32)
        self.ctx._chk(self.ctx.lib.mgm_cv_download(self.ctx.h, self.h, _ptr(out)))
        return out

    def free(self):
        if self.h:
            self.ctx.lib.mgm_cv_free(self.ctx.h, self.h)
            self.h = None


class Context:
    """One device + stream + workspace (``mgm_ctx``)."""

    def __init__(self, device=0, _borrowed=None):
        self.lib = load_library()
        self.owned = _borrowed is None
        if _borrowed is not None:  # a rank's context of an mgm_multi handle (destroyed with it)
            self.h = C.c_void_p(_borrowed)
            return
        h = C.c_void_p()
        r = self.lib.mgm_ctx_create(device, C.byref(h))
        if r:
            raise MgmError(r, "mgm_ctx_create(%d) failed: no usable HIP device" % device)
        self.h = h

    def close(self):
        if self.h and self.owned:
            self.lib.mgm_ctx_destroy(self.h)
        self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, r):
        if r:
            raise MgmError(r, self.lib.mgm_last_error(self.h).decode())

    def synchronize(self):
        self._chk(self.lib.mgm_ctx_synchronize(self.h))

    def trim(self):
        """Release the grow-only workspace (mgm_ctx_trim)."""
        self._chk(self.lib.mgm_ctx_trim(self.h))

    def set_workspace_limit(self, nbytes):
        """Cap on the workspace of one pass launch; larger batches run as several launches (mgm_ctx_set_workspace_limit)."""
        self._chk(self.lib.mgm_ctx_set_workspace_limit(self.h, int(nbytes)))

    def set_pipeline(self, depth):
        """depth >= 2: aggregate calls are deferred and gathered, `depth` of them run as one batched launch (results after the
        last call of a group, synchronize() or any download); see mgm_ctx_set_pipeline in mgm_hip.h."""
        self._chk(self.lib.mgm_ctx_set_pipeline(self.h, int(depth)))

    def mem_info(self):
        """(free, total) bytes of the context's device right now."""
        f, t = C.c_ulonglong(0), C.c_ulonglong(0)
        self._chk(self.lib.mgm_ctx_mem_info(self.h, C.byref(f), C.byref(t)))
        return int(f.value), int(t.value)

    def stream_ptr(self):
        """The hipStream_t everything is enqueued on (for torch.cuda.ExternalStream)."""
        return int(self.lib.mgm_ctx_stream(self.h) or 0)

    # ---- containers ----
    def upload_image(self, a):
        a = _f32(a)
        if a.ndim == 2:
            a = a[None]
        nch, ny, nx = a.shape
        h = C.c_void_p()
        self._chk(self.lib.mgm_img_upload(self.h, _ptr(a), nx, ny, nch, C.byref(h)))
        return Image(self, h)

    def new_image(self, nx, ny, nch=1):
        h = C.c_void_p()
        self._chk(self.lib.mgm_img_create(self.h, nx, ny, nch, C.byref(h)))
        return Image(self, h)

    def upload_volume(self, dense, dmin):
        dense = _f32(dense)
        ny, nx, L = dense.shape
        h = C.c_void_p()
        self._chk(self.lib.mgm_cv_upload(self.h, _ptr(dense), nx, ny, dmin, dmin + L - 1, C.byref(h)))
        return CostVolume(self, h)

    # ---- the path ----
    def costvolume_dev(self, u, v, dmin, dmax, prefilter="none", distance="ad", truncDist=float("inf"),
                       census_win=3, into=None):
        h = C.c_void_p(into.h.value) if into is not None else C.c_void_p()
        self._chk(self.lib.mgm_costvolume_build_dev(self.h, u.h, v.h, dmin, dmax, prefilter.encode(),
                                                    distance.encode(), truncDist, census_win, C.byref(h)))
        return into if into is not None else CostVolume(self, h)

    def costvolume_ranged_dev(self, u, v, dminI, dmaxI, hull_min, hull_max, prefilter="none", distance="ad", truncDist=float("inf"),
                              census_win=3, into=None):
        """Device form with per-pixel range images (device Images): the ragged volume over [hull_min, hull_max]."""
        h = C.c_void_p(into.h.value) if into is not None else C.c_void_p()
        self._chk(self.lib.mgm_costvolume_build_ranged_dev(self.h, u.h, v.h, dminI.h, dmaxI.h, hull_min, hull_max, prefilter.encode(),
                                                           distance.encode(), truncDist, census_win, C.byref(h)))
        return into if into is not None else CostVolume(self, h)

    def costvolume(self, u, v, dminI, dmaxI, prefilter="none", distance="ad", truncDist=float("inf"), census_win=3):
        """Host-buffer form with per-pixel range images, like the reference."""
        u, v = _f32(u), _f32(v)
        if u.ndim == 2:
            u, v = u[None], v[None]
        nch, ny, nx = u.shape
        _, vny, vnx = v.shape
        dminI, dmaxI = _f32(dminI), _f32(dmaxI)
        h = C.c_void_p()
        self._chk(self.lib.mgm_costvolume_build(self.h, _ptr(u), _ptr(v), nx, ny, nch, vnx, vny, _ptr(dminI),
                                                _ptr(dmaxI), prefilter.encode(), distance.encode(), truncDist,
                                                census_win, C.byref(h)))
        return CostVolume(self, h)

    def weights_dev(self, u, aP, aThresh, into=None):
        h = C.c_void_p(into.h.value) if into is not None else C.c_void_p()
        self._chk(self.lib.mgm_weights_dev(self.h, u.h, aP, aThresh, C.byref(h)))
        return into if into is not None else Image(self, h)

    def aggregate_dev(self, Cv, P1, P2, NDIR, MGM, use_fh=0, fix_overcount=1, w8=None, refine=None, out=None,
                      outcost=None, want_S=False):
        nx, ny, _, _ = Cv.dims
        out = out or self.new_image(nx, ny)
        outcost = outcost or self.new_image(nx, ny)
        S = C.c_void_p()
        self._chk(self.lib.mgm_aggregate_dev(self.h, Cv.h, w8.h if w8 is not None else None, P1, P2, NDIR, MGM,
                                             use_fh, fix_overcount, refine.encode() if refine else None, out.h,
                                             outcost.h, C.byref(S) if want_S else None))
        return (CostVolume(self, S) if want_S else None), out, outcost

    def wta_windowed_dev(self, Cv, NDIR, fix_overcount, refine, dminI, dmaxI, out=None, outcost=None):
        nx, ny, _, _ = Cv.dims
        out = out or self.new_image(nx, ny)
        outcost = outcost or self.new_image(nx, ny)
        self._chk(self.lib.mgm_wta_windowed_dev(self.h, Cv.h, NDIR, fix_overcount, refine.encode() if refine else None,
                                                dminI.h, dmaxI.h, out.h, outcost.h))
        return out, outcost

    def update_ranges_dev(self, outoff, dminI, dmaxI, slack=3, radius=2):
        self._chk(self.lib.mgm_update_ranges_dev(self.h, outoff.h, dminI.h, dmaxI.h, slack, radius))

    def median_dev(self, img, radius, out=None):
        nch, ny, nx = img.shape
        out = out or self.new_image(nx, ny, nch)
        self._chk(self.lib.mgm_median_dev(self.h, img.h, radius, out.h))
        return out

    def leftright_dev(self, d, other, tau, out=None):
        _, ny, nx = d.shape
        out = out or self.new_image(nx, ny)
        self._chk(self.lib.mgm_leftright_dev(self.h, d.h, other.h, tau, out.h))
        return out

    def backproject_dev(self, u, v, disp, out=None):
        nch, ny, nx = u.shape
        out = out or self.new_image(nx, ny, nch)
        self._chk(self.lib.mgm_backproject_dev(self.h, u.h, v.h, disp.h, out.h))
        return out

    def aggregate_batch_dev(self, Cvs, P1, P2, NDIR, MGM, use_fh=0, fix_overcount=1, w8s=None, refine=None, outs=None,
                            outcosts=None, want_S=False):
        """mgm() over several volumes of identical geometry in one pass launch: ([S...] or None, [out...], [outcost...])."""
        n = len(Cvs)
        nx, ny, _, _ = Cvs[0].dims
        outs = outs or [self.new_image(nx, ny) for _ in range(n)]
        outcosts = outcosts or [self.new_image(nx, ny) for _ in range(n)]
        arr = lambda hs: (C.c_void_p * n)(*hs)
        S = (C.c_void_p * n)()
        self._chk(self.lib.mgm_aggregate_batch_dev(
            self.h, n, arr([cv.h for cv in Cvs]), arr([w.h for w in w8s]) if w8s is not None else None, P1, P2, NDIR, MGM,
            use_fh, fix_overcount, refine.encode() if refine else None, arr([o.h for o in outs]),
            arr([o.h for o in outcosts]), S if want_S else None))
        return ([CostVolume(self, C.c_void_p(h)) for h in S] if want_S else None), outs, outcosts

    def aggregate(self, Cv, P1, P2, NDIR, MGM, use_fh=0, fix_overcount=1, w8=None, refine=None, want_S=True):
        """mgm(): returns (S or None, out, outcost) with host arrays for out/outcost."""
        nx, ny, _, _ = Cv.dims
        out = np.empty((ny, nx), np.float32)
        outc = np.empty((ny, nx), np.float32)
        S = C.c_void_p()
        w = _f32(w8) if w8 is not None else None
        self._chk(self.lib.mgm_aggregate(self.h, Cv.h, _ptr(w) if w is not None else None, P1, P2, NDIR, MGM, use_fh,
                                         fix_overcount, refine.encode() if refine else None, _ptr(out), _ptr(outc),
                                         C.byref(S) if want_S else None))
        return (CostVolume(self, S) if want_S else None), out, outc

    def debug_lr(self, Cv, p):
        nx, ny, dmin, dmax = Cv.dims
        out = np.empty((ny, nx, dmax - dmin + 1), np.float32)
        self._chk(self.lib.mgm_debug_download_lr(self.h, p, _ptr(out)))
        return out

    def refine(self, S, method, out, outcost):
        out = np.array(out, np.float32, copy=True)
        outcost = np.array(outcost, np.float32, copy=True)
        self._chk(self.lib.mgm_refine(self.h, S.h, method.encode(), _ptr(out), _ptr(outcost)))
        return out, outcost

    # ---- direction sharding (see mgm_amd/dist.py) ----
    def aggregate_passes_dev(self, Cv, P1, P2, MGM, use_fh, first_pass, n_passes, w8=None):
        self._chk(self.lib.mgm_aggregate_passes_dev(self.h, Cv.h, w8.h if w8 is not None else None, P1, P2, MGM, use_fh,
                                                    first_pass, n_passes))

    def aggregate_passes_at_dev(self, Cv, P1, P2, MGM, use_fh, first_pass, n_passes, slot0, n_slots, NDIR_total, w8=None):
        self._chk(self.lib.mgm_aggregate_passes_at_dev(self.h, Cv.h, w8.h if w8 is not None else None, P1, P2, MGM, use_fh,
                                                       first_pass, n_passes, slot0, n_slots, NDIR_total))

    def lr_device_ptr(self, slot):
        return self.lib.mgm_lr_device_ptr(self.h, slot)

    def wta_rows_dev(self, Cv, row0, nrows, lr_slabs_ptr, NDIR, fix_overcount, refine, out_ptr, outcost_ptr):
        self._chk(self.lib.mgm_wta_rows_dev(self.h, Cv.h, row0, nrows, lr_slabs_ptr, NDIR, fix_overcount,
                                            refine.encode() if refine else None, out_ptr, outcost_ptr))

    def set_placement_tries(self, tries):
        self._chk(self.lib.mgm_ctx_set_placement_tries(self.h, tries))

    def probe_workspace(self, nstreams=8):
        g = C.c_float()
        self._chk(self.lib.mgm_debug_probe_workspace(self.h, nstreams, C.byref(g)))
        return g.value

    def selftest_div3(self):
        n = C.c_ulonglong(0)
        self._chk(self.lib.mgm_selftest_div3(self.h, C.byref(n)))
        return n.value

    # ---- timing ----
    def timing(self, enable=True):
        self.lib.mgm_timing_enable(self.h, 1 if enable else 0)

    def timing_reset(self):
        self.lib.mgm_timing_reset(self.h)

    def timings(self):
        n = self.lib.mgm_timing_count(self.h)
        res = []
        for k in range(n):
            name, ms = C.c_char_p(), C.c_float()
            self._chk(self.lib.mgm_timing_get(self.h, k, C.byref(name), C.byref(ms)))
            res.append((name.value.decode(), ms.value))
        return res


def multi_plan(n, NDIR, ny):
    """mgm_multi_plan: [(first_pass, n_passes, row0, nrows)] per rank (no device needed)."""
    L = load_library()
    a = [(C.c_int * n)() for _ in range(4)]
    r = L.mgm_multi_plan(n, NDIR, ny, *a)
    if r:
        raise MgmError(r, "mgm_multi_plan: bad arguments")
    return [tuple(int(x[k]) for x in a) for k in range(n)]


class Multi:
    """n GPUs of one node behind one handle (``mgm_multi``): direction sharding of one volume with an RCCL slab exchange."""

    def __init__(self, device_ids):
        self.lib = load_library()
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = C.c_void_p()
        r = self.lib.mgm_multi_create(ids, len(device_ids), C.byref(h))
        if r:
            raise MgmError(r, "mgm_multi_create(%s) failed: %s" % (list(device_ids), self.lib.mgm_multi_last_error(None).decode()))
        self.h = h
        self.n = len(device_ids)
        self.ctx = [Context(_borrowed=self.lib.mgm_multi_ctx(h, k)) for k in range(self.n)]

    def transport(self):
        """"rccl", "peer" or "loopback" (mgm_multi_transport)."""
        return self.lib.mgm_multi_transport(self.h).decode()

    def aggregate_dev(self, Cvs, P1, P2, NDIR, MGM, use_fh, fix_overcount, refine, out, outc, w8s=None):
        """mgm_multi_aggregate into images on device 0 (returns when they are complete)."""
        arr = lambda hs: (C.c_void_p * self.n)(*hs)
        r = self.lib.mgm_multi_aggregate(self.h, arr([cv.h for cv in Cvs]), arr([w.h for w in w8s]) if w8s is not None else None, P1, P2,
                                         NDIR, MGM, use_fh, fix_overcount, refine.encode() if refine else None, out.h, outc.h)
        if r:
            raise MgmError(r, self.lib.mgm_multi_last_error(self.h).decode())

    def aggregate(self, Cvs, P1, P2, NDIR, MGM, use_fh=0, fix_overcount=1, refine=None, w8s=None):
        """mgm_multi_aggregate: returns host arrays (out, outcost)."""
        nx, ny, _, _ = Cvs[0].dims
        out, outc = self.ctx[0].new_image(nx, ny), self.ctx[0].new_image(nx, ny)
        self.aggregate_dev(Cvs, P1, P2, NDIR, MGM, use_fh, fix_overcount, refine, out, outc, w8s)
        res = out.download()[0], outc.download()[0]
        out.free(), outc.free()
        return res

    def close(self):
        if self.h:
            self.lib.mgm_multi_destroy(self.h)
            self.h = None
            for c in self.ctx:
                c.h = None
