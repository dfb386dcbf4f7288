"""A stand-in for mgm_amd.Context that computes NOTHING: bench.py's launcher, rendezvous, barrier / max-over-ranks timing,
extra legs and JSON contract can then be driven on CPU ranks (gloo) by tests/test_dist_cpu.py.  It is a test double, not a
CPU path of the product: bench.py does not know this file -- tests/run_bench_stub.py imports bench and hands it this class --
and bench.py labels the line `"data": "stub (no device work)"`."""
import os
import time


class _Handle:
    def __init__(self, shape=None):
        self.shape = shape

    def free(self):
        pass

    def update(self, a):
        pass


class StubContext:
    def __init__(self, device=0):
        self.device, self._timing, self._t = device, False, []

    def upload_image(self, a):
        return _Handle(getattr(a, "shape", None))

    def new_image(self, nx, ny, nch=1):
        return _Handle((nch, ny, nx))

    def costvolume_dev(self, u, v, dmin, dmax, prefilter="none", distance="ad", truncDist=float("inf"), census_win=3, into=None):
        if self._timing:
            self._t += [("k_census", 0.01), ("k_census", 0.01), ("k_cost", 0.1)]
        return into if into is not None else _Handle()

    def costvolume_ranged_dev(self, u, v, dminI, dmaxI, hull_min, hull_max, prefilter="none", distance="ad", truncDist=float("inf"),
                              census_win=3, into=None):
        return self.costvolume_dev(u, v, hull_min, hull_max, prefilter, distance, truncDist, census_win, into)

    def upload_volume(self, dense, dmin):
        return _Handle(getattr(dense, "shape", None))

    def update_ranges_dev(self, outoff, dminI, dmaxI, slack=3, radius=2):
        pass

    def wta_windowed_dev(self, Cv, NDIR, fix_overcount, refine, dminI, dmaxI, out=None, outcost=None):
        if self._timing:
            self._t += [("k_wta", 0.5)]
        return out, outcost

    def weights_dev(self, u, aP, aThresh, into=None):
        return into if into is not None else _Handle()

    def aggregate_batch_dev(self, Cvs, P1, P2, NDIR, MGM, use_fh=0, fix_overcount=1, w8s=None, refine=None, outs=None, outcosts=None,
                            want_S=False):
        time.sleep(0.002 * (1 + self.device))  # rank 1 is the slow one
        if os.environ.get("MGM_STUB_HANG_AT") == "cfg5" and len(Cvs) == 16 and self.device == 1:
            time.sleep(3600)  # (tests/test_dist_cpu.py: a rank that never comes back from the cfg5 leg)
        if self._timing:
            self._t += [("k_pass2", 1.0 * len(Cvs))] + [("k_wta", 0.5)] * len(Cvs)
        return None, outs, outcosts

    def set_pipeline(self, depth):
        pass

    def synchronize(self):
        pass

    def trim(self):
        pass

    def timing(self, enable=True):
        self._timing = bool(enable)

    def timing_reset(self):
        self._t = []

    def timings(self):
        return list(self._t)

    def close(self):
        pass
