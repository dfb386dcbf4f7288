"""mgm_ctx_set_placement_tries: the context times the pass launch of an aggregation that has just (re)allocated its workspace on
several physical placements and keeps the fastest -- the RESULTS must not know (same inputs, the launch repeated)."""
import numpy as np
import pytest

import mgm_amd
from helpers import ndiff
from mgm_amd import synth

pytestmark = pytest.mark.gpu


def test_placement_tries_do_not_change_results():
    nx, ny, L = 640, 512, 128  # 8 passes x 640x512x128 floats = 1.3 GB of Lr volumes: above the 256 MB the tuning starts at
    u, v, _ = synth.stereo_pair(nx, ny, -96, 0, seed=21)
    res = []
    for tries in (0, 3):
        with mgm_amd.Context(0) as ctx:
            ctx.set_placement_tries(tries)
            du, dv = ctx.upload_image(u), ctx.upload_image(v)
            cv = ctx.costvolume_dev(du, dv, -(L - 1), 0, "none", "census", float("inf"), 5)
            _, o, c = ctx.aggregate_dev(cv, 2.0, 20000.0, 8, 3, 1, 1, None, "vfit")
            a = (o.download(), c.download())
            _, o2, c2 = ctx.aggregate_dev(cv, 2.0, 20000.0, 8, 3, 1, 1, None, "vfit")  # (the placed workspace, a second launch)
            assert ndiff(a[0], o2.download()) == 0 and ndiff(a[1], c2.download()) == 0
            res.append(a)
    assert ndiff(res[0][0], res[1][0]) == 0 and ndiff(res[0][1], res[1][1]) == 0
    with pytest.raises(mgm_amd.MgmError):
        with mgm_amd.Context(0) as ctx:
            ctx.set_placement_tries(99)
