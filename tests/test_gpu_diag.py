"""Per-ray traversal diagnostics: ri_bvh_intersect zeroes and fills a caller's ri_bvh_diag_t through
`user` (/root/reference/src/render/bvh.c:451-456, bvh.h:103-110; the testbed's heat maps, simplerender.cpp:202-218).
ri_hipbvh_intersect used to drop it.  Now: the same three numbers for this build's tree, per ray, from the sequential walk --
compared here with the host model of that walk (tests/cpu_model), with the launch totals of the statistics switch, and
through the plain-C mirror."""
import numpy as np
import pytest

import lucille_amd as la
from oracle import pyoracle as po
from tests.helpers import Model, assert_hits_equal

pytestmark = pytest.mark.gpu


def test_per_ray_counts_equal_the_host_model():
    P, idx, org, dr = po.soup(60000, 30000, 0.01, 21)
    org[::5] = org[::5] * 3.0 - 1.0                                   # some rays start outside the scene box
    m = Model(P, idx, nthreads=4)
    exp_hits, exp_diag = m.trace_diag(org, dr, qnodes=2)              # the 4-wide 16-bit grid walk, nearest child first
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build="host")
    hits, diag = acc.intersect_diag(org, dr)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    assert_hits_equal(hits, o.intersect(org, dr, nthreads=8), "diag path records")
    assert_hits_equal(exp_hits, hits, "model records")
    assert np.array_equal(diag[:, :3], exp_diag[:, :3]), np.nonzero((diag[:, :3] != exp_diag[:, :3]).any(1))[0][:10]
    assert np.array_equal(diag[:, 3], exp_diag[:, 3])                 # fp64 tests too (same pending-list schedule)
    assert diag[:, 0].min() >= 0 and diag[:, 0].max() > 20 and (diag[:, 1] <= diag[:, 2]).all()
    # totals: the statistics switch counts the same launch -- its sums are the sums of the rows
    acc.trace_statistics(True)
    acc.statistics(clear=True)
    hits2, diag2 = acc.intersect_diag(org[:5000], dr[:5000])
    st = acc.statistics()
    assert st["rays"] == 5000 and st["nodes"] == int(diag2[:, 0].sum()) and st["tris"] == int(diag2[:, 2].sum()) and st["exact"] == int(diag2[:, 3].sum())
    assert st["hits"] == int((hits2[0] != po.MISS).sum())
    acc.trace_statistics(False)
    # one ray at a time: the same rows
    for i in (0, 1, 7, 123):
        _, d1 = acc.intersect_diag(org[i:i + 1], dr[i:i + 1])
        assert np.array_equal(d1[0], diag[i])
    acc.close()
    # a device-built tree: records unchanged, counts those of ITS tree (plausible, not the host tree's)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build="device"); acc.wait_exact()
    hits3, diag3 = acc.intersect_diag(org, dr)
    assert_hits_equal(hits3, hits, "device-built tree")
    assert 0.7 < diag3[:, 0].sum() / diag[:, 0].sum() < 1.4
    acc.close()
    # an empty scene: zeros
    e = la.HipAccel(0); e.commit()
    h0, d0 = e.intersect_diag(org[:10], dr[:10])
    assert (h0[0] == po.MISS).all() and not d0.any()
    e.close()
