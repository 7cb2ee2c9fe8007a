"""The oracle against the LIVE compiled reference (oracle/_ref), where that
library exists (it is built from /root/reference by oracle/Makefile and travels
to the GPU box as a .so).  Fresh seeds, so this is not a replay of the goldens."""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(not po.ref_available(stat=True), reason="oracle/_ref not built (no /root/reference)")


@pytest.fixture(scope="module")
def ref():
    return po.RefLib(stat=True)


@pytest.mark.parametrize("ntri,nrays,he,seed", [(40000, 30000, 0.005, 12345), (800, 20000, 0.08, 99), (17, 5000, 0.3, 7),
                                                 (1, 2000, 0.5, 3)])
def test_bit_identical_to_reference(ref, ntri, nrays, he, seed):
    P, idx, org, dr = po.soup(ntri, nrays, he, seed)
    ref.reset(); ref.add_mesh(P, idx); ref.build()
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    a = o.intersect(org, dr, counters=True)
    b = ref.intersect(org, dr, counters=True)
    for k in range(4):
        assert np.array_equal(a[k], b[k])
    assert a[4] == b[4]
    assert o.tree_stats() == ref.tree_stats()


def test_multi_mesh_prim_numbering(ref):
    """primitive id = running index over geom_list order then triangle order
    (create_triangle_list, bvh.c:1792-1821)"""
    P1, i1, org, dr = po.soup(300, 4000, 0.1, 11)
    P2, i2, _, _ = po.soup(500, 1, 0.1, 22)
    ref.reset(); ref.add_mesh(P1, i1); ref.add_mesh(P2, i2); ref.build()
    o = po.Oracle(); o.add_mesh(P1, i1); o.add_mesh(P2, i2); o.build()
    a = o.intersect(org, dr); b = ref.intersect(org, dr)
    for k in range(4):
        assert np.array_equal(a[k], b[k])
    assert (a[0][a[0] != po.MISS] >= 300).any() and (a[0] < 300).any()


def test_survey_check_values_soup_200k(ref):
    """a size the CPU suite can afford of the SURVEY Appendix C generator"""
    P, idx, org, dr = po.soup(200000, 50000)
    ref.reset(); ref.add_mesh(P, idx); ref.build()
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    a = o.intersect(org, dr, nthreads=4); b = ref.intersect(org, dr)
    for k in range(4):
        assert np.array_equal(a[k], b[k])
