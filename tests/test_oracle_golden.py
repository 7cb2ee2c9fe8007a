"""The oracle (oracle/lucille_oracle.c) against the committed golden vectors
that the COMPILED REFERENCE produced (tests/golden/make_golden.py).  Bit-exact:
prim ids, t/u/v as doubles, per-batch traversal counters and tree shape."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import load_golden


@pytest.mark.parametrize("name", ["soup_20k", "soup_3k_fat"])
def test_oracle_matches_reference_golden(name):
    g = load_golden(name)
    P, idx, org, dr = po.soup(int(g["ntri"]), int(g["nrays"]), float(g["half_extent"]), int(g["seed"]))
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    prim, t, u, v, cnt = o.intersect(org, dr, counters=True)
    assert np.array_equal(prim, g["prim"])
    assert np.array_equal(t, g["t"]) and np.array_equal(u, g["u"]) and np.array_equal(v, g["v"])
    assert [cnt[k] for k in ("ninner", "nleaf", "ntested", "nhit", "nrays")] == list(map(int, g["counters"]))
    ts = o.tree_stats()
    assert [ts[k] for k in ("ninner", "nleaf", "max_depth", "max_leaf_tris", "ntriangles")] == list(map(int, g["tree"]))
    bmin, bmax = o.bbox()
    assert np.array_equal(bmin, g["bmin"]) and np.array_equal(bmax, g["bmax"])


def test_oracle_threads_agree():
    P, idx, org, dr = po.soup(5000, 4000)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    a = o.intersect(org, dr, nthreads=1); b = o.intersect(org, dr, nthreads=5)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_oracle_tree_independent_of_bvh():
    """the reference's result equals a brute-force arg-min with the
    triangle_isect arithmetic, so any conservative BVH reproduces it."""
    P, idx, org, dr = po.soup(3000, 1500, 0.02)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    a = o.intersect(org, dr); b = o.brute_force(org, dr, nthreads=4)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_oracle_empty_scene_always_misses():
    o = po.Oracle(); o.add_mesh(np.zeros((0, 3)), np.zeros(0, np.uint32)); o.build()
    prim, t, u, v = o.intersect([[0, 0, 0]], [[0, 0, 1]])
    assert prim[0] == po.MISS and t[0] == 1.0e38


def test_oracle_known_answer_two_triangles():
    """the survey's hand-checkable case: a quad at z=5 hit at t=5"""
    P = np.array([[-1, -1, 5], [1, -1, 5], [1, 1, 5], [-1, 1, 5]], np.float64)
    o = po.Oracle(); o.add_mesh(P, [0, 1, 2, 0, 2, 3]); o.build()
    prim, t, u, v = o.intersect([[0.1, -0.5, 0.0]], [[0.0, 0.0, 1.0]])
    assert prim[0] == 0 and t[0] == 5.0
    # barycentrics: P = v0 + u e1 + v e2 with e1=(2,0,0), e2=(2,2,0)
    assert abs(u[0] * 2 + v[0] * 2 - 1.1) < 1e-15 and abs(v[0] * 2 - 0.5) < 1e-15
