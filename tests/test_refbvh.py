"""Host logic: the product's REFERENCE-ORDER tree (lucille_amd/csrc/lh_refbvh.c) must be the
reference's tree -- same shape, same leaves, same triangle order inside every leaf -- because
beam visibility and exact-t tie winners depend on exactly that (bvh.c:780,850,1080,2435-2542)."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import Model, assert_hits_equal, grid_mesh, load_golden


@pytest.mark.parametrize("ntri,he,seed,threads", [(20000, 0.005, po.SOUP_SEED, 1), (20000, 0.005, po.SOUP_SEED, 6),
                                                   (120000, 0.004, 5, 8), (300, 0.2, 9, 2), (16, 0.3, 1, 1), (17, 0.3, 1, 1)])
def test_same_tree_as_the_reference(ntri, he, seed, threads):
    P, idx, _, _ = po.soup(ntri, 1, he, seed)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    m = Model(P, idx)
    info = m.ref_build(nthreads=threads, use_for_ties=False)
    ts = o.tree_stats()
    assert (info["ninner"], info["nleaf"], info["max_depth"]) == (ts["ninner"], ts["nleaf"], ts["max_depth"])
    lp, pf = m.ref_leaf_order(); olp, opf = o.leaf_order()
    assert np.array_equal(lp, olp)            # the reference's leaf-sorted triangle order, exactly
    assert np.array_equal(pf, opf)            # and the same leaf boundaries
    bmin, bmax = m.ref_bbox(); obmin, obmax = o.bbox()
    assert np.array_equal(bmin, obmin) and np.array_equal(bmax, obmax)


def test_golden_tree_shape():
    g = load_golden("soup_20k")               # tree stats recorded from the compiled reference
    P, idx, _, _ = po.soup(int(g["ntri"]), 1, float(g["half_extent"]), int(g["seed"]))
    m = Model(P, idx)
    info = m.ref_build(use_for_ties=False)
    assert [info["ninner"], info["nleaf"], info["max_depth"]] == list(map(int, g["tree"][:3]))


def test_exact_t_ties_follow_the_reference():
    """rays through shared vertices / edges / quad diagonals of an axis-aligned grid: many
    bit-equal t.  With the reference-order tree the model returns the REFERENCE's winner."""
    P, idx = grid_mesh(8, 8)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    m = Model(P, idx)
    m.ref_build(use_for_ties=True)
    try:
        xs = np.linspace(0.0, 1.0, 33); tx, ty = np.meshgrid(xs, xs)
        tgt = np.stack([tx.ravel(), ty.ravel(), np.zeros(tx.size)], 1)
        rng = np.random.default_rng(0)
        nties = 0
        for oz in (1.0, 37.5, -2.0):
            org = np.tile(np.array([[0.3, 0.45, oz]]), (tgt.shape[0], 1)) + rng.uniform(-0.2, 0.2, (tgt.shape[0], 3)) * [1, 1, 0]
            for dr in (tgt - org, (tgt - org) / np.linalg.norm(tgt - org, axis=1, keepdims=True)):
                exp = o.intersect(org, dr)
                got, _ = m.trace(org, dr)
                nties += int((o.count_equal_t(org, dr, exp[1]) >= 2).sum())
                assert_hits_equal(got, exp, "grid oz=%g" % oz)      # bit-exact INCLUDING the tie rays
        org = np.stack([tx.ravel(), ty.ravel(), np.ones(tx.size)], 1)
        dr = np.tile(np.array([[0.0, 0.0, -1.0]]), (org.shape[0], 1))
        exp = o.intersect(org, dr); got, _ = m.trace(org, dr)
        nties += int((o.count_equal_t(org, dr, exp[1]) >= 2).sum())
        assert_hits_equal(got, exp, "axis-parallel")
        assert nties > 100                                          # the test really exercises ties
    finally:
        Model.ref_off()


def test_parallel_top_levels_keep_lucilles_tree():
    """above 2^17 primitives the top of lucille's tree is binned / partitioned by a thread pool (lh_tpool.h): integer histograms,
    min / max bounds and a partition whose layout follows from per-chunk counts -- the tree and the leaf order (= the tie rule's
    input) must equal the single-thread build's and the oracle's, bit for bit"""
    from lucille_amd import scenes
    P, idx, _ = scenes.soup_triangles(400000, 0.007)
    a = Model(P, idx, nthreads=8); a.ref_build(nthreads=1)
    b = Model(P, idx, nthreads=8); b.ref_build(nthreads=8)
    la_, lb_ = a.ref_leaf_order(), b.ref_leaf_order()
    assert np.array_equal(la_[0], lb_[0]) and np.array_equal(la_[1], lb_[1])
    assert all(np.array_equal(x, y) for x, y in zip(a.ref_bbox(), b.ref_bbox()))
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    ol = o.leaf_order()
    assert np.array_equal(ol[0], lb_[0]) and np.array_equal(ol[1], lb_[1])
