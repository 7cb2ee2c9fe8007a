"""Hits the reference can miss (lh_reftrace.h): the reference returns the closest triangle its traversal
REACHES; its exact-fp64 box test can fail by rounding for a hit on an edge/corner of a box of its own tree
(a ray aimed exactly at a vertex of an isolated triangle), and then it misses a hit its own triangle test
accepts.  Such hits are flagged (lh_hit_fragile) and re-traced with the reference's own walk."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import Model, assert_hits_equal, chain_scene, vertex_aimed_rays


def _oracle(P, idx):
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    return o


@pytest.mark.parametrize("n", [40, 120, 400])
def test_model_answers_what_the_reference_answers_on_box_corner_hits(n):
    P, idx = chain_scene(n)
    org, dr = vertex_aimed_rays(np.random.default_rng(n), P, idx, 6000)
    o = _oracle(P, idx)
    exp = o.intersect(org, dr)
    bf = o.brute_force(org, dr, nthreads=4)
    lost = int(((bf[0] != po.MISS) & (exp[0] == po.MISS)).sum())
    m = Model(P, idx, nthreads=1); m.ref_build(nthreads=1, use_for_ties=True)
    try:
        for q in (4, 2, 0):
            got, _ = m.trace(org, dr, qnodes=q, nthreads=1)
            assert_hits_equal(got, exp, "chain %d fmt %d" % (n, q))
            occ, _ = m.trace(org, dr, anyhit=True, qnodes=q, nthreads=1)
            assert np.array_equal(occ.astype(bool), exp[0] != po.MISS)
    finally:
        Model.ref_off()
    if n <= 120:
        assert lost > 0, "the scene is meant to contain hits the reference's traversal loses"


def test_without_the_reference_tree_the_fast_path_returns_the_brute_force_answer():
    """LH_REFTREE=0 semantics (no retrace): closest among ALL triangles passing triangle_isect"""
    P, idx = chain_scene(40)
    org, dr = vertex_aimed_rays(np.random.default_rng(1), P, idx, 4000)
    o = _oracle(P, idx)
    bf = o.brute_force(org, dr, nthreads=4)
    m = Model(P, idx, nthreads=1)
    got, _ = m.trace(org, dr, qnodes=2, nthreads=1)
    assert_hits_equal(got, bf, "no reference tree")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [40, 120, 400])
def test_hip_answers_what_the_reference_answers_on_box_corner_hits(n):
    """also the deep-tree path: from n = 120 the 4-wide walk's stack bound exceeds the LDS rows and the
    launch falls back to the 2-wide walk"""
    import torch
    import lucille_amd as la
    P, idx = chain_scene(n)
    org, dr = vertex_aimed_rays(np.random.default_rng(n), P, idx, 20000)
    exp = _oracle(P, idx).intersect(org, dr, nthreads=8)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
    o_ = torch.from_numpy(org).cuda(); d_ = torch.from_numpy(np.ascontiguousarray(dr)).cuda()
    for variant in (la.VARIANT_DEFAULT, la.VARIANT_DIRECT):
        out = acc.intersect_device(o_, d_, variant=variant)
        occ = acc.intersect_device(o_, d_, mode=la.MODE_ANY, variant=variant)[0]
        torch.cuda.synchronize()
        got = (out[0].cpu().numpy().view(np.uint32), out[1].cpu().numpy(), out[2].cpu().numpy(), out[3].cpu().numpy())
        assert_hits_equal(got, exp, "chain %d variant %d" % (n, variant))
        assert np.array_equal(occ.cpu().numpy().astype(bool), exp[0] != po.MISS)
    assert_hits_equal(acc.intersect_host(org[:3000], dr[:3000]), tuple(e[:3000] for e in exp), "host path")


@pytest.mark.gpu
def test_hip_vertex_aimed_rays_on_a_dense_soup():
    import torch
    import lucille_amd as la
    P, idx, _, _ = po.soup(100000, 1, 0.01, 5)
    org, dr = vertex_aimed_rays(np.random.default_rng(9), P, idx, 200000, spread=1.0)
    exp = _oracle(P, idx).intersect(org, dr, nthreads=16)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
    out, cnt = acc.intersect_device(torch.from_numpy(org).cuda(), torch.from_numpy(np.ascontiguousarray(dr)).cuda(), counters=True)
    torch.cuda.synchronize()
    got = (out[0].cpu().numpy().view(np.uint32), out[1].cpu().numpy(), out[2].cpu().numpy(), out[3].cpu().numpy())
    assert_hits_equal(got, exp, "soup, vertex-aimed")
