"""The product's one-ray host walk (lucille_amd/csrc/lh_hostwalk.c: what lh_accel_intersect1 answers with on the calling thread when
the scene's trees live on the host) against the oracle, without a device: the file is compiled as it is into the test library
(tests/helpers.build_model) and walked ray by ray.  Records bit for bit -- soups, lucille's example scene with rays aimed at
vertices and edges (exact-t ties, fragile hits: lucille's own tree decides), zero-area segments, direction components beyond
deg_dcap (the reference's own walk decides), an empty scene."""
import numpy as np
import pytest

from lucille_amd import scenes
from oracle import pyoracle as po
from tests.helpers import Model, assert_hits_equal, load_golden


def oracle_of(P, idx, org, dr):
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    return o.intersect(org, dr, nthreads=8)


@pytest.mark.parametrize("ntri,he,seed", [(1, 0.3, 1), (7, 0.2, 2), (3000, 0.03, 3), (120000, 0.006, 4)])
def test_soups(ntri, he, seed):
    P, idx, org, dr = po.soup(ntri, 40000, he, 500 + seed)
    m = Model(P, idx); m.ref_build()
    try:
        assert_hits_equal(m.hostwalk(org, dr), oracle_of(P, idx, org, dr), "soup %d" % ntri)
    finally:
        Model.ref_off()


def test_example_scene_vertices_edges_and_unnormalised_directions():
    g = load_golden("ao_c1")
    P = np.concatenate([g["pos%d" % k][:, :3] for k in range(int(g["ngeoms"]))])
    off = np.cumsum([0] + [g["pos%d" % k].shape[0] for k in range(int(g["ngeoms"]))])
    idx = np.concatenate([g["idx%d" % k].astype(np.uint32) + np.uint32(off[k]) for k in range(int(g["ngeoms"]))])
    rng = np.random.default_rng(11); n = 60000
    T = P[idx.astype(np.int64)].reshape(-1, 3, 3); pick = rng.integers(0, T.shape[0], n)
    w = rng.random((n, 3)); w /= w.sum(1, keepdims=True)
    tgt = (T[pick] * w[:, :, None]).sum(1)
    tgt[:15000] = T[pick[:15000], rng.integers(0, 3, 15000)]                       # exactly a vertex
    tgt[15000:30000] = 0.5 * (T[pick[15000:30000], 0] + T[pick[15000:30000], 1])    # exactly on an edge
    org = tgt + rng.normal(size=(n, 3)) * 4.0
    dr = (tgt - org) * rng.uniform(0.01, 50.0, (n, 1))                             # t is in units of |dir| (ray.h:22-68)
    ok = np.abs(dr[:, 1]) > 1e-14
    org, dr = np.ascontiguousarray(org[ok]), np.ascontiguousarray(dr[ok])
    m = Model(P, idx); m.ref_build()
    try:
        exp = oracle_of(P, idx, org, dr)
        assert (exp[0] != po.MISS).mean() > 0.9
        assert_hits_equal(m.hostwalk(org, dr), exp, "example scene")
    finally:
        Model.ref_off()


def test_zero_area_segments_and_big_directions():
    g = load_golden("ao_c1")
    P, idx = scenes.tessellate(g["pos0"], g["idx0"], 6)            # the cone: 40 960 zero-area triangles on ten segments, outside the tree
    T = P[idx.astype(np.int64)].reshape(-1, 3, 3)
    zero = (T[:, 0] == T[:, 1]).all(1) | (T[:, 0] == T[:, 2]).all(1) | (T[:, 1] == T[:, 2]).all(1)
    rng = np.random.default_rng(5); n = 30000
    seg = T[zero][rng.integers(0, zero.sum(), n)]
    tgt = seg[:, 0] + (seg[:, 2] - seg[:, 0]) * rng.random((n, 1)) + (seg[:, 1] - seg[:, 0]) * rng.random((n, 1))
    org = tgt + rng.normal(size=(n, 3)) * 3.0
    dr = tgt - org; dr[n // 2:] += rng.normal(size=(n - n // 2, 3)) * 0.02
    ok = np.abs(dr[:, 1]) > 1e-14
    org, dr = np.ascontiguousarray(org[ok]), np.ascontiguousarray(dr[ok])
    m = Model(P, idx); m.ref_build()
    try:
        assert_hits_equal(m.hostwalk(org, dr), oracle_of(P, idx, org, dr), "cone, rays at the segments")
        big = dr * 4096.0                                            # beyond deg_dcap = 1024: lucille's own walk on its own tree decides
        assert_hits_equal(m.hostwalk(org, big), oracle_of(P, idx, org, big), "cone, direction components beyond 1024")
    finally:
        Model.ref_off()


def test_empty_scene_always_misses():
    m = Model(np.zeros((0, 3)), np.zeros(0, np.uint32))
    prim, t, u, v = m.hostwalk(np.zeros((5, 3)), np.ones((5, 3)), use_ref=False)
    assert (prim == po.MISS).all() and (t == 1.0e38).all() and (u == 0).all() and (v == 0).all()


def test_zero_area_triangles_that_stay_in_the_tree():
    """zero-area triangles too large to be left out (v1 == v2 at 15 units; three different points on one line): the reference reports
    them by its determinant's rounding noise for rays that reach their leaf in ITS tree with large direction components -- rays
    beyond deg_dcap = 1 / (|e1|_1 |e2|_1) are its own walk's (lh_bvh.c tri_zero_area_s2; found by tools/fuzz_parity.py)"""
    rng = np.random.default_rng(3)
    c = rng.uniform(0, 1, (60, 1, 3)); T = (c + rng.normal(size=(60, 3, 3)) * 0.3) * 28.0
    T[::3, 2] = T[::3, 1]
    T[1::5, 2] = T[1::5, 0] + 2.0 * (T[1::5, 1] - T[1::5, 0])
    P = T.reshape(-1, 3).copy(); idx = np.arange(180, dtype=np.uint32)
    n = 40000; pick = rng.integers(0, 60, n)
    w = rng.random((n, 3)); w /= w.sum(1, keepdims=True); tgt = (T[pick] * w[:, :, None]).sum(1)
    org = tgt + rng.normal(size=(n, 3)) * 28.0 * 30.0
    dr = (tgt - org) * rng.uniform(0.001, 1000.0, (n, 1))
    dr[-n // 4:] /= np.abs(dr[-n // 4:]).max(1, keepdims=True) * rng.uniform(1.0, 400.0, (n // 4, 1))
    ok = np.abs(dr[:, 1]) > 1e-14 * np.abs(dr).max(1)
    org, dr = np.ascontiguousarray(org[ok]), np.ascontiguousarray(dr[ok])
    exp = oracle_of(P, idx, org, dr)
    za = np.zeros(60, bool); za[::3] = True; za[1::5] = True
    assert ((exp[0] != po.MISS) & za[np.minimum(exp[0], 59)]).sum() > 10
    m = Model(P, idx); m.ref_build()
    try:
        assert_hits_equal(m.hostwalk(org, dr), exp, "zero-area triangles in the tree")
    finally:
        Model.ref_off()


@pytest.mark.parametrize("name,rays,prim", [("fuzz_r06_f662_99", (1156, 6023, 6936), 29), ("fuzz_r06_f661_359", (), None)])
def test_near_collinear_triangles_of_round_6_fuzz(name, rays, prim):
    """two scenes of tools/fuzz_parity.py's kind 9 (seeds 662 / 661, rounds 99 / 359; saved with FUZZ_SAVE=..., the first 8000 / 4000 rays):
    triangles whose three points were PUT on a line and then scaled and shifted -- collinear up to the rounding of coordinates of 8 and
    9 600 units, max |n_k| = 3e-15 |e1|_1 |e2|_1, beyond round 5's 8.9e-16.  The reference reports them by its determinant's noise at a t
    BEFORE their box, behind a nearer triangle's hit (rays 1156, 6023, 6936 of the first scene: three rays aimed at a vertex and an edge
    of triangle 29).  lh_bvh.h lh_zero_area_weight now counts them among the triangles that bound deg_dcap"""
    z = load_golden(name); P, idx, org, dr = z["P"], z["idx"], z["org"], z["dr"]
    exp = oracle_of(P, idx, org, dr)
    for r in rays:
        assert exp[0][r] == prim
    m = Model(P, idx); m.ref_build()
    try:
        assert_hits_equal(m.hostwalk(org, dr), exp, name)
    finally:
        Model.ref_off()

