"""Host logic: the kernel ALGORITHM (fp32 conservative traversal + tolerance-
carrying fp32 Moeller-Trumbore filter + fp64 resolve), run on the CPU through
tests/cpu_model/ with the arithmetic shared with the HIP kernel
(lucille_amd/csrc/lh_filter.h), must reproduce the oracle bit for bit:
the filter may never lose a hit the fp64 reference finds."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import Model, assert_hits_equal, grid_mesh, load_golden, random_rays


def check(P, idx, org, dr, what, ties_possible=False):
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=4)
    m = Model(P, idx)
    got, cnt = m.trace(org, dr)
    if ties_possible:
        # exact-t ties (a ray through a shared edge/vertex): the reference's winner
        # depends on ITS tree's leaf order (bvh.c:780,850) -- t must still agree bit
        # for bit, the winner must be one of the tied triangles, and every
        # non-tied ray must be bit-exact
        tie = o.count_equal_t(org, dr, exp[1]) >= 2
        assert np.array_equal(got[1], exp[1]), what + ": t differs"
        assert ((got[0] == po.MISS) == (exp[0] == po.MISS)).all()
        bf = o.brute_force(org[tie], dr[tie])
        assert np.array_equal(got[0][tie], bf[0]), what + ": tie rule (largest prim id) not honoured"
        got = tuple(g[~tie] for g in got); exp = tuple(e[~tie] for e in exp)
    assert_hits_equal(got, exp, what)
    if ties_possible:
        return cnt
    occ, _ = m.trace(org, dr, anyhit=True)
    assert np.array_equal(occ.astype(bool), exp[0] != po.MISS), what + ": any-hit != (closest hit exists)"
    return cnt


@pytest.mark.parametrize("name", ["soup_20k", "soup_3k_fat"])
def test_model_matches_reference_goldens(name):
    g = load_golden(name)
    P, idx, org, dr = po.soup(int(g["ntri"]), int(g["nrays"]), float(g["half_extent"]), int(g["seed"]))
    m = Model(P, idx)
    got, _ = m.trace(org, dr)
    assert_hits_equal(got, (g["prim"], g["t"], g["u"], g["v"]), name)


@pytest.mark.parametrize("ntri,he,seed", [(100000, 0.005, 1), (20000, 0.0005, 2), (500, 0.2, 3), (3, 0.4, 4)])
def test_model_matches_oracle_soups(ntri, he, seed):
    P, idx, org, dr = po.soup(ntri, 60000, he, seed)
    check(P, idx, org, dr, "soup %d" % ntri)


@pytest.mark.parametrize("ntri,he,seed", [(100000, 0.005, 1), (20000, 0.0005, 2), (500, 0.2, 3), (3, 0.4, 4), (1, 0.4, 5)])
def test_eight_wide_model_matches_oracle(ntri, he, seed):
    """the 8-wide 16-bit-grid tree (lh_q8node_t: what ray dumps walk on scenes larger than the Infinity Cache) in octant order:
    the same records as the oracle, and fewer node visits than the 4-wide tree"""
    P, idx, org, dr = po.soup(ntri, 60000, he, seed)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=4)
    m = Model(P, idx)
    got, c8 = m.trace(org, dr, qnodes=4)
    assert_hits_equal(got, exp, "8-wide model, soup %d" % ntri)
    occ, _ = m.trace(org, dr, anyhit=True, qnodes=4)
    assert np.array_equal(occ.astype(bool), exp[0] != po.MISS)
    _, c4 = m.trace(org, dr, qnodes=2)
    assert c8["nodes"] <= c4["nodes"] and c8["tris"] <= 1.2 * c4["tris"] + 10


def test_axis_aligned_grid_vertex_edge_diagonal_rays():
    """rays aimed exactly at shared vertices, edges and quad diagonals of an
    axis-aligned plane (zero-thickness boxes): the worst case for fp32 culling"""
    P, idx = grid_mesh(8, 8)
    xs = np.linspace(0.0, 1.0, 33)      # every 1/4 cell: vertices, edge midpoints, diagonals
    tx, ty = np.meshgrid(xs, xs)
    tgt = np.stack([tx.ravel(), ty.ravel(), np.zeros(tx.size)], 1)
    rng = np.random.default_rng(0)
    for oz in (1.0, 37.5, 1e-3):
        org = np.tile(np.array([[0.3, 0.45, oz]]), (tgt.shape[0], 1)) + rng.uniform(-0.2, 0.2, (tgt.shape[0], 3)) * [1, 1, 0]
        dr = tgt - org
        check(P, idx, org, dr, "grid targets oz=%g" % oz, ties_possible=True)   # unnormalised dir on purpose
        check(P, idx, org, dr / np.linalg.norm(dr, axis=1, keepdims=True), "grid targets normalised", ties_possible=True)
    # straight-down rays: two zero direction components
    org = np.stack([tx.ravel(), ty.ravel(), np.ones(tx.size)], 1)
    dr = np.tile(np.array([[0.0, 0.0, -1.0]]), (org.shape[0], 1))
    check(P, idx, org, dr, "axis-parallel rays", ties_possible=True)


def test_far_origin_and_large_coordinates():
    rng = np.random.default_rng(5)
    P, idx, _, _ = po.soup(5000, 1, 0.02, 77)
    P2 = P * 50.0 + 1000.0                      # scene far from the origin: big |coordinates|
    org, dr = random_rays(rng, 20000)
    org2 = org * 50.0 + 1000.0
    check(P2, idx, org2, dr, "offset scene")
    # origins thousands of scene-sizes away, aimed at the scene
    far = rng.normal(size=(5000, 3)); far = far / np.linalg.norm(far, axis=1, keepdims=True) * 3000.0 + 0.5
    tgt = rng.uniform(0, 1, (5000, 3))
    check(P, idx, far, tgt - far, "far origins")


def test_rays_starting_on_surfaces_like_ao_rays():
    """AO rays start 1e-6 above the surface they came from (ambientocclusion.c:65-73):
    t ~ 0 self-hits must be decided by the fp64 resolve, not the filter"""
    P, idx = grid_mesh(4, 4)
    rng = np.random.default_rng(9)
    n = 20000
    o = np.stack([rng.uniform(0, 1, n), rng.uniform(0, 1, n), np.full(n, 1e-6)], 1)
    d = rng.normal(size=(n, 3)); d[:, 2] = np.abs(d[:, 2]); d /= np.linalg.norm(d, axis=1, keepdims=True)
    check(P, idx, o, d, "above surface")
    o[:, 2] = 0.0                                # exactly ON the plane: t == 0 is accepted by the reference
    check(P, idx, o, d, "on surface")
    o[:, 2] = -1e-9
    check(P, idx, o, d, "just below")


def test_counters_definition():
    P, idx, org, dr = po.soup(20000, 5000, 0.005, 8)
    cnt = check(P, idx, org, dr, "counters")
    assert cnt["rays"] == 5000 and cnt["nodes"] >= 5000 and cnt["tris"] > 0 and cnt["exact"] <= cnt["tris"]
