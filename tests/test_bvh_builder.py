"""Host logic: the product's BVH builder (lucille_amd/csrc/lh_bvh.c) -- structural
invariants of the flattened tree the kernels traverse."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import Model, grid_mesh

EMPTY = -(2 ** 31)


def walk(nodes, tri32, tri_dbl):
    """returns (visited prim ids, max depth); checks child boxes bound their subtree"""
    refs = nodes[:, 12:14].view(np.int32)
    seen = []
    maxd = 0

    def bounds_of(ref, depth):
        nonlocal maxd
        maxd = max(maxd, depth)
        if ref == EMPTY:
            return None
        if ref < 0:
            x = (~ref) & 0xFFFFFFFF
            first, cnt = x >> 2, (x & 3) + 1
            prims = tri32[first:first + cnt, 9].view(np.uint32)
            seen.extend(int(p) for p in prims)
            tv = tri_dbl[prims].reshape(-1, 3)
            return tv.min(0), tv.max(0)
        n = nodes[ref]
        out = []
        for c, (lo, hi) in enumerate(((n[0:3], n[3:6]), (n[6:9], n[9:12]))):
            b = bounds_of(int(refs[ref, c]), depth + 1)
            if b is None:
                assert lo[0] > hi[0], "empty child must carry an inverted box"
                continue
            assert (lo.astype(np.float64) <= b[0]).all() and (hi.astype(np.float64) >= b[1]).all(), \
                "fp32 child box must contain its fp64 triangles (outward rounding)"
            out.append(b)
        return np.min([o[0] for o in out], 0), np.max([o[1] for o in out], 0)

    bounds_of(0, 0)
    return seen, maxd


@pytest.mark.parametrize("ntri,he", [(1, 0.2), (2, 0.2), (5, 0.1), (1000, 0.02), (30000, 0.005)])
def test_every_triangle_in_exactly_one_leaf(ntri, he):
    P, idx, _, _ = po.soup(ntri, 1, he, 4242)
    m = Model(P, idx)
    nodes, tri32 = m.nodes(), m.tri32()
    tri_dbl = P[idx].reshape(-1, 9)
    seen, maxd = walk(nodes, tri32, tri_dbl)
    assert sorted(seen) == list(range(ntri))
    assert m.ntris == ntri and maxd <= m.max_depth <= 60
    # filter record = fp64 differences rounded once
    prims = tri32[:, 9].view(np.uint32)
    v = tri_dbl[prims].reshape(-1, 3, 3)
    assert np.array_equal(tri32[:, 0:3], v[:, 0].astype(np.float32))
    assert np.array_equal(tri32[:, 3:6], (v[:, 1] - v[:, 0]).astype(np.float32))
    assert np.array_equal(tri32[:, 6:9], (v[:, 2] - v[:, 0]).astype(np.float32))
    assert (tri32[:, 10] >= np.linalg.norm(v[:, 1] - v[:, 0], axis=1)).all()
    assert (tri32[:, 11] >= np.linalg.norm(v[:, 2] - v[:, 0], axis=1)).all()


def test_empty_scene_builds_to_nothing():
    m = Model(np.zeros((0, 3)), np.zeros(0, np.uint32))
    assert m.ntris == 0 and m.nnodes == 0


def test_coincident_and_degenerate_triangles():
    """identical triangles (centroids coincide) and zero-area triangles must not
    break the builder or exceed the depth bound"""
    base = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float64)
    P = np.concatenate([base] * 300 + [np.zeros((3, 3))] * 50 + [np.array([[0, 0, 0], [1, 1, 1], [2, 2, 2.0]])] * 10)
    idx = np.arange(P.shape[0], dtype=np.uint32)
    m = Model(P, idx)
    seen, maxd = walk(m.nodes(), m.tri32(), P[idx].reshape(-1, 9))
    # round 5: a triangle with two EQUAL vertices can never be reported by the reference (triangle_isect, bvh.c:754: its
    # determinant is exactly zero) and stays out of the traversal tree (lh_bvh.c tri_dead_class); a zero-area triangle with
    # three different vertices stays in
    assert sorted(seen) == list(range(300)) + list(range(350, 360)) and m.max_depth <= 60
    # ... their records are still there (unreferenced, behind the live ones): every primitive has one
    assert sorted(m.tri32()[:, 9].view(np.uint32).tolist()) == list(range(360))


def test_zero_area_triangles_outside_the_tree_do_not_change_a_record():
    """the example scene's cone has ten collapsed "quads" at its apex (v0 == v1); tessellated, each leaves hundreds of zero-area
    triangles along one segment.  With them outside the tree every record is still the oracle's, bit for bit -- also for rays
    aimed at those segments -- and the tree is smaller"""
    import os
    from tests.helpers import load_golden
    from lucille_amd import scenes
    g = load_golden("ao_c1")
    P, idx = scenes.tessellate(g["pos0"], g["idx0"], 3)          # geom 0 = the cone: 20 triangles -> 1280, half of them zero-area
    T = P[idx].reshape(-1, 3, 3)
    zero = ((T[:, 0] == T[:, 1]).all(1) | (T[:, 0] == T[:, 2]).all(1) | (T[:, 1] == T[:, 2]).all(1))
    assert zero.sum() == 640
    # v0 == v1 or v0 == v2: the reference's determinant is exactly 0 for any ray; v1 == v2: rounding noise, provably below its
    # 1e-14 for |dir| components <= 1024 only if (|ex| + |ey| + |ez|)^2 <= 10 / 1024 (lh_bvh.c tri_dead_class) -- at three
    # levels of subdivision only some of those edges are that short
    s1 = np.abs(T[:, 1] - T[:, 0]).sum(1)
    dead = (T[:, 0] == T[:, 1]).all(1) | (T[:, 0] == T[:, 2]).all(1) | ((T[:, 1] == T[:, 2]).all(1) & (s1 * s1 * (1 + 1e-9) <= 10.0 / 1024.0))
    assert 300 < dead.sum() <= 640
    m = Model(P, idx)
    seen, _ = walk(m.nodes(), m.tri32(), P[idx].reshape(-1, 9))
    assert sorted(seen) == np.nonzero(~dead)[0].tolist()
    rng = np.random.default_rng(7)
    n = 4000
    # rays from outside aimed at points ON the degenerate segments (apex -> base vertices), and random ones
    seg = T[zero][rng.integers(0, zero.sum(), n)]
    target = seg[:, 0] + (seg[:, 2] - seg[:, 0]) * rng.random((n, 1)) + (seg[:, 1] - seg[:, 0]) * rng.random((n, 1))
    org = target + rng.normal(size=(n, 3)) * 3.0
    dr = target - org
    dr[n // 2:] += rng.normal(size=(n - n // 2, 3)) * 0.05
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr)
    m.ref_build()            # the segments are edges of live triangles too: exact-t ties follow lucille's own tree (which keeps every triangle)
    try:
        got, _ = m.trace(org, dr)
    finally:
        Model.ref_off()
    assert np.array_equal(got[0], exp[0])
    for k in (1, 2, 3):
        assert np.array_equal(got[k], exp[k])
    assert (exp[0] != po.MISS).sum() > 100


def test_indexed_mesh_shared_vertices():
    P, idx = grid_mesh(17, 9)
    m = Model(P, idx)
    seen, _ = walk(m.nodes(), m.tri32(), P[idx].reshape(-1, 9))
    assert sorted(seen) == list(range(2 * 17 * 9))


def test_parallel_build_is_equivalent():
    """subtree tasks on the pthread pool must give a valid tree of the same size"""
    P, idx, org, dr = po.soup(150000, 3000, 0.005, 5)
    a = Model(P, idx, nthreads=1); b = Model(P, idx, nthreads=6)
    assert a.nnodes == b.nnodes and a.nleaves == b.nleaves
    ra, _ = a.trace(org, dr); rb, _ = b.trace(org, dr)
    for x, y in zip(ra, rb):
        assert np.array_equal(x, y)


def _wide_walk(words, refs, width, tri32, tri_dbl, grid_lo, grid_step):
    """every primitive reached exactly once through a wide 16-bit-grid tree; every decoded child box contains its triangles"""
    seen = []; visited = set(); maxd = 0
    stack = [(0, 1)]
    while stack:
        k, d = stack.pop()
        assert k not in visited
        visited.add(k); maxd = max(maxd, d)
        for c in range(width):
            ref = int(refs[k, c])
            w = words[k, 3 * c:3 * c + 3]
            lo = grid_lo + (w & 0xFFFF).astype(np.float64) * grid_step; hi = grid_lo + (w >> 16).astype(np.float64) * grid_step
            if ref == EMPTY:
                assert ((w & 0xFFFF) > (w >> 16)).any(), "empty slot must carry an inverted box"
                continue
            if ref >= 0:
                stack.append((ref, d + 1)); continue
            x = (~ref) & 0xFFFFFFFF
            first, cnt = x >> 2, (x & 3) + 1
            prims = tri32[first:first + cnt, 9].view(np.uint32)
            seen.extend(int(p) for p in prims)
            tv = tri_dbl[prims].reshape(-1, 3)
            assert (lo <= tv.min(0) + 1e-12).all() and (hi >= tv.max(0) - 1e-12).all()
    return seen, len(visited), maxd


@pytest.mark.parametrize("ntri,he", [(1, 0.2), (3, 0.2), (9, 0.1), (1000, 0.02), (30000, 0.005)])
def test_wide_collapses_cover_every_triangle_once(ntri, he):
    """the 4-wide (lh_q4node_t) and 8-wide (lh_q8node_t) collapses of the same binary tree: the slot choice minimises the summed
    area of the kept nodes (dp4_fill); whatever it chooses, every triangle hangs under exactly one leaf reference, empty slots are
    inverted, inner children are distinct records, and a decoded 16-bit box contains its triangles"""
    P, idx, _, _ = po.soup(ntri, 1, he, 99)
    m = Model(P, idx)
    tri32 = m.tri32(); tri_dbl = P[idx].reshape(-1, 9)
    lo, step = m.grid()
    q4 = np.ascontiguousarray(m.q4nodes()).view(np.uint32).reshape(-1, 16); n4, d4 = m.q4info()
    seen, nv, md = _wide_walk(q4[:, :12], q4[:, 12:16].view(np.int32), 4, tri32, tri_dbl, lo, step)
    # (a sibling group of two or more starts at an even index -- two 64-byte nodes to a 128-byte line; the skipped slots hold
    # empty records nobody refers to)
    refs4 = q4[:n4, 12:16].view(np.int32)
    pads = np.nonzero((refs4 == EMPTY).all(axis=1))[0] if n4 > 1 else np.zeros(0, np.int64)
    assert (q4[pads, :12] == 65535).all()
    assert sorted(seen) == list(range(ntri)) and nv + len(pads) == n4 and md == d4
    for k in range(n4):
        kids = refs4[k][refs4[k] >= 0]
        if len(kids) >= 2:
            assert kids.min() % 2 == 0 and sorted(kids) == list(range(kids.min(), kids.min() + len(kids)))      # adjacent, starting on a line
    q8 = m.q8nodes(); n8, d8 = m.q8info()
    seen, nv, md = _wide_walk(q8[:, :24], q8[:, 24:32].view(np.int32), 8, tri32, tri_dbl, lo, step)
    assert sorted(seen) == list(range(ntri)) and nv == n8 and md == d8
    assert n8 <= n4 and d8 <= d4


@pytest.mark.parametrize("layout,pairs", [("dfs", "0"), ("level", "0"), ("level", "1")])
def test_node_orders_hold_the_same_tree(layout, pairs, monkeypatch):
    """LH_Q4_LAYOUT / LH_Q4_PAIRS only renumber the 4-wide nodes (depth-first as in rounds 1-2, level order, level order with
    sibling groups on 128-byte lines): the same set of (boxes, leaf references) records, every triangle under one leaf"""
    P, idx, _, _ = po.soup(3000, 1, 0.02, 123)
    def records(layout_, pairs_):
        monkeypatch.setenv("LH_Q4_LAYOUT", layout_); monkeypatch.setenv("LH_Q4_PAIRS", pairs_)
        m = Model(P, idx)
        q4 = np.ascontiguousarray(m.q4nodes()).view(np.uint32).reshape(-1, 16); n4, d4 = m.q4info()
        tri32 = m.tri32(); lo, step = m.grid()
        seen, nv, md = _wide_walk(q4[:, :12], q4[:, 12:16].view(np.int32), 4, tri32, P[idx].reshape(-1, 9), lo, step)
        refs = q4[:n4, 12:16].view(np.int32)
        real = ~(refs == EMPTY).all(axis=1)
        # a record without its child numbering: boxes + leaf refs (inner refs -> a marker)
        canon = np.concatenate([q4[:n4, :12], np.where(refs >= 0, 1, refs).view(np.uint32)], axis=1)[real]
        return sorted(map(bytes, canon)), sorted(seen), nv, md
    base = records("dfs", "0")
    got = records(layout, pairs)
    assert got[0] == base[0] and got[1] == base[1] == list(range(3000)) and got[2] == base[2] and got[3] == base[3]
