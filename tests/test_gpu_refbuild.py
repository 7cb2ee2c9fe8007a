"""lucille's own tree (ri_bvh_build, bvh.c:276-379) built on the device, level by level (lh_refbuild.hip), against the
host restatement (lh_refbvh.c, itself pinned on the compiled reference in tests/test_refbvh.py): the same tree node for
node -- split axes, child boxes bit for bit, leaves with the same primitives in the same order.  Node numbering differs
(breadth-first here, depth-first there) and nothing depends on it."""
import os

import numpy as np
import pytest

import lucille_amd as la
from lucille_amd import scenes
from oracle import pyoracle as po
from tests.helpers import assert_hits_equal, load_golden

pytestmark = pytest.mark.gpu


def trees(P, idx):
    host = la.HipAccel(0); host.add_mesh(P, idx); host.commit()
    dev = la.HipAccel(0); dev.add_mesh(P, idx); dev.commit(on_device=True)
    th = host.ref_tree(); td = dev.ref_tree()
    return host, dev, th, td


def assert_same_tree(th, td, label):
    (hn, hp), (dn, dp) = th, td
    assert hn.shape[0] == dn.shape[0], "%s: %d nodes on the host, %d on the device" % (label, hn.shape[0], dn.shape[0])
    assert np.array_equal(hp, dp), "%s: leaf order differs" % label
    stack = [(0, 0)]; seen = 0
    assert hn[0]["parent"] == -1 and dn[0]["parent"] == -1
    while stack:
        h, d = stack.pop(); seen += 1
        a, b = hn[h], dn[d]
        assert a["is_leaf"] == b["is_leaf"] and a["depth"] == b["depth"], "%s: node kind / depth at host node %d" % (label, h)
        if a["is_leaf"]:
            assert a["first"] == b["first"] and a["count"] == b["count"], "%s: leaf range at host node %d" % (label, h)
            continue
        assert a["axis"] == b["axis"], "%s: split axis at host node %d" % (label, h)
        assert a["box"].tobytes() == b["box"].tobytes(), "%s: child boxes at host node %d" % (label, h)
        for k in range(2):
            ch, cd = int(a["child"][k]), int(b["child"][k])
            assert hn[ch]["parent"] == h and dn[cd]["parent"] == d
            stack.append((ch, cd))
    assert seen == hn.shape[0]


@pytest.mark.parametrize("ntri,he,seed", [(1, 0.2, 1), (16, 0.2, 2), (17, 0.2, 3), (33, 0.1, 4), (500, 0.05, 5), (3000, 0.05, 6), (200000, 0.008, 7)])
def test_device_built_reference_tree_equals_the_host_one_on_soups(ntri, he, seed):
    P, idx, org, dr = po.soup(ntri, 1000, he, 3000 + seed)
    host, dev, th, td = trees(P, idx)
    assert_same_tree(th, td, "soup %d" % ntri)
    assert dev.info()["ref_build_seconds"] > 0.0
    host.close(); dev.close()


def test_device_built_reference_tree_on_a_tessellated_scene_and_its_tie_winners():
    g = load_golden("ao_c1")
    meshes = [scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 4) for k in range(int(g["ngeoms"]))]
    host = la.HipAccel(0); dev = la.HipAccel(0)
    for P, I in meshes:
        host.add_mesh(P, I); dev.add_mesh(P, I)
    host.commit(); dev.commit(on_device=True)
    assert_same_tree(host.ref_tree(), dev.ref_tree(), "tessellated example scene")
    # rays through shared vertices and edge midpoints: the winners among equal-t hits come from this tree
    P, I = meshes[0]
    rng = np.random.default_rng(9)
    T = P[I.astype(np.int64)].reshape(-1, 3, 3)
    pick = rng.integers(0, T.shape[0], 20000)
    tgt = T[pick, rng.integers(0, 3, 20000)].copy()
    tgt[::2] = 0.5 * (T[pick[::2], 0] + T[pick[::2], 1])
    org = tgt + rng.normal(size=tgt.shape) * 3.0
    dr = tgt - org
    ok = np.abs(dr[:, 1]) > 1e-14
    org, dr = np.ascontiguousarray(org[ok]), np.ascontiguousarray(dr[ok])
    assert_hits_equal(dev.intersect_host(org, dr), host.intersect_host(org, dr), "ties on the device-built reference tree")
    host.close(); dev.close()


def test_device_built_reference_tree_on_degenerate_distributions():
    """equal boxes (every element on one side of any cut: the reference halves the list, bvh.c:1480-1486 -- after a partition
    that wrote the rights backwards), a flat scene (zero extent along one axis: no bins there), an exponentially spaced line"""
    rng = np.random.default_rng(21)
    tri = rng.uniform(-1.0, 1.0, (3, 3))
    same = np.ascontiguousarray(np.tile(tri, (700, 1)))
    flat = rng.uniform(-1.0, 1.0, (900, 3, 3)); flat[:, :, 1] = 0.25
    line = rng.uniform(-1e-3, 1e-3, (600, 3, 3)) + (1.1 ** np.arange(600) * 1e-6)[:, None, None] * np.array([1.0, 0.5, 0.25])
    mixed = np.concatenate([np.tile(tri, (300, 1)).reshape(-1, 3, 3), rng.uniform(-3.0, 3.0, (300, 3, 3)) * 0.01 + 2.0])
    for label, V in (("equal boxes", same.reshape(-1, 3, 3)), ("flat", flat), ("exponential line", line), ("half equal", mixed)):
        P = np.ascontiguousarray(V.reshape(-1, 3)); idx = np.arange(P.shape[0], dtype=np.uint32)
        host, dev, th, td = trees(P, idx)
        assert_same_tree(th, td, label)
        host.close(); dev.close()


def test_the_host_thread_still_builds_it_when_asked(monkeypatch):
    P, idx, org, dr = po.soup(5000, 1000, 0.05, 77)
    monkeypatch.setenv("LH_REF_BUILD", "host")
    dev = la.HipAccel(0); dev.add_mesh(P, idx); dev.commit(on_device=True)
    monkeypatch.delenv("LH_REF_BUILD")
    host = la.HipAccel(0); host.add_mesh(P, idx); host.commit()
    assert_same_tree(host.ref_tree(), dev.ref_tree(), "LH_REF_BUILD=host")       # ref_tree() waits for the thread
    host.close(); dev.close()
