"""The traversal tree built on the device (lh_build.hip: Morton LBVH -> the same 4-wide 16-bit-grid nodes; lucille's own
tree by a background host thread) against the oracle and against the host-built tree: hit records do not depend on the tree
(SURVEY 8a-10), so everything must stay bit for bit -- ids, fp64 t / u / v, exact-t ties, AO frames."""
import numpy as np
import pytest

import lucille_amd as la
from lucille_amd import render, scenes
from oracle import pyoracle as po
from tests.helpers import assert_hits_equal, load_golden

pytestmark = pytest.mark.gpu


def dev_accel(P, idx):
    acc = la.HipAccel(0); acc.add_mesh(P, idx)
    info = acc.commit(on_device=True)
    return acc, info


@pytest.mark.parametrize("ntri,he,seed", [(1, 0.2, 1), (3, 0.2, 2), (4, 0.2, 3), (5, 0.2, 4), (37, 0.1, 5), (3000, 0.05, 6), (200000, 0.008, 7)])
def test_device_built_tree_parity(ntri, he, seed):
    import torch
    P, idx, org, dr = po.soup(ntri, 60000, he, 1000 + seed)
    acc, info = dev_accel(P, idx)
    assert info["ntriangles"] == ntri and info["nnodes_traversal"] >= 1
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=8)
    # before the background reference tree is attached (or after: timing decides) and after waiting for it
    got0 = acc.intersect_host(org, dr)
    assert_hits_equal(got0, exp, "device-built %d (no wait)" % ntri)
    acc.wait_exact()
    assert_hits_equal(acc.intersect_host(org, dr), exp, "device-built %d" % ntri)
    assert np.array_equal(acc.intersect_host(org, dr, mode=la.MODE_ANY).astype(bool), exp[0] != po.MISS)
    d_o = torch.from_numpy(org).cuda(); d_d = torch.from_numpy(dr).cuda()
    # the 2-wide fp32 nodes exist only in the host builder: the textbook variant runs as the default walk on this scene
    out0 = acc.intersect_device(d_o, d_d, variant=la.VARIANT_DIRECT); torch.cuda.synchronize()
    assert_hits_equal(tuple(x.cpu().numpy() for x in out0), exp, "device-built %d, variant 0 -> default walk" % ntri)
    acc.close()


def test_device_built_tree_exact_t_ties_and_vertex_rays():
    """rays through shared edges / vertices of a tessellated plane: after wait_exact the winners follow lucille's own tree
    (built in the background), exactly as with the host builder"""
    g = load_golden("ao_c1")
    P, I = scenes.tessellate(g["pos0"], g["idx0"], 3)
    acc, _ = dev_accel(P, I)
    acc.wait_exact()
    o = po.Oracle(); o.add_mesh(P, I); o.build()
    rng = np.random.default_rng(4)
    T = P[I.astype(np.int64)].reshape(-1, 3, 3)
    pick = rng.integers(0, T.shape[0], 20000)
    tgt = T[pick, rng.integers(0, 3, 20000)].copy()                         # exactly a vertex
    tgt[::2] = 0.5 * (T[pick[::2], 0] + T[pick[::2], 1])                    # exactly an edge midpoint
    org = tgt + rng.normal(size=tgt.shape) * 3.0
    dr = tgt - org
    ok = np.abs(dr[:, 1]) > 1e-14
    org, dr = np.ascontiguousarray(org[ok]), np.ascontiguousarray(dr[ok])
    exp = o.intersect(org, dr, nthreads=8)
    assert_hits_equal(acc.intersect_host(org, dr), exp, "device-built, ties")
    assert o.count_equal_t(org[:4000], dr[:4000], exp[1][:4000]).max() >= 2       # ties really occur
    acc.close()


def test_device_built_scene_renders_the_same_frames_and_builds_fast():
    import torch
    g = load_golden("ao_c1")
    host = la.HipAccel(0); dev = la.HipAccel(0)
    for k in range(int(g["ngeoms"])):
        Pk, Ik = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 5)
        host.add_mesh(Pk, Ik); dev.add_mesh(Pk, Ik)
    ih = host.commit(); idv = dev.commit(on_device=True)
    c = g["camera"]; cam = la.Camera.make(320, 240, c[16], c[:16], int(c[19]))
    a, sa = render.render_ao_frame(host, cam, 2, 16, tile=320, seed=3)
    b, sb = render.render_ao_frame(dev, cam, 2, 16, tile=320, seed=3)
    torch.cuda.synchronize()
    assert sa == sb and torch.equal(a, b)
    assert idv["ntriangles"] == ih["ntriangles"] == 322 * 4 ** 5
    assert idv["build_seconds"] < ih["build_seconds"]
    # beams need lucille's own tree: the call waits for the background build
    bg = load_golden("beams_300")
    host.close(); dev.close()


def test_degenerate_distributions():
    """equal centroids (ties in the Morton order are broken by position) and an exponentially spaced line of triangles
    (a radix tree as deep as the code is long: the commit falls back to the host builder when the kernel's stack bound
    would not hold)"""
    rng = np.random.default_rng(9)
    n = 3000
    c = np.zeros((n, 1, 3)) + 0.5
    tri = c + rng.uniform(-0.3, 0.3, (n, 3, 3)); tri -= tri.mean(axis=1, keepdims=True) - 0.5      # every centroid = (0.5, 0.5, 0.5)
    P = tri.reshape(-1, 3); idx = np.arange(3 * n, dtype=np.uint32)
    org = rng.uniform(-1, 2, (20000, 3)); dr = rng.uniform(0, 1, (20000, 3)) - org
    acc, _ = dev_accel(P, idx); acc.wait_exact()
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    assert_hits_equal(acc.intersect_host(org, dr), o.intersect(org, dr, nthreads=8), "equal centroids")
    acc.close()
    m = 60
    x = 2.0 ** -np.arange(m)
    tri = np.stack([np.stack([x, np.zeros(m), np.zeros(m)], 1), np.stack([x * 1.0001, np.full(m, 1e-3), np.zeros(m)], 1),
                    np.stack([x, np.zeros(m), np.full(m, 1e-3)], 1)], 1)
    tri = np.concatenate([tri, tri + np.array([0, 2e-3, 0]), tri + np.array([0, 4e-3, 0]), tri + np.array([0, 6e-3, 0]), tri + np.array([0, 8e-3, 0])])
    P = tri.reshape(-1, 3); idx = np.arange(P.shape[0], dtype=np.uint32)
    org = rng.uniform(-0.5, 1.5, (20000, 3)); tgt = P[rng.integers(0, P.shape[0], 20000)] + rng.normal(scale=1e-4, size=(20000, 3))
    dr = tgt - org
    acc, info = dev_accel(P, idx); acc.wait_exact()
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    assert_hits_equal(acc.intersect_host(org, dr), o.intersect(org, dr, nthreads=8), "exponential line")
    acc.close()


def test_stack_overflow_paths_keep_every_bit():
    """a device-built tree can be deeper than the 64-row LDS stack holds in the worst case (BASELINE config 5's scene: 4-wide
    depth 22 -> 71 rows).  Rays that would overflow are finished by k_overflow_fix (ray dumps, materialised AO) or by the
    reference walk (fused AO).  Reached here by capping the stack at 8 rows ("stack_cap"): hit records and AO frames stay
    bit equal to the uncapped walk's, fused or not, on a device-built and on a host-built tree"""
    import torch
    g = load_golden("ao_c1")
    meshes = [scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 4) for k in range(int(g["ngeoms"]))]
    allp = np.concatenate([m[0] for m in meshes]); lo, hi = allp.min(0), allp.max(0)
    c = g["camera"]; cam = la.Camera.make(160, 120, c[16], c[:16], int(c[19]))
    rng = np.random.default_rng(4)
    org = rng.uniform(lo - 1, hi + 1, (200000, 3)); dr = rng.uniform(lo, hi, (200000, 3)) - org
    do = torch.from_numpy(org).cuda(); dd = torch.from_numpy(dr).cuda()
    for on_device in (True, False):
        acc = la.HipAccel(0)
        for P, I in meshes:
            acc.add_mesh(P, I)
        info = acc.commit(on_device=on_device); acc.wait_exact()
        assert info["max_depth"] >= 6
        ref_img, ref_stats = render.render_ao_frame(acc, cam, 2, 16, tile=160, seed=5)
        ref_hits = acc.intersect_host(org, dr); ref_occ = acc.intersect_host(org, dr, la.MODE_ANY)
        base = acc.intersect_device(do, dd, counters=True)[-1]["retraced"]
        acc.set_param("stack_cap", 8)
        over = acc.intersect_device(do, dd, counters=True)[-1]["retraced"]
        assert over >= base + 20, (base, over)                       # the capped walk really overflows
        assert_hits_equal(acc.intersect_host(org, dr), ref_hits, "capped stack, closest")
        assert np.array_equal(acc.intersect_host(org, dr, la.MODE_ANY), ref_occ)
        img, stats = render.render_ao_frame(acc, cam, 2, 16, tile=160, seed=5)           # fused: overflow -> reference walk
        acc.set_param("ao_fused", 0)
        img2, stats2 = render.render_ao_frame(acc, cam, 2, 16, tile=160, seed=5)         # materialised: k_overflow_fix
        torch.cuda.synchronize()
        assert stats == stats2 == ref_stats and torch.equal(img, ref_img) and torch.equal(img2, ref_img)
        acc.close()


def test_device_commit_does_not_leak_device_memory():
    """lh_device_build's temporaries were never freed (dfree called itself).  A scene re-committed every
    frame (scene.c:84-98) must leave the device's free memory where it was."""
    import torch
    P, idx, org, dr = po.soup(300000, 1000, 0.006, 77)
    def once():
        acc, _ = dev_accel(P, idx)
        acc.intersect_host(org, dr)
        acc.close()
    once(); once()                                    # allocator pools, code objects
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(6):
        once()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    # one commit's temporaries are ~120 B / triangle = 36 MB here: six leaked commits would be > 200 MB
    assert free0 - free1 < 16 << 20, "device memory shrank by %.1f MB over six device commits" % ((free0 - free1) / 1e6)


def test_long_skewed_distribution_through_the_sah_top():
    """150 000 small triangles whose spacing grows geometrically along a line, plus a uniform cloud: the SAH over the radix
    tree's subtree roots peels a few items per level on such data (its recursion is bounded and falls back to halving), the
    collapsed tree may need the checked walk or the host builder -- hit records stay the oracle's"""
    rng = np.random.default_rng(21)
    n = 150000
    x = np.cumsum(1e-6 * 1.00008 ** np.arange(n))
    c = np.stack([x, 1e-3 * rng.standard_normal(n), 1e-3 * rng.standard_normal(n)], 1)
    c[::7] = rng.uniform(0, x[-1], (len(c[::7]), 3)) * np.array([1, 0.01, 0.01])
    tri = c[:, None, :] + rng.uniform(-1, 1, (n, 3, 3)) * (2e-6 + 1e-4 * x[:, None, None] / x[-1])
    P = tri.reshape(-1, 3); idx = np.arange(3 * n, dtype=np.uint32)
    tgt = tri[rng.integers(0, n, 30000)].mean(axis=1)                            # a point inside a triangle
    org = tgt + rng.normal(size=(30000, 3)) * np.array([0.05 * x[-1], 0.02, 0.02])
    dr = tgt - org
    ok = np.abs(dr[:, 1]) > 1e-14
    org, dr = np.ascontiguousarray(org[ok]), np.ascontiguousarray(dr[ok])
    acc, info = dev_accel(P, idx); acc.wait_exact()
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=8)
    assert (exp[0] != po.MISS).mean() > 0.5
    assert_hits_equal(acc.intersect_host(org, dr), exp, "skewed line, device-built")
    acc.close()


@pytest.mark.parametrize("ntri,he,seed", [(1, 0.2, 1), (2, 0.2, 8), (9, 0.2, 2), (37, 0.1, 5), (3000, 0.05, 6), (200000, 0.008, 7)])
def test_device_built_eight_wide_nodes(ntri, he, seed):
    """set_param("wide8", 1) before a device commit: the builder also collapses its binary tree to the 8-wide one-cache-line
    nodes (k_collapse8_level) and ray dumps walk those -- same records, bit for bit, and the dump record size says which
    nodes were walked"""
    P, idx, org, dr = po.soup(ntri, 60000, he, 2000 + seed)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.set_param("wide8", 1)
    acc.commit(on_device=True)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=8)
    assert_hits_equal(acc.intersect_host(org, dr), exp, "device-built 8-wide %d (no wait)" % ntri)
    acc.wait_exact()
    assert_hits_equal(acc.intersect_host(org, dr), exp, "device-built 8-wide %d" % ntri)
    assert np.array_equal(acc.intersect_host(org, dr, mode=la.MODE_ANY).astype(bool), exp[0] != po.MISS)
    if ntri > 16:
        assert acc.dump_node_bytes() == 128
    # the same scene without the knob stays on the 4-wide nodes
    ref = la.HipAccel(0); ref.add_mesh(P, idx); ref.set_param("wide8", 0); ref.commit(on_device=True)
    assert ref.dump_node_bytes() == 64
    assert_hits_equal(ref.intersect_host(org, dr), exp, "device-built 4-wide %d" % ntri)
    # asked for only after the commit: the build is repeated for its 8-wide collapse when the first dump needs it
    before = ref.info()["device_bytes"]
    ref.set_param("wide8", 1)
    assert_hits_equal(ref.intersect_host(org, dr), exp, "device-built, 8-wide nodes on first use %d" % ntri)
    if ntri > 16:
        assert ref.dump_node_bytes() == 128 and ref.info()["device_bytes"] > before
    acc.close(); ref.close()


def test_device_built_eight_wide_nodes_can_outnumber_the_four_wide_ones():
    """eight far-apart pairs of triangles: the 4-wide collapse needs 5 nodes (root -> 4 x [2 pairs]), the 8-wide one 9
    (root -> 8 pairs), so the 8-wide array cannot be sized by the 4-wide count"""
    rng = np.random.default_rng(11)
    tris = []
    for cx in (0.0, 100.0):
        for cy in (0.0, 100.0):
            for cz in (0.0, 100.0):
                for off in (0.0, 6.0):
                    c = np.array([cx + off, cy + off, cz + off])
                    tris.append(c + rng.uniform(-1.0, 1.0, (3, 3)))
    P = np.ascontiguousarray(np.array(tris).reshape(-1, 3)); idx = np.arange(P.shape[0], dtype=np.uint32)
    T = P.reshape(-1, 3, 3)
    pick = rng.integers(0, T.shape[0], 20000)
    w = rng.dirichlet((1.0, 1.0, 1.0), 20000)
    tgt = (T[pick] * w[:, :, None]).sum(axis=1)
    org = tgt + rng.normal(size=tgt.shape) * 150.0
    dr = np.ascontiguousarray(tgt - org); org = np.ascontiguousarray(org)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=8)
    assert (exp[0] != po.MISS).mean() > 0.9
    for wide8 in (1, 0):
        acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.set_param("wide8", wide8)
        info = acc.commit(on_device=True)
        assert acc.dump_node_bytes() == (128 if wide8 else 64)
        assert_hits_equal(acc.intersect_host(org, dr), exp, "pairs, wide8=%d" % wide8)
        if not wide8:
            assert info["nnodes_traversal"] == 6       # root, one empty record (the sibling group starts on a 128-byte line), 4 nodes
        acc.close()


def test_commit_chooses_its_builders_by_the_size_of_the_scene(monkeypatch):
    """build_threads == 0: the device builders from LH_AUTO_DEVICE_TRIANGLES (1 M; lowered here through the environment)
    triangles on, the host builders below; LH_BUILD_ON_HOST / a thread count / LH_BUILD say otherwise; records never change"""
    P, idx, org, dr = po.soup(5000, 20000, 0.05, 4242)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=8)

    def built_on_device(**kw):
        acc = la.HipAccel(0); acc.add_mesh(P, idx); info = acc.commit(**kw)
        assert_hits_equal(acc.intersect_host(org, dr), exp, "auto policy %r" % (kw,))
        acc.close()
        return info["nnodes"] == info["nnodes_traversal"]

    assert not built_on_device()                                   # 5 000 triangles: host
    monkeypatch.setenv("LH_AUTO_DEVICE_TRIANGLES", "5000")
    assert built_on_device()                                       # at the threshold: device
    assert not built_on_device(build="host") and not built_on_device(build_threads=4)
    monkeypatch.setenv("LH_AUTO_DEVICE_TRIANGLES", "5001")
    assert not built_on_device() and built_on_device(build="device")
    monkeypatch.setenv("LH_BUILD", "device")
    assert built_on_device() and built_on_device(build="host")     # the environment has the last word
    monkeypatch.setenv("LH_BUILD", "host")
    assert not built_on_device(build="device")
