"""Zero-area triangles with two EQUAL vertices stay out of the traversal tree (round 5; lh_bvh.c tri_dead_class, lh_build.hip
k_prim_boxes): the reference's determinant test (triangle_isect, bvh.c:754) rejects them for every ray -- exactly for v0 == v1
and v0 == v2, provably up to direction components of 1024 for v1 == v2 with short edges, and beyond that the reference's own
walk on its own tree decides (lh_walk.h ray_needs_ref_walk).  Hit records must not change by a bit, on either builder."""
import os

import numpy as np
import pytest

import lucille_amd as la
from lucille_amd import scenes
from oracle import pyoracle as po
from tests.helpers import assert_hits_equal, load_golden

pytestmark = pytest.mark.gpu


def cone(levels):
    g = load_golden("ao_c1")
    P, I = scenes.tessellate(g["pos0"], g["idx0"], levels)        # the example scene's cone: ten collapsed apex "quads" (v0 == v1)
    T = P[I.astype(np.int64)].reshape(-1, 3, 3)
    s1 = np.abs(T[:, 1] - T[:, 0]).sum(1)
    zero = (T[:, 0] == T[:, 1]).all(1) | (T[:, 0] == T[:, 2]).all(1) | (T[:, 1] == T[:, 2]).all(1)
    dead = (T[:, 0] == T[:, 1]).all(1) | (T[:, 0] == T[:, 2]).all(1) | ((T[:, 1] == T[:, 2]).all(1) & (s1 * s1 * (1 + 1e-9) <= 10.0 / 1024.0))
    return P, I, T, zero, dead


def rays_at_segments(T, zero, n, seed):
    rng = np.random.default_rng(seed)
    seg = T[zero][rng.integers(0, zero.sum(), n)]
    target = seg[:, 0] + (seg[:, 2] - seg[:, 0]) * rng.random((n, 1)) + (seg[:, 1] - seg[:, 0]) * rng.random((n, 1))
    org = target + rng.normal(size=(n, 3)) * 3.0
    dr = target - org
    dr[n // 2:] += rng.normal(size=(n - n // 2, 3)) * 0.02          # half exactly at a segment (a shared edge: exact-t ties), half near it
    ok = np.abs(dr[:, 1]) > 1e-14
    return np.ascontiguousarray(org[ok]), np.ascontiguousarray(dr[ok])


@pytest.mark.parametrize("build", ["host", "device"])
def test_dropped_triangles_change_no_record(build):
    import torch
    P, I, T, zero, dead = cone(6)
    assert zero.sum() == 10 * 4 ** 6 and dead.sum() == zero.sum()          # at this level every v1 == v2 edge is short enough
    acc = la.HipAccel(0); acc.add_mesh(P, I)
    info = acc.commit(build=build)
    acc.wait_exact()
    assert info["ntriangles"] == T.shape[0] and acc.info()["ntriangles_in_tree"] == T.shape[0] - int(dead.sum())
    o = po.Oracle(); o.add_mesh(P, I); o.build()
    org, dr = rays_at_segments(T, zero, 60000, 11)
    exp = o.intersect(org, dr, nthreads=8)
    assert (exp[0] != po.MISS).sum() > 10000
    assert_hits_equal(acc.intersect_host(org, dr), exp, "%s tree, rays at the degenerate segments" % build)
    assert np.array_equal(acc.intersect_host(org, dr, mode=la.MODE_ANY).astype(bool), exp[0] != po.MISS)
    # the batch paths on the device: the default walk, the textbook walk, a handful of rays (k_trace_small)
    d_o = torch.from_numpy(org).cuda(); d_d = torch.from_numpy(dr).cuda()
    for variant in (la.VARIANT_DEFAULT, la.VARIANT_DIRECT):
        out = acc.intersect_device(d_o, d_d, variant=variant); torch.cuda.synchronize()
        assert_hits_equal(tuple(x.cpu().numpy() for x in out), exp, "%s tree, variant %d" % (build, variant))
    assert_hits_equal(acc.intersect_host(org[:40], dr[:40]), tuple(x[:40] for x in exp), "%s tree, small batch" % build)
    # unnormalised directions beyond 1024 per component: the traversal tree alone cannot vouch for the v1 == v2 triangles there,
    # the reference's own walk decides -- the records are the reference's (t scales with 1 / |dir|)
    big = dr * 4096.0
    assert (np.abs(big).max(1) > 1024.0).mean() > 0.9
    expb = o.intersect(org, big, nthreads=8)
    assert_hits_equal(acc.intersect_host(org, big), expb, "%s tree, direction components beyond 1024" % build)
    assert np.array_equal(acc.intersect_host(org, big, mode=la.MODE_ANY).astype(bool), expb[0] != po.MISS)
    outb = acc.intersect_device(torch.from_numpy(org).cuda(), torch.from_numpy(big).cuda(), variant=la.VARIANT_DIRECT); torch.cuda.synchronize()
    assert_hits_equal(tuple(x.cpu().numpy() for x in outb), expb, "%s tree, textbook walk, big directions" % build)
    assert_hits_equal(acc.intersect_host(org[:33], big[:33]), tuple(x[:33] for x in expb), "%s tree, small batch, big directions" % build)
    acc.close()


def test_switch_keeps_every_triangle_in_the_tree_and_the_frame():
    """LH_DROP_DEGENERATE=0 builds the tree of rounds 1-4; the AO frame of the tessellated example scene is the same frame"""
    import torch
    from lucille_amd import render
    g = load_golden("ao_c1")
    c = g["camera"]; cam = la.Camera.make(192, 192, c[16], c[:16], int(c[19]))
    frames = []; in_tree = []
    for sw in ("1", "0"):
        os.environ["LH_DROP_DEGENERATE"] = sw
        try:
            acc = la.HipAccel(0)
            for k in range(int(g["ngeoms"])):
                P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 4); acc.add_mesh(P, I)
            acc.commit(build="device" if sw == "1" else "host")
            acc.wait_exact()
            in_tree.append(acc.info()["ntriangles_in_tree"])
            img, st = render.render_ao_frame(acc, cam, 1, 16, tile=192, seed=5)
            frames.append((img.cpu(), st)); acc.close()
        finally:
            del os.environ["LH_DROP_DEGENERATE"]
    assert in_tree[1] == 322 * 256 and in_tree[0] < in_tree[1]
    assert torch.equal(frames[0][0], frames[1][0]) and frames[0][1] == frames[1][1]


def big_zero_area_scene(seed=3):
    """60 triangles of ~15 units, every third with v1 == v2, every fifth with three different points on one line: zero-area triangles
    that STAY in the tree (too large for class 2; not two equal vertices).  The reference's determinant for them is rounding noise
    that clears 1e-14 once (largest direction component) x |e1|_1 |e2|_1 is large -- for any ray that reaches their leaf in ITS
    tree, near them or not (found by tools/fuzz_parity.py, seed 11 round 8): rays beyond 1 / s2 go through the reference's own walk"""
    rng = np.random.default_rng(seed)
    c = rng.uniform(0, 1, (60, 1, 3)); T = (c + rng.normal(size=(60, 3, 3)) * 0.3) * 28.0
    T[::3, 2] = T[::3, 1]
    T[1::5, 2] = T[1::5, 0] + 2.0 * (T[1::5, 1] - T[1::5, 0])
    P = T.reshape(-1, 3).copy(); idx = np.arange(180, dtype=np.uint32)
    n = 60000; pick = rng.integers(0, 60, n)
    w = rng.random((n, 3)); w /= w.sum(1, keepdims=True); tgt = (T[pick] * w[:, :, None]).sum(1)
    org = tgt + rng.normal(size=(n, 3)) * 28.0 * 30.0
    dr = (tgt - org) * rng.uniform(0.001, 1000.0, (n, 1))
    dr[-n // 4:] /= np.abs(dr[-n // 4:]).max(1, keepdims=True) * rng.uniform(1.0, 400.0, (n // 4, 1))      # small directions too: below 1 / s2
    ok = np.abs(dr[:, 1]) > 1e-14 * np.abs(dr).max(1)
    return P, idx, np.ascontiguousarray(org[ok]), np.ascontiguousarray(dr[ok])


@pytest.mark.parametrize("build", ["host", "device"])
@pytest.mark.parametrize("name", ["fuzz_r06_f662_99", "fuzz_r06_f661_359"])
def test_near_collinear_triangles_of_round_6_fuzz(name, build):
    """tests/test_hostwalk.py's two scenes on the device: collinear triangles of max |n_k| = 3e-15 |e1|_1 |e2|_1 (round 5's rule stopped at
    8.9e-16), which the reference reports at a t before their box -- 3 of 60 000 rays differed on the host-built tree of the first scene,
    1 of 60 000 on the device-built tree of the second (tools/fuzz_parity.py seeds 662 / 661)"""
    import torch
    z = load_golden(name); P, idx, org, dr = z["P"], z["idx"], np.ascontiguousarray(z["org"]), np.ascontiguousarray(z["dr"])
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=8)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build=build); acc.wait_exact()
    assert_hits_equal(acc.intersect_host(org, dr), exp, "%s, %s tree" % (name, build))
    assert np.array_equal(acc.intersect_host(org, dr, mode=la.MODE_ANY).astype(bool), exp[0] != po.MISS)
    d_o = torch.from_numpy(org).cuda(); d_d = torch.from_numpy(dr).cuda()
    out = acc.intersect_device(d_o, d_d); torch.cuda.synchronize()
    assert_hits_equal(tuple(x.cpu().numpy() for x in out), exp, "%s, %s tree, device batch" % (name, build))
    acc.close()


@pytest.mark.parametrize("build", ["host", "device"])
def test_zero_area_triangles_that_stay_in_the_tree(build):
    import torch
    P, idx, org, dr = big_zero_area_scene()
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=8)
    T = P.reshape(-1, 3, 3)
    za = np.zeros(60, bool); za[::3] = True; za[1::5] = True
    hit_za = (exp[0] != po.MISS) & za[np.minimum(exp[0], 59)]
    assert hit_za.sum() > 20                          # the reference does report such triangles (at t = +-0, by its determinant's noise)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build=build); acc.wait_exact()
    assert acc.info()["ntriangles_in_tree"] == 60
    assert_hits_equal(acc.intersect_host(org, dr), exp, "%s tree" % build)
    assert np.array_equal(acc.intersect_host(org, dr, mode=la.MODE_ANY).astype(bool), exp[0] != po.MISS)
    d_o = torch.from_numpy(org).cuda(); d_d = torch.from_numpy(dr).cuda()
    out = acc.intersect_device(d_o, d_d); torch.cuda.synchronize()
    assert_hits_equal(tuple(x.cpu().numpy() for x in out), exp, "%s tree, device batch" % build)
    assert_hits_equal(acc.intersect_host(org[:40], dr[:40]), tuple(x[:40] for x in exp), "%s tree, small batch" % build)
    acc.close()


def test_every_queued_any_hit_ray_gets_its_answer():
    """a fan of triangles about ONE vertex, far from the origin: half of an any-hit dump's rays leave the persistent walk (stack rows, or
    a first hit the reference may not reach) and the reference's own walk on such a scene runs for a long time in one lane.  The sweep
    behind the launch used to give up after half a second without progress -- the answer slots of the rays it left were never
    written (tools/fuzz_parity.py big, seed 41 round 7: 110 000 of 400 000).  The output is filled with a sentinel first."""
    import torch
    rng = np.random.default_rng(41)
    ntri, n = 60000, 100000
    c = rng.uniform(0, 1, (ntri, 1, 3)); T = c + rng.normal(size=(ntri, 3, 3)) * 0.03; T[:, 0] = T[0, 0]
    T = T * 1.5 + np.array([-57.4, -105.4, 104.0])
    P = T.reshape(-1, 3).copy(); idx = np.arange(3 * ntri, dtype=np.uint32)
    pick = rng.integers(0, ntri, n); w = rng.random((n, 3)); w /= w.sum(1, keepdims=True); tgt = (T[pick] * w[:, :, None]).sum(1)
    tgt[:n // 4] = T[pick[:n // 4], rng.integers(0, 3, n // 4)]
    org = tgt + rng.normal(size=(n, 3)) * 1.5; dr = (tgt - org) * rng.uniform(0.001, 1000.0, (n, 1))
    ok = np.abs(dr[:, 1]) > 1e-14 * np.abs(dr).max(1); org, dr = np.ascontiguousarray(org[ok]), np.ascontiguousarray(dr[ok]); n = org.shape[0]
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=16)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build="host"); acc.wait_exact()
    d_o = torch.from_numpy(org).cuda(); d_d = torch.from_numpy(dr).cuda()
    # patience of one tick: the pass beside the launch leaves at once, the sweep behind it (which had the same time limit) takes the queue
    for patience in (0, 1):
        acc.set_param("coop_patience", patience)
        out = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
        acc.intersect_device(d_o, d_d, out=(out,), mode=la.MODE_ANY); torch.cuda.synchronize()
        g = out.cpu().numpy()
        assert int((g == 7).sum()) == 0, "%d any-hit answers were never written (patience %d)" % (int((g == 7).sum()), patience)
        assert np.array_equal(g.astype(bool), exp[0] != po.MISS)
    pr = torch.full((n,), 0x77777777, dtype=torch.int32, device="cuda"); tt = torch.zeros(n, dtype=torch.float64, device="cuda"); uu = torch.zeros_like(tt); vv = torch.zeros_like(tt)
    acc.intersect_device(d_o, d_d, out=(pr, tt, uu, vv)); torch.cuda.synchronize()
    assert_hits_equal((pr.cpu().numpy(), tt.cpu().numpy(), uu.cpu().numpy(), vv.cpu().numpy()), exp, "the fan, closest hit")
    acc.close()


@pytest.mark.parametrize("build", ["host", "device"])
def test_one_large_collinear_triangle_sends_every_ray_source_through_the_reference_walk(build):
    """a numerically collinear triangle with |e1|_1 |e2|_1 > 1 that stays in the tree pulls deg_dcap = 1 / s2
    below 1 -- below 0.577 EVERY unit-length ray has a component beyond it and belongs to the reference's own walk.  The ray dumps
    (rays from arrays) applied that rule, the fused AO stage (rays generated in the refill) skipped it "because AO rays are unit
    vectors": the tile pipeline and the dump path answered the same ray by different rules.  Now every source takes the test:
    the fused stage's rays take ONE node step each (the root, under a negative culling bound) and go to the reference walk,
    the frame equals the materialised stage's (dump path) bit for bit and the AO rays' occlusion equals the oracle's."""
    import torch
    g = load_golden("ao_c1")
    allp = np.concatenate([g["pos%d" % k][:, :3] for k in range(int(g["ngeoms"]))]); lo, hi = allp.min(0), allp.max(0)
    rng = np.random.default_rng(77)
    ntri = 600
    ctr = rng.uniform(0, 1, (ntri, 1, 3)); T = (ctr + rng.normal(size=(ntri, 3, 3)) * 0.08) * (hi - lo) + lo
    a = 0.5 * (lo + hi); d = np.array([0.6, 0.1, 0.79]); d /= np.linalg.norm(d)
    Z = np.stack([a - 4.0 * d, a + 5.0 * d, a - 4.0 * d + 1.75 * (9.0 * d)])[None]          # three different points on one line, 9 and 15.75 units
    P = np.concatenate([T, Z]).reshape(-1, 3).copy(); idx = np.arange(P.shape[0], dtype=np.uint32)
    c = g["camera"]; W, H, ns = 64, 48, 16
    cam = la.Camera.make(W, H, c[16], c[:16], int(c[19]))
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build=build); acc.wait_exact()
    assert acc.info()["ntriangles_in_tree"] == ntri + 1              # too large for class 2, no two equal vertices: it stays
    acc.set_param("ao_fused", 1)
    img_f, st_f = acc.render_ao_tile(cam, 0, 0, W, H, 1, ns, seed=9)
    acc.set_param("ao_fused", 0)
    img_m, st_m = acc.render_ao_tile(cam, 0, 0, W, H, 1, ns, seed=9)
    aorg = acc.scratch(8, np.float64, 3); adir = acc.scratch(9, np.float64, 3); occ = acc.scratch(10, np.uint8, 1)
    torch.cuda.synchronize()
    assert st_f == st_m and st_f["ao_rays"] > 5000 and torch.equal(img_f, img_m)
    ok = np.abs(adir[:, 1]) > 1e-14
    exp = o.intersect(aorg[ok], adir[ok], nthreads=8)
    assert np.array_equal(occ[ok].astype(bool).ravel(), exp[0] != po.MISS)
    # the fused stage by itself, counted, under round 5's rule (LH_DANGER_BOXES=0 at commit: EVERY ray beyond the cap, whatever its
    # source): one node step per ray (camera rays and AO rays alike), everything else by the reference walk -- and the same frame.
    # (Round 6's default asks the box of the sliver's leaf first: rays that miss it walk the traversal tree -- the frame above.)
    os.environ["LH_DANGER_BOXES"] = "0"
    try:
        acc5 = la.HipAccel(0); acc5.add_mesh(P, idx); acc5.commit(build=build); acc5.wait_exact()
    finally:
        del os.environ["LH_DANGER_BOXES"]
    acc5.set_param("ao_fused", 1)
    acc5.trace_statistics(True); acc5.statistics(clear=True)
    img_5, st_5 = acc5.render_ao_tile(cam, 0, 0, W, H, 1, ns, seed=9); torch.cuda.synchronize()
    cnt = acc5.statistics(clear=True); acc5.trace_statistics(False)
    assert st_5 == st_f and torch.equal(img_5, img_f)
    assert cnt["rays"] == st_f["primary_rays"] + st_f["ao_rays"]
    assert 0 < cnt["nodes"] <= cnt["rays"] and cnt["tris"] == 0, cnt
    acc.close(); acc5.close()


@pytest.mark.parametrize("build", ["host", "device"])
def test_one_sliver_no_longer_sends_the_whole_dump_through_the_reference_walk(build):
    """Round 6 (second half): the cap 1 / s2 of a zero-area triangle that stays in the tree concerns only rays that can
    reach it in the reference's walk -- rays that hit the box of its leaf in lucille's own tree (lh_commit.hip lh_danger_scan,
    lh_walk.h ray_needs_ref_walk).  A soup in a box of 2 000 units (a scene in millimetres) with ONE collinear triangle of 30 x 50
    units in a corner: s2 = 1 500 and more, so the cap is far below every unit direction -- round 5 sent all 60 000 rays through the
    single-lane reference walk; now only the rays that pass that corner go, and the records still equal the oracle's bit for bit."""
    import torch
    rng = np.random.default_rng(91)
    ntri, n = 40000, 60000
    c = rng.uniform(0, 1, (ntri, 1, 3)); T = (c + rng.normal(size=(ntri, 3, 3)) * 0.004) * 2000.0
    a = np.array([60.0, 80.0, 40.0]); u = np.array([0.6, 0.1, 0.79]); u /= np.linalg.norm(u)
    Z = np.stack([a, a + 30.0 * u, a + 1.6 * 30.0 * u])[None]                  # three different points on one line: it stays in the tree
    P = np.concatenate([T, Z]).reshape(-1, 3).copy(); idx = np.arange(P.shape[0], dtype=np.uint32)
    pick = rng.integers(0, ntri + 1, n); w = rng.random((n, 3)); w /= w.sum(1, keepdims=True)
    tri = P.reshape(-1, 3, 3); tgt = (tri[pick] * w[:, :, None]).sum(1)
    tgt[:2000] = Z[0, 0] + (Z[0, 2] - Z[0, 0]) * rng.random((2000, 1))      # 2 000 rays aimed AT the sliver's line (the reference's noise determinant decides)
    org = tgt + rng.normal(size=(n, 3)) * 300.0
    dr = tgt - org; dr /= np.linalg.norm(dr, axis=1, keepdims=True)            # unit directions: every one beyond the cap
    ok = np.abs(dr[:, 1]) > 1e-14
    org, dr = np.ascontiguousarray(org[ok]), np.ascontiguousarray(dr[ok]); n = org.shape[0]
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=8)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build=build); acc.wait_exact()
    assert acc.info()["ntriangles_in_tree"] == ntri + 1
    d_o = torch.from_numpy(org).cuda(); d_d = torch.from_numpy(dr).cuda()
    out, cnt = acc.intersect_device(d_o, d_d, counters=True); torch.cuda.synchronize()
    assert_hits_equal(tuple(x.cpu().numpy() for x in out), exp, "%s tree, device batch" % build)
    assert 1500 < cnt["retraced"] < n // 4, "rays through the reference walk: %d of %d" % (cnt["retraced"], n)
    occ = acc.intersect_device(d_o, d_d, mode=la.MODE_ANY)[0]; torch.cuda.synchronize()
    assert np.array_equal(occ.cpu().numpy().astype(bool), exp[0] != po.MISS)
    assert_hits_equal(acc.intersect_host(org[:3000], dr[:3000]), tuple(x[:3000] for x in exp), "%s tree, host batch" % build)
    for k in range(0, 400, 7):                                                  # ray by ray: the host walk (host-built trees) / the coalesced device path
        hit, p_, t_, u_, v_ = acc.intersect1(org[k], dr[k])
        assert p_ == int(exp[0][k]) and (p_ == po.MISS or (t_, u_, v_) == (exp[1][k], exp[2][k], exp[3][k])), k
    # round 5's rule (LH_DANGER_BOXES=0 at commit): every ray takes the reference walk -- and gives the same records
    os.environ["LH_DANGER_BOXES"] = "0"
    try:
        acc2 = la.HipAccel(0); acc2.add_mesh(P, idx); acc2.commit(build=build); acc2.wait_exact()
        out2, cnt2 = acc2.intersect_device(d_o, d_d, counters=True); torch.cuda.synchronize()
    finally:
        del os.environ["LH_DANGER_BOXES"]
    assert cnt2["retraced"] >= n - 10
    assert all(torch.equal(x, y) for x, y in zip(out, out2))
    acc.close(); acc2.close()
