"""Parity tests proper: the HIP path (through the C ABI, liblucille_hip.so) against
the oracle on the same seeded inputs, against the committed reference goldens,
and -- at BASELINE sizes -- through size-independent properties.
Bit-exact for prim ids AND for t/u/v (stronger than the 1e-5 the north star asks)."""
import numpy as np
import pytest

import lucille_amd as la
from lucille_amd import scenes
from oracle import pyoracle as po
from tests.helpers import Model, assert_hits_equal, grid_mesh, load_golden, random_rays

pytestmark = pytest.mark.gpu

VARIANTS = [la.VARIANT_DIRECT, la.VARIANT_SPEC]      # the textbook reference walk, the tuned default


def torch_rays(org, dr):
    import torch
    return (torch.from_numpy(np.ascontiguousarray(org, np.float64)).cuda(),
            torch.from_numpy(np.ascontiguousarray(dr, np.float64)).cuda())


def gpu_closest(acc, org, dr, variant):
    import torch
    o, d = torch_rays(org, dr)
    out = acc.intersect_device(o, d, variant=variant)
    torch.cuda.synchronize()
    return (out[0].cpu().numpy().view(np.uint32), out[1].cpu().numpy(), out[2].cpu().numpy(), out[3].cpu().numpy())


def gpu_any(acc, org, dr, variant):
    import torch
    o, d = torch_rays(org, dr)
    out = acc.intersect_device(o, d, mode=la.MODE_ANY, variant=variant)
    torch.cuda.synchronize()
    return out[0].cpu().numpy()


def make_accel(P, idx, build=None):
    acc = la.HipAccel(0)
    acc.add_mesh(P, idx)
    acc.commit(build=build)
    return acc


@pytest.mark.parametrize("name", ["soup_20k", "soup_3k_fat"])
@pytest.mark.parametrize("variant", VARIANTS)
def test_reference_goldens(name, variant):
    g = load_golden(name)
    P, idx, org, dr = po.soup(int(g["ntri"]), int(g["nrays"]), float(g["half_extent"]), int(g["seed"]))
    acc = make_accel(P, idx)
    assert_hits_equal(gpu_closest(acc, org, dr, variant), (g["prim"], g["t"], g["u"], g["v"]), name)
    assert np.array_equal(gpu_any(acc, org, dr, variant).astype(bool), g["prim"] != po.MISS)


@pytest.mark.parametrize("ntri,nrays,he,seed", [(200000, 300000, 0.005, 21), (50000, 100000, 0.0007, 22), (7, 30001, 0.4, 23)])
@pytest.mark.parametrize("variant", VARIANTS)
def test_oracle_parity_seeded(ntri, nrays, he, seed, variant):
    P, idx, org, dr = po.soup(ntri, nrays, he, seed)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=16)
    acc = make_accel(P, idx)
    assert_hits_equal(gpu_closest(acc, org, dr, variant), exp, "soup %d v%d" % (ntri, variant))
    assert np.array_equal(gpu_any(acc, org, dr, variant).astype(bool), exp[0] != po.MISS)


@pytest.mark.parametrize("wide8", [0, 1])
def test_node_formats(wide8):
    """every node format the kernels can walk -- fp32 2-wide (the textbook variant), 16-bit grid 4-wide (default),
    16-bit grid 8-wide (ray dumps over large scenes; forced here) -- same records"""
    fmt = "wide8=%d" % wide8
    P, idx, org, dr = po.soup(60000, 120000, 0.01, 8)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=16)
    acc = make_accel(P, idx); acc.set_param("wide8", wide8)
    for variant in VARIANTS:
        assert_hits_equal(gpu_closest(acc, org, dr, variant), exp, "format %s variant %d" % (fmt, variant))
        assert np.array_equal(gpu_any(acc, org, dr, variant).astype(bool), exp[0] != po.MISS)
    P, idx = grid_mesh(8, 8)                      # shared vertices / edges: exact-t ties
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    acc = make_accel(P, idx); acc.set_param("wide8", wide8)
    org, dr = random_rays(np.random.default_rng(3), 20000)
    assert_hits_equal(gpu_closest(acc, org, dr, la.VARIANT_DEFAULT), o.intersect(org, dr), "format %s grid" % fmt)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 256, 257, 1000])
def test_ragged_batch_sizes(n):
    P, idx, org, dr = po.soup(5000, 1000, 0.03, 31)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org[:n], dr[:n])
    acc = make_accel(P, idx)
    for variant in VARIANTS:
        assert_hits_equal(gpu_closest(acc, org[:n], dr[:n], variant), exp, "n=%d v%d" % (n, variant))


def test_empty_scene_and_empty_batch():
    import torch
    acc = la.HipAccel(0); acc.add_mesh(np.zeros((0, 3)), np.zeros(0, np.uint32)); acc.commit()
    org, dr = random_rays(np.random.default_rng(0), 100)
    prim, t, u, v = gpu_closest(acc, org, dr, la.VARIANT_DEFAULT)
    assert (prim == la.MISS).all() and (t == 1.0e38).all() and (u == 0).all() and (v == 0).all()
    assert (gpu_any(acc, org, dr, la.VARIANT_DEFAULT) == 0).all()
    P, idx, _, _ = po.soup(100, 1, 0.1, 1)
    acc2 = make_accel(P, idx)
    e = torch.empty((0, 3), dtype=torch.float64, device="cuda")
    out = acc2.intersect_device(e, e)
    assert out[0].numel() == 0
    # accel with no meshes at all
    acc3 = la.HipAccel(0); acc3.commit()
    assert (acc3.intersect_host(org, dr)[0] == la.MISS).all()


def test_host_batch_and_single_ray_entry_points():
    """lh_accel_intersect_host / lh_accel_intersect1 == accel_intersect_func semantics"""
    P, idx, org, dr = po.soup(3000, 2000, 0.05, 41)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr)
    acc = make_accel(P, idx)
    assert_hits_equal(acc.intersect_host(org, dr), exp, "host batch")
    assert np.array_equal(acc.intersect_host(org, dr, mode=la.MODE_ANY).astype(bool), exp[0] != po.MISS)
    for i in range(25):
        hit, prim, t, u, v = acc.intersect1(org[i], dr[i])
        assert hit == int(exp[0][i] != po.MISS) and prim == exp[0][i] and t == exp[1][i] and u == exp[2][i] and v == exp[3][i]


def test_large_host_batches_are_pipelined_in_chunks():
    """>= 2 M rays from host arrays go through a ring of pinned staging blocks (a batch of less than four blocks in four chunks): same records
    as the device path, ragged last chunk, both modes, optional outputs left out"""
    import torch
    P, idx, org, dr = po.soup(200000, 5 * (1 << 20) + 12345, 0.005, 4)
    acc = make_accel(P, idx)
    o_, d_ = torch_rays(org, dr)
    dev = acc.intersect_device(o_, d_); occ_dev = acc.intersect_device(o_, d_, mode=la.MODE_ANY)[0]
    torch.cuda.synchronize()
    got = acc.intersect_host(org, dr)
    assert np.array_equal(got[0], dev[0].cpu().numpy().view(np.uint32))
    for k in (1, 2, 3):
        assert np.array_equal(got[k], dev[k].cpu().numpy())
    assert np.array_equal(acc.intersect_host(org, dr, mode=la.MODE_ANY), occ_dev.cpu().numpy())
    n = len(org); prim = np.empty(n, np.uint32)                      # only prim requested
    rc = acc.L.lh_accel_intersect_host(acc.h, n, org.ctypes.data, dr.ctypes.data, prim.ctypes.data, None, None, None, None, 0)
    assert rc == 0 and np.array_equal(prim, got[0])
    ex = po.Oracle(); ex.add_mesh(P, idx); ex.build()
    assert_hits_equal(tuple(g[:100000] for g in got), ex.intersect(org[:100000], dr[:100000], nthreads=16), "pipelined host path prefix")


@pytest.mark.parametrize("chunk,depth,threads", [(131072, 2, 3), (196608, 3, 0), (65536, 8, 8)])
def test_pipelined_host_batches_reuse_their_ring_of_staging_blocks(monkeypatch, chunk, depth, threads):
    """the same path with small chunks: 2.3 M rays are 12 .. 36 chunks through a ring of 2 / 3 / 8 staging blocks (every block is
    reused, the last chunk is ragged), copied by the pool's threads or by the caller alone; two such calls on one accelerator"""
    import torch
    monkeypatch.setenv("LH_PIPE_CHUNK", str(chunk)); monkeypatch.setenv("LH_PIPE_DEPTH", str(depth)); monkeypatch.setenv("LH_COPY_THREADS", str(threads))
    P, idx, org, dr = po.soup(60000, (1 << 21) + 250001, 0.01, 21)
    acc = make_accel(P, idx)
    o_, d_ = torch_rays(org, dr)
    dev = acc.intersect_device(o_, d_); occ_dev = acc.intersect_device(o_, d_, mode=la.MODE_ANY)[0]
    torch.cuda.synchronize()
    for _ in range(2):
        got = acc.intersect_host(org, dr)
        assert np.array_equal(got[0], dev[0].cpu().numpy().view(np.uint32))
        for k in (1, 2, 3):
            assert np.array_equal(got[k], dev[k].cpu().numpy())
        assert np.array_equal(acc.intersect_host(org, dr, mode=la.MODE_ANY), occ_dev.cpu().numpy())


def test_concurrent_host_threads_like_the_reference_render_threads():
    """lucille calls accel->intersect from up to 16 pthreads at once (render.c:1043-1105): batches of
    different sizes and single rays from 12 threads through ONE accelerator, every record exact"""
    import threading
    P, idx, org, dr = po.soup(20000, 60000, 0.02, 77)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=8)
    acc = make_accel(P, idx)
    errors = []

    def worker(k):
        try:
            rng = np.random.default_rng(k)
            for it in range(6):
                n = int(rng.integers(1, 9000)); a = int(rng.integers(0, len(org) - n))
                got = acc.intersect_host(org[a:a + n], dr[a:a + n])
                for c in range(4):
                    if not np.array_equal(got[c], exp[c][a:a + n]):
                        errors.append("thread %d batch %d field %d" % (k, it, c)); return
                occ = acc.intersect_host(org[a:a + n], dr[a:a + n], mode=la.MODE_ANY)
                if not np.array_equal(occ.astype(bool), exp[0][a:a + n] != po.MISS):
                    errors.append("thread %d any-hit %d" % (k, it)); return
                for i in rng.integers(0, len(org), 15):
                    hit, prim, t, u, v = acc.intersect1(org[i], dr[i])
                    if not (prim == exp[0][i] and t == exp[1][i] and u == exp[2][i] and v == exp[3][i]):
                        errors.append("thread %d single ray %d" % (k, i)); return
        except Exception as e:                                   # noqa: BLE001
            errors.append("thread %d: %r" % (k, e))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(12)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errors, errors[:3]


def test_bad_input_is_refused_or_harmless():
    """a NaN / infinite vertex is refused at commit with a message (the fp32 filter cannot bound it);
    NaN / infinite / zero rays inside a batch come back as misses and do not disturb their neighbours"""
    P, idx, org, dr = po.soup(5000, 4000, 0.03, 12)
    for bad in (np.nan, np.inf, -1e300):
        Q = P.copy(); Q[101, 2] = bad
        acc = la.HipAccel(0); acc.add_mesh(Q, idx)
        with pytest.raises(la.LucilleHipError, match="NaN, infinite"):
            acc.commit()
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr)
    org2, dr2 = org.copy(), dr.copy()
    weird = np.arange(0, len(org), 37)
    dr2[weird[0::4]] = np.nan; org2[weird[1::4]] = np.inf; dr2[weird[2::4]] = 0.0; dr2[weird[3::4], 0] = -np.inf
    acc = make_accel(P, idx)
    for variant in (la.VARIANT_DEFAULT, la.VARIANT_DIRECT):
        got = gpu_closest(acc, org2, dr2, variant)
        keep = np.ones(len(org), bool); keep[weird] = False
        assert_hits_equal(tuple(g[keep] for g in got), tuple(e[keep] for e in exp), "neighbours of bad rays")
        assert (got[0][weird[0::4]] == la.MISS).all() and (got[0][weird[1::4]] == la.MISS).all()
        occ = gpu_any(acc, org2, dr2, variant)
        assert np.array_equal(occ[keep].astype(bool), exp[0][keep] != po.MISS)


def test_multi_mesh_prim_lookup():
    P1, i1, org, dr = po.soup(300, 5000, 0.1, 11)
    P2, i2, _, _ = po.soup(500, 1, 0.1, 22)
    P2w = np.concatenate([P2, np.ones((P2.shape[0], 1))], 1)      # lucille's double[4] stride
    acc = la.HipAccel(0); acc.add_mesh(P1, i1); acc.add_mesh(P2w, i2); acc.commit()
    o = po.Oracle(); o.add_mesh(P1, i1); o.add_mesh(P2, i2); o.build()
    exp = o.intersect(org, dr)
    assert_hits_equal(acc.intersect_host(org, dr), exp, "two meshes")
    assert acc.prim_lookup(299) == (0, 3 * 299) and acc.prim_lookup(300) == (1, 0) and acc.prim_lookup(799) == (1, 3 * 499)
    with pytest.raises(la.LucilleHipError):
        acc.prim_lookup(800)


def test_tolerance_band_geometry():
    """axis-aligned shared-vertex grid, rays through vertices/edges/diagonals, far
    origins, AO-style origins 1e-6 above the surface: the fp32 filter's band cases"""
    P, idx = grid_mesh(8, 8)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    acc = make_accel(P, idx)
    xs = np.linspace(0.0, 1.0, 33); tx, ty = np.meshgrid(xs, xs)
    tgt = np.stack([tx.ravel(), ty.ravel(), np.zeros(tx.size)], 1)
    rng = np.random.default_rng(0)
    for oz in (1.0, 37.5, 1e-3):
        org = np.tile(np.array([[0.3, 0.45, oz]]), (tgt.shape[0], 1)) + rng.uniform(-0.2, 0.2, (tgt.shape[0], 3)) * [1, 1, 0]
        dr = tgt - org
        exp = o.intersect(org, dr)
        for variant in VARIANTS:
            got = gpu_closest(acc, org, dr, variant)
            # exact-t ties included: the reference-order tree on the device picks the reference's winner
            assert_hits_equal(got, exp, "grid oz=%g v%d" % (oz, variant))
            assert np.array_equal(gpu_any(acc, org, dr, variant).astype(bool), exp[0] != po.MISS)
    n = 50000
    org = np.stack([rng.uniform(0, 1, n), rng.uniform(0, 1, n), np.full(n, 1e-6)], 1)
    d = rng.normal(size=(n, 3)); d[:, 2] = np.abs(d[:, 2]); d /= np.linalg.norm(d, axis=1, keepdims=True)
    for z in (1e-6, 0.0, -1e-9):
        org[:, 2] = z
        exp = o.intersect(org, d, nthreads=8)
        assert_hits_equal(gpu_closest(acc, org, d, la.VARIANT_DEFAULT), exp, "surface z=%g" % z)
        assert np.array_equal(gpu_any(acc, org, d, la.VARIANT_DEFAULT).astype(bool), exp[0] != po.MISS)


def test_counters_match_host_model():
    """the COUNT kernel variant reports the node visits / triangle tests the roofline
    formula uses; the host model of the same algorithm must count the same"""
    import torch
    P, idx, org, dr = po.soup(30000, 20000, 0.005, 61)
    acc = make_accel(P, idx)
    o_, d_ = torch_rays(org, dr)
    _, cnt = acc.intersect_device(o_, d_, counters=True, variant=la.VARIANT_DIRECT)     # 2-wide fp32 nodes
    m = Model(P, idx)
    _, mc = m.trace(org, dr, qnodes=0)
    assert cnt["rays"] == mc["rays"] == 20000
    # v_rcp_f32 vs 1/x may move a handful of band decisions: counts agree to 1e-4
    for k in ("nodes", "tris"):
        assert abs(cnt[k] - mc[k]) <= 1e-4 * mc[k] + 4, (k, cnt[k], mc[k])
    # default kernel: 4-wide nodes, leaves parked and tested in batches (the culling bound
    # shrinks a little later than in the sequential model): same order of work, within 5 %
    _, c4 = acc.intersect_device(o_, d_, counters=True)
    _, m4 = m.trace(org, dr, qnodes=2)
    for k in ("nodes", "tris"):
        assert abs(c4[k] - m4[k]) <= 0.05 * m4[k], (k, c4[k], m4[k])
    assert c4["nodes"] < 0.6 * cnt["nodes"]          # the 4-wide walk visits about half the records


@pytest.mark.parametrize("build", ["host", "auto"])
def test_full_size_properties_soup_1m(build):
    """BASELINE config 3 scale (1M triangles): properties that need no CPU reference:
    any-hit == (closest hit exists); variants agree bit for bit; t >= 0; the hit point
    lies inside the hit triangle's box; first 200k rays bit-exact vs the oracle.
    build "auto": what lh_accel_commit chooses by itself at this size -- the device builders (there the textbook
    variant runs as the default walk: the 2-wide fp32 nodes exist only in the host builder)"""
    import torch
    ntri, nrays = 1000000, 4000000
    P, idx, org, dr = po.soup(ntri, nrays)
    acc = make_accel(P, idx, build)
    info = acc.info()
    assert (info["nnodes"] == info["nnodes_traversal"]) == (build == "auto")     # a device-built scene has no 2-wide nodes of its own
    o_, d_ = torch_rays(org, dr)
    res = {}
    for variant in VARIANTS:
        out = acc.intersect_device(o_, d_, variant=variant)
        occ = acc.intersect_device(o_, d_, mode=la.MODE_ANY, variant=variant)[0]
        torch.cuda.synchronize()
        res[variant] = [x.clone() for x in out]
        hit = out[0] != -1
        assert torch.equal(hit, occ.bool())
        assert (out[1][hit] >= 0).all() and (out[1][~hit] == 1.0e38).all()
    for variant in VARIANTS[1:]:
        for a, b in zip(res[VARIANTS[0]], res[variant]):
            assert torch.equal(a, b)
    prim = res[0][0].cpu().numpy().view(np.uint32); t = res[0][1].cpu().numpy()
    hit = prim != la.MISS
    tri = P[idx].reshape(-1, 3, 3)[prim[hit]]
    X = org[hit] + dr[hit] * t[hit][:, None]
    assert (X >= tri.min(1) - 1e-9).all() and (X <= tri.max(1) + 1e-9).all()
    n = 200000
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org[:n], dr[:n], nthreads=32)
    assert_hits_equal(tuple(x.cpu().numpy()[:n] for x in res[la.VARIANT_SPEC]), exp, "soup-1M prefix")


def test_full_size_properties_soup_10m():
    """BASELINE config 5 scale (S-soup-10M: 10 M triangles, half-extent 0.002): properties over 8 M
    rays, and the first 200 k rays bit for bit against the oracle (its single-threaded build of this
    scene is what takes most of this test's ~45 s)"""
    import torch
    ntri, nrays = 10000000, 8000000
    P, idx, org, dr = po.soup(ntri, nrays, 0.002)
    acc = make_accel(P, idx, "host")                            # (the 2-wide direct walk below needs the host builder's nodes)
    info = acc.info()
    assert info["ntriangles"] == ntri
    o_, d_ = torch_rays(org, dr)
    out = acc.intersect_device(o_, d_)
    occ = acc.intersect_device(o_, d_, mode=la.MODE_ANY)[0]
    out0 = acc.intersect_device(o_[:500000].contiguous(), d_[:500000].contiguous(), variant=0)
    torch.cuda.synchronize()
    hit_t = out[0] != -1
    assert torch.equal(hit_t, occ.bool())
    assert (out[1][hit_t] >= 0).all() and (out[1][~hit_t] == 1.0e38).all()
    for a, b in zip(out, out0):
        assert torch.equal(a[:500000], b)                       # 4-wide speculative walk == 2-wide direct walk
    prim = out[0].cpu().numpy().view(np.uint32); t = out[1].cpu().numpy()
    hit = prim != la.MISS
    assert 0.5 < hit.mean() < 0.999
    tri = P[idx].reshape(-1, 3, 3)[prim[hit]]
    X = org[hit] + dr[hit] * t[hit][:, None]
    assert (X >= tri.min(1) - 1e-9).all() and (X <= tri.max(1) + 1e-9).all()
    n = 200000
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org[:n], dr[:n], nthreads=32)
    assert_hits_equal(tuple(x.cpu().numpy()[:n] for x in out), exp, "soup-10M prefix")


@pytest.mark.parametrize("kind,log_scale,off", [("soup", -3, 0.0), ("soup", 3, -800.0), ("slivers", 0, 10.0), ("axis", 2, 500.0),
                                                ("degenerate", -1, -3.0), ("slivers", -2, 0.5)])
def test_scaled_translated_degenerate_scenes(kind, log_scale, off):
    """magnitude robustness of the conservative filter on the device: tiny / huge / far-from-origin
    scenes, sliver and degenerate triangles, axis-parallel and unnormalised rays (same generator as
    the CPU fuzz test, tests/test_filter_fuzz.py)"""
    from tests.test_filter_fuzz import make_scene
    rng = np.random.default_rng(1234)
    scale = 10.0 ** log_scale; offset = np.array([off, -0.5 * off, 0.25 * off])
    P, idx = make_scene(rng, 300, scale, offset, kind)
    n = 20000
    tgt = rng.uniform(-0.1, 1.1, (n, 3)) * scale + offset
    org = tgt + rng.normal(size=(n, 3)) * scale * 3.0
    # rays aimed exactly at vertices / edge midpoints / centroids of proper triangles, origins on triangles
    T = P[idx].reshape(-1, 3, 3)
    area = np.linalg.norm(np.cross(T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]), axis=1)
    proper = np.nonzero(area > 1e-9 * scale * scale)[0]
    pick = proper[rng.integers(0, len(proper), n)]
    tgt[::5] = T[pick[::5], rng.integers(0, 3, len(pick[::5]))]
    tgt[1::7] = 0.5 * (T[pick[1::7], 0] + T[pick[1::7], 1])
    tgt[2::9] = T[pick[2::9]].mean(axis=1)
    org[3::23] = T[pick[3::23], 2]
    org[4::29] = T[pick[4::29]].mean(axis=1)
    dr = tgt - org
    dr[::11, 0] = 0.0; dr[1::13, 1] = 1e-3 * scale; dr[2::17] *= 1e-3; dr[3::19, 2] = 1e-20
    keep = np.linalg.norm(dr, axis=1) > 0
    org, dr = org[keep], dr[keep]
    ok = np.abs(dr[:, 1]) > 1e-14
    org, dr = np.ascontiguousarray(org[ok]), np.ascontiguousarray(dr[ok])
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=8)
    acc = make_accel(P, idx)
    for variant in VARIANTS:
        assert_hits_equal(gpu_closest(acc, org, dr, variant), exp, "%s 1e%d v%d" % (kind, log_scale, variant))
    assert np.array_equal(gpu_any(acc, org, dr, la.VARIANT_DEFAULT).astype(bool), exp[0] != po.MISS)


def test_eight_wide_walk_parity():
    """the 8-wide 16-bit-grid nodes (lh_q8node_t, 128-byte records: what ray dumps walk on scenes larger than the Infinity
    Cache; forced here with set_param("wide8", 1)): goldens, seeded soups against the oracle, exact-t ties, deep chains,
    vertex-aimed rays, ragged batches, a capped stack (overflowing rays finished over the 4-wide nodes) -- same records"""
    import torch
    from tests.helpers import chain_scene, vertex_aimed_rays
    def wide(P, idx):
        acc = make_accel(P, idx); acc.set_param("wide8", 1)
        assert acc.dump_node_bytes() == 128
        return acc
    for name in ("soup_20k", "soup_3k_fat"):
        g = load_golden(name)
        P, idx, org, dr = po.soup(int(g["ntri"]), int(g["nrays"]), float(g["half_extent"]), int(g["seed"]))
        acc = wide(P, idx)
        assert_hits_equal(gpu_closest(acc, org, dr, la.VARIANT_DEFAULT), (g["prim"], g["t"], g["u"], g["v"]), name)
        assert np.array_equal(gpu_any(acc, org, dr, la.VARIANT_DEFAULT).astype(bool), g["prim"] != po.MISS)
        acc.close()
    for ntri, nrays, he, seed in [(200000, 300000, 0.005, 21), (50000, 100000, 0.0007, 22), (7, 30001, 0.4, 23), (1, 5000, 0.3, 24)]:
        P, idx, org, dr = po.soup(ntri, nrays, he, seed)
        o = po.Oracle(); o.add_mesh(P, idx); o.build()
        exp = o.intersect(org, dr, nthreads=16)
        acc = wide(P, idx)
        assert_hits_equal(gpu_closest(acc, org, dr, la.VARIANT_DEFAULT), exp, "8-wide soup %d" % ntri)
        assert np.array_equal(gpu_any(acc, org, dr, la.VARIANT_DEFAULT).astype(bool), exp[0] != po.MISS)
        for n in (1, 63, 65, 257):
            assert_hits_equal(gpu_closest(acc, org[:n], dr[:n], la.VARIANT_DEFAULT), tuple(x[:n] for x in exp), "8-wide ragged %d" % n)
        if ntri == 200000:
            acc.set_param("stack_cap", 16)          # overflow path: k_overflow_fix over the 4-wide nodes
            do = torch.from_numpy(org).cuda(); dd = torch.from_numpy(dr).cuda()
            _, c = acc.intersect_device(do, dd, counters=True)
            assert c["retraced"] > 0
            assert_hits_equal(gpu_closest(acc, org, dr, la.VARIANT_DEFAULT), exp, "8-wide capped stack")
            assert np.array_equal(gpu_any(acc, org, dr, la.VARIANT_DEFAULT).astype(bool), exp[0] != po.MISS)
        acc.close()
    P, idx = grid_mesh(8, 8)                      # shared vertices / edges: exact-t ties
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    acc = wide(P, idx)
    org, dr = random_rays(np.random.default_rng(3), 20000)
    assert_hits_equal(gpu_closest(acc, org, dr, la.VARIANT_DEFAULT), o.intersect(org, dr), "8-wide grid")
    vo, vd = vertex_aimed_rays(np.random.default_rng(5), P, idx, 20000)
    assert_hits_equal(gpu_closest(acc, vo, vd, la.VARIANT_DEFAULT), o.intersect(vo, vd), "8-wide vertex-aimed")
    acc.close()
    P, idx = chain_scene(40)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    acc = wide(P, idx)
    org, dr = random_rays(np.random.default_rng(6), 20000, lo=-1.0, hi=2.0)
    assert_hits_equal(gpu_closest(acc, org, dr, la.VARIANT_DEFAULT), o.intersect(org, dr), "8-wide chain")
    acc.close()


def test_wide_walk_is_chosen_by_footprint():
    """hot set (4-wide nodes + 48-byte triangle records) above 256 MiB -> ray dumps walk the 8-wide nodes; the records equal the
    4-wide walk's and the oracle's"""
    P, idx, st = scenes.soup_triangles(4000000, 0.003)
    org, dr, _ = scenes.soup_rays(400000, st)
    acc = make_accel(P, idx)
    info = acc.info()
    assert info["nnodes_traversal"] * 64 + info["ntriangles"] * 48 > 256 << 20 and acc.dump_node_bytes() == 128
    a = gpu_closest(acc, org, dr, la.VARIANT_DEFAULT); oa = gpu_any(acc, org, dr, la.VARIANT_DEFAULT)
    acc.set_param("wide8", 0)
    assert acc.dump_node_bytes() == 64
    b = gpu_closest(acc, org, dr, la.VARIANT_DEFAULT)
    assert_hits_equal(a, b, "8-wide == 4-wide")
    assert np.array_equal(oa, gpu_any(acc, org, dr, la.VARIANT_DEFAULT))
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org[:100000], dr[:100000], nthreads=16)
    assert_hits_equal(tuple(x[:100000] for x in a), exp, "8-wide, 4 M triangles")
    acc.close()
    small = make_accel(*po.soup(20000, 10, 0.01, 3)[:2])
    assert small.dump_node_bytes() == 64
    small.close()


@pytest.mark.parametrize("budget", [1, 3, 17])
def test_cooperative_walk_keeps_every_bit(budget):
    """set_param("ray_budget", b): a ray that visits more than b nodes leaves the persistent kernel and is finished by the
    wave-cooperative walk (one wave per ray, idle lanes take over the bottom of busy lanes' stacks; per-lane bests merged with
    the reference's tie rule).  With tiny budgets nearly every ray takes that path: records must stay the oracle's bits --
    seeded soups, exact-t ties on a shared-edge grid, vertex-aimed rays, a deep chain, any-hit, and the 8-wide nodes."""
    import torch
    from tests.helpers import chain_scene, vertex_aimed_rays
    def check(P, idx, org, dr, label, wide8=0, expect_coop=False):
        o = po.Oracle(); o.add_mesh(P, idx); o.build()
        exp = o.intersect(org, dr, nthreads=16)
        acc = make_accel(P, idx); acc.set_param("wide8", wide8); acc.set_param("ray_budget", budget)
        do = torch.from_numpy(org).cuda(); dd = torch.from_numpy(dr).cuda()
        out, c = acc.intersect_device(do, dd, counters=True)
        assert c["retraced"] > 0.2 * org.shape[0] or not expect_coop, (label, c)          # the cooperative walk really ran
        assert_hits_equal(tuple(x.cpu().numpy().view(np.uint32) if k == 0 else x.cpu().numpy() for k, x in enumerate(out)), exp, label)
        assert np.array_equal(gpu_any(acc, org, dr, la.VARIANT_DEFAULT).astype(bool), exp[0] != po.MISS), label
        for n in (1, 63, 65):
            assert_hits_equal(gpu_closest(acc, org[:n], dr[:n], la.VARIANT_DEFAULT), tuple(x[:n] for x in exp), label + " ragged")
        acc.close()
    P, idx, org, dr = po.soup(60000, 120000, 0.01, 8)
    check(P, idx, org, dr, "soup b%d" % budget, expect_coop=True)
    check(P, idx, org, dr, "soup 8-wide b%d" % budget, wide8=1, expect_coop=budget < 10)
    P, idx, org, dr = po.soup(3000, 40000, 0.08, 9)            # fat triangles: many candidates per ray
    check(P, idx, org, dr, "fat soup b%d" % budget)
    P, idx = grid_mesh(8, 8)                                   # shared vertices / edges: exact-t ties across lanes' subtrees
    org, dr = random_rays(np.random.default_rng(3), 20000)
    check(P, idx, org, dr, "grid b%d" % budget)
    vo, vd = vertex_aimed_rays(np.random.default_rng(5), P, idx, 20000)
    check(P, idx, vo, vd, "grid vertex-aimed b%d" % budget)
    P, idx = chain_scene(40)
    org, dr = random_rays(np.random.default_rng(6), 20000, lo=-1.0, hi=2.0)
    check(P, idx, org, dr, "chain b%d" % budget)


def test_cooperative_walk_in_the_fused_ao_stage():
    """the AO frame must not depend on the visit budget: tiny budgets send most AO rays through the fix-up queue and the
    cooperative walk (SRC 1: rays regenerated from (slot, sample)); a queue that overflows falls back to the materialised
    stage -- same frame bit for bit"""
    import torch
    from lucille_amd import render, scenes
    g = load_golden("ao_c1")
    acc = la.HipAccel(0)
    for k in range(int(g["ngeoms"])):
        P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 3); acc.add_mesh(P, I)
    acc.commit()
    c = g["camera"]; cam = la.Camera.make(200, 150, c[16], c[:16], int(c[19]))
    ref_img, ref_stats = render.render_ao_frame(acc, cam, 2, 16, tile=200, seed=5)
    for budget in (2, 9, 40):
        acc.set_param("ray_budget", budget)
        img, stats = render.render_ao_frame(acc, cam, 2, 16, tile=200, seed=5)
        torch.cuda.synchronize()
        assert stats == ref_stats and torch.equal(img, ref_img), budget
    acc.set_param("ray_budget", 1)                 # 960 k AO rays, all out of budget: fits the queue (2^20)
    big = la.Camera.make(400, 300, c[16], c[:16], int(c[19]))
    acc.set_param("ray_budget", 256)
    ref_big, st_big = render.render_ao_frame(acc, big, 2, 16, tile=400, seed=5)
    acc.set_param("ray_budget", 1)                 # ~4 M AO rays out of budget: the queue overflows -> materialised stage
    img, stats = render.render_ao_frame(acc, big, 2, 16, tile=400, seed=5)
    torch.cuda.synchronize()
    assert stats == st_big and torch.equal(img, ref_big)
    acc.close()
