"""Generate the committed golden fixtures by running the COMPILED REFERENCE
(oracle/_ref, built from /root/reference by oracle/Makefile).  Runs only where
/root/reference exists.  Fixtures are data (inputs by seed, expected outputs);
no reference source is stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def soup_fixture(name, ntri, nrays, half_extent):
    """S-soup (SURVEY.md Appendix C generator): inputs are reproducible from
    (seed, ntri, nrays, half_extent); outputs are the reference's hit records,
    per-batch traversal counters (RI_BVH_TRACE_STATISTICS build) and tree shape."""
    P, idx, org, dr = po.soup(ntri, nrays, half_extent)
    ref = po.RefLib(stat=True)
    ref.add_mesh(P, idx)
    ref.build()
    prim, t, u, v, cnt = ref.intersect(org, dr, counters=True)
    tree = ref.tree_stats()
    bmin, bmax = ref.bbox()
    # exact-t ties between two different triangles would make the winner
    # tree-dependent; record that none occur
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    bf = o.brute_force(org[:2000], dr[:2000], nthreads=os.cpu_count())
    assert np.array_equal(bf[0], prim[:2000]) and np.array_equal(bf[1], t[:2000])
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), seed=np.uint64(po.SOUP_SEED), ntri=ntri, nrays=nrays,
        half_extent=half_extent, prim=prim, t=t, u=u, v=v,
        counters=np.array([cnt[k] for k in ("ninner", "nleaf", "ntested", "nhit", "nrays")], np.uint64),
        tree=np.array([tree[k] for k in ("ninner", "nleaf", "max_depth", "max_leaf_tris", "ntriangles")], np.uint64),
        bmin=bmin, bmax=bmax)
    print(name, "hits", int((prim != po.MISS).sum()), "of", nrays, "sum t", float(t[prim != po.MISS].sum()), cnt, tree)


def state_scene(seed=2024):
    """three small meshes exercising every branch of ri_intersection_state_build (intersection_state.c:99-248):
    0: normals + tangents + binormals + colours + shared texcoords, two_side; 1: normals only + unshared texcoords;
    2: nothing but positions (+ colours).  Inputs are reproducible from the seed."""
    rng = np.random.default_rng(seed)
    meshes = []
    for k, (ntri, off) in enumerate(((60, 0.0), (50, 0.7), (40, -0.6))):
        c = rng.uniform(-0.5, 0.5, (ntri, 1, 3)) + np.array([off, 0.1 * k, 0.0])
        tri = c + rng.uniform(-0.15, 0.15, (ntri, 3, 3))
        npos = 3 * ntri // 2                               # shared vertices: indices point into a smaller pool
        P = tri.reshape(-1, 3)[:npos].copy()
        idx = rng.integers(0, npos, 3 * ntri).astype(np.uint32)
        m = {"P": P, "idx": idx, "two_side": 0}
        unit = lambda a: a / np.linalg.norm(a, axis=1, keepdims=True)
        if k == 0:
            idx2 = np.concatenate([idx, idx[::-1]]).astype(np.uint32)      # two_side: second half = back faces
            m.update(idx=idx2, two_side=1, N=unit(rng.normal(size=(npos, 3))), T=unit(rng.normal(size=(npos, 3))),
                     B=unit(rng.normal(size=(npos, 3))), C=rng.uniform(0, 1, (npos, 3)), ST=rng.uniform(0, 4, (npos, 2)))
        elif k == 1:
            m.update(N=unit(rng.normal(size=(npos, 3))), STU=rng.uniform(-1, 1, (idx.shape[0], 2)))
        else:
            m.update(C=rng.uniform(0, 1, (npos, 3)))
        meshes.append(m)
    n = 6000
    org = rng.uniform(-2.5, 2.5, (n, 3)); tgt = rng.uniform(-0.6, 0.9, (n, 3)); tgt[:, 0] += rng.choice([0.0, 0.7, -0.6], n)
    return meshes, org, tgt - org


def apply_state_scene(target, meshes, is_ref):
    for k, m in enumerate(meshes):
        target.add_mesh(m["P"], m["idx"])
        if is_ref:
            target.set_attribute(k, -1, m.get("N"), two_side=m["two_side"])
        elif "N" in m or m["two_side"]:
            target.set_normals(k, m.get("N"), m["two_side"])
        for kind, key in ((0, "C"), (1, "T"), (2, "B"), (3, "ST"), (4, "STU")):
            if key in m:
                target.set_attribute(k, kind, m[key])


def state_fixture(name="state_attr"):
    """the whole hit record of the compiled reference (ri_raytrace -> ri_intersection_state_build) for a scene with
    colours / tangents / binormals / texture coordinates: prim + 24 doubles per ray"""
    meshes, org, dr = state_scene()
    ref = po.RefLib(); apply_state_scene(ref, meshes, True); ref.build()
    prim, st = ref.state_batch(org, dr)
    o = po.Oracle(); apply_state_scene(o, meshes, False); o.build()
    op, ost = o.state_batch(org, dr)
    assert np.array_equal(op, prim) and np.array_equal(ost, st), "oracle restatement != compiled reference"
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=2024, prim=prim, state=st)
    print(name, "hits", int((prim != po.MISS).sum()), "of", org.shape[0], "inside", int(st[:, 23].sum()))


def ao_fixture(name, rib, width, height, gather_nsamples, pixel_samples=1):
    """The reference's own AO render of one of its example scenes (single thread, Ri C API
    driven by oracle/ref_rib.py): the triangles it actually traced (after its RIB ingest),
    its camera, every ray it issued IN ORDER with the hit record, and the float image.
    Rays are not stored: the oracle regenerates them bit for bit (checked here) and the
    fixture pins their SHA-256."""
    import hashlib
    from oracle import ref_rib
    tmp = os.path.join("/tmp", name + "_ref.npz")
    r = ref_rib.render_rib_subprocess(rib, tmp, width=width, height=height, gather_nsamples=gather_nsamples,
                                      pixel_samples=pixel_samples)
    R = r["records"]
    ng = int(r["ngeoms"])
    base = np.cumsum([0] + [len(r["idx%d" % g]) // 3 for g in range(ng)])
    prim = np.where(R["hit"] == 1, base[np.minimum(R["geom"], ng - 1)] + R["index"] // 3, po.MISS).astype(np.uint32)
    o = po.Oracle()
    for g in range(ng):
        o.add_mesh(r["pos%d" % g], r["idx%d" % g])
        if ("nrm%d" % g) in r.files:
            o.set_normals(g, r["nrm%d" % g], int(r["two_side%d" % g]))
    o.build()
    cam = po.Camera.from_ref(r["camera"])
    img, rec = o.render_ao(cam, pixel_samples, gather_nsamples)
    assert np.array_equal(rec["org"], R["org"]) and np.array_equal(rec["dir"], R["dir"]), "oracle ray stream != reference"
    assert np.array_equal(rec["prim"], prim) and np.array_equal(rec["t"], R["t"])
    assert np.array_equal(img, r["image"])
    hit = prim != po.MISS
    ties = int((o.count_equal_t(R["org"][hit], R["dir"][hit], R["t"][hit]) >= 2).sum())
    d = {"width": width, "height": height, "gather_nsamples": gather_nsamples, "pixel_samples": pixel_samples,
         "camera": r["camera"], "ngeoms": ng, "image": r["image"], "prim": prim,
         "t_hit": R["t"][hit], "u_hit": R["u"][hit], "v_hit": R["v"][hit], "nrays": len(R),
         "rays_sha256": hashlib.sha256(R["org"].tobytes() + R["dir"].tobytes()).hexdigest(), "exact_t_ties": ties}
    for g in range(ng):
        d["pos%d" % g] = r["pos%d" % g]; d["idx%d" % g] = r["idx%d" % g]; d["two_side%d" % g] = r["two_side%d" % g]
        if ("nrm%d" % g) in r.files:
            d["nrm%d" % g] = r["nrm%d" % g]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "rays", len(R), "hits", int(hit.sum()), "tris", int(base[-1]), "exact-t ties among hits", ties,
          "image mean", float(r["image"].mean()))


def beam_fixture(name, ntri, half_extent, seed, nbeams):
    """ri_beam_set + ri_bvh_intersect_beam_visibility of the compiled reference on seeded beams
    (tests.helpers.random_beams) over a seeded S-soup: expected classes only."""
    from tests.helpers import random_beams
    P, idx, _, _ = po.soup(ntri, 1, half_extent, seed)
    ref = po.RefLib(); ref.add_mesh(P, idx); ref.build()
    out = {}
    for spread in (0.001, 0.01, 0.05):
        org, d = random_beams(np.random.default_rng(int(spread * 1e6) + seed), nbeams, spread)
        out["res_%g" % spread] = ref.beam_visibility(org, d)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), ntri=ntri, half_extent=half_extent, seed=seed, nbeams=nbeams, **out)
    print(name, {k: np.bincount(v + 1, minlength=4).tolist() for k, v in out.items()})


def beam_raster_fixture(name="beam_raster"):
    """ri_beam_set + ri_raster_plane_setup + ri_bvh_invalidate_cache + ri_bvh_intersect_beam of the compiled reference
    (ref_harness.c lref_beam_raster, run in a child process: the path asserts and writes unchecked) on the seeded cases of
    tests.helpers.raster_case: expected plane->t per beam.  Inputs are reproducible from CASES."""
    from tests.helpers import raster_case, RASTER_GOLDEN_CASES
    out = {}
    for k, kw in enumerate(RASTER_GOLDEN_CASES):
        c = raster_case(**kw)
        beams = [(c["org"][i], c["dirs"][i], c["width"], c["height"], c["frame"], c["corners"][i], c["eye"], c["fov"])
                 for i in range(c["org"].shape[0])]
        res = po.ref_beam_raster_child([(c["P"], c["idx"])], beams)
        assert all(r is not None for r in res), "the reference died on a beam of case %d" % k
        out["rc%d" % k] = np.array([r[0] for r in res], np.int32)
        out["t%d" % k] = np.stack([r[1] for r in res])
        print(name, "case", k, kw, "pixels written", [int((r[1] != 0).sum()) for r in res])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), ncases=len(RASTER_GOLDEN_CASES), **out)


def rib_fixture():
    """What the reference's RenderMan front end makes of the RIB files under tests/golden/rib/
    (its geoms after ingest, in order, and its camera), captured through its own Ri C API
    (oracle/ref_rib.py): the expected output of the product's RIB reader (lh_rib.c).
    The .rib files are scene DATA: the reference's parser-test inputs (tests/ribparse/), two of
    its example scenes, and synth.rib written for this repo."""
    from oracle import ref_rib
    d = {}
    for name in ("ambient_occlusion", "synth", "tut1"):
        tmp = os.path.join("/tmp", "rib_" + name + ".npz")
        r = ref_rib.render_rib_subprocess(os.path.join(OUT, "rib", name + ".rib"), tmp, width=8, height=8, gather_nsamples=1,
                                          pixel_samples=1, record=False)
        ng = int(r["ngeoms"]); d[name + "_ngeoms"] = ng
        d[name + "_camera"] = np.append(r["camera"][[*range(17), 19]], float(r["ortho"]))   # c2w[16], flength, is_rh, ortho
        for g in range(ng):
            d["%s_pos%d" % (name, g)] = r["pos%d" % g]; d["%s_idx%d" % (name, g)] = r["idx%d" % g]
            d["%s_two%d" % (name, g)] = r["two_side%d" % g]
            if "nrm%d" % g in r:
                d["%s_nrm%d" % (name, g)] = r["nrm%d" % g]
        print("rib", name, ng, "geoms", sum(len(r["idx%d" % g]) // 3 for g in range(ng)), "triangles")
    np.savez_compressed(os.path.join(OUT, "rib_parse.npz"), **d)


def hdr_fixture():
    """Bytes the reference's "file" display driver (hdrdrv.c -> rgbe.c, run-length coded) writes for
    seeded float frames: inputs are reproducible from the recipe in tests/test_rib.py::hdr_frames."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_rib import hdr_frames
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "liblucille_ref.so"))
    R.hdr_dd_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
    R.hdr_dd_write.argtypes = [C.c_int, C.c_int, C.c_void_p]
    d = {}
    for name, img in hdr_frames():
        h, w, _ = img.shape
        path = os.path.join("/tmp", "golden_%s.hdr" % name)
        assert R.hdr_dd_open(path.encode(), w, h, 32, b"rgb", b"float") == 1
        for y in range(h):
            for x in range(w):
                px = np.ascontiguousarray(img[y, x]); R.hdr_dd_write(x, y, px.ctypes.data)
        R.hdr_dd_close()
        d[name] = np.frombuffer(open(path, "rb").read(), np.uint8)
        print("hdr", name, img.shape, len(d[name]), "bytes")
    np.savez_compressed(os.path.join(OUT, "hdr_bytes.npz"), **d)


if __name__ == "__main__":
    if "--rib" in sys.argv:
        rib_fixture(); hdr_fixture(); sys.exit(0)
    if "--state" in sys.argv:
        state_fixture(); sys.exit(0)
    if "--beam-raster" in sys.argv:
        beam_raster_fixture(); sys.exit(0)
    if not po.ref_available(stat=True):
        po.build_ref()
    soup_fixture("soup_20k", 20000, 20000, 0.005)
    soup_fixture("soup_3k_fat", 3000, 10000, 0.05)
    beam_fixture("beams_2k", 2000, 0.03, 77, 4000)
    beam_fixture("beams_300", 300, 0.1, 78, 4000)
    # BASELINE config 1: examples/ambient_occlusion.rib, 256x256, 16 AO samples, 1 thread
    ao_fixture("ao_c1", "/root/reference/examples/ambient_occlusion/ambient_occlusion.rib", 256, 256, 16)
    # examples/plane_sphere (BASELINE config 4's scene): 1 986 triangles, vertex normals (Ns is
    # interpolated), ReadArchive, lh orientation; small frame, 2x2 pixel samples
    ao_fixture("ao_ps", "/root/reference/examples/plane_sphere/Scene_DEFAULT_Set0.rib", 96, 96, 9, pixel_samples=2)
    rib_fixture()
    hdr_fixture()
    state_fixture()
    beam_raster_fixture()
    # check values of the full S-soup-1M are pinned in
    # tests/test_oracle_vs_ref.py against the live reference, not stored here.
