"""The host mirror of lucille's plugin API (include/lucille_accel.h, lh_host.c) driven by a plain
C program written the way the reference's testbed drives its accelerator
(src/testbed/main.cpp:53-65): ri_geom_* -> ri_scene_add_geom -> ri_accel_bind -> ri_scene_build_accel
-> ri_raytrace / ri_raytrace_batch, beams through ri_beam_set + ri_hipbvh_intersect_beam_visibility,
statistics through ri_hipbvh_*_stat_traversal.  Every record is compared with the oracle."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import ROOT, CSRC, random_beams

PROG_SRC = os.path.join(ROOT, "tests", "c", "host_api_prog.c")


def build_prog(tmp_path):
    exe = str(tmp_path / "host_api_prog")
    subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), PROG_SRC,
                           "-o", exe, "-L" + CSRC, "-llucille_hip", "-Wl,-rpath," + CSRC])
    return exe


def test_c_program_compiles_and_links_against_the_header(tmp_path):
    """no GPU needed: the mirror header is valid C99 and every function it declares is exported"""
    import __graft_entry__ as g
    g.build()
    build_prog(tmp_path)


def _vec4(a):
    out = np.zeros((len(a), 4)); out[:, :3] = a
    return out


@pytest.mark.gpu
def test_lucille_style_c_program_matches_the_oracle(tmp_path):
    exe = build_prog(tmp_path)
    rng = np.random.default_rng(31)
    # three geoms (prim ids run through the geom list, bvh.c:1792-1821); the second carries normals
    meshes = []
    for m, (ntri, he) in enumerate(((400, 0.05), (150, 0.08), (1, 0.3))):
        P, idx, _, _ = po.soup(ntri, 1, he, 100 + m)
        N = None
        if m == 1:
            N = rng.normal(size=P.shape); N /= np.linalg.norm(N, axis=1, keepdims=True)
        meshes.append((P, idx, N))
    nrays = 3000
    org = rng.uniform(-0.2, 1.2, (nrays, 3)); tgt = rng.uniform(0, 1, (nrays, 3)); dr = tgt - org
    dr[::7] *= 3.0                                               # unnormalised directions
    bb = [random_beams(rng, 100, sp) for sp in (0.0005, 0.01, 0.2)]
    borg = np.concatenate([b[0] for b in bb]); bdir = np.concatenate([b[1] for b in bb])

    o = po.Oracle()
    for i, (P, idx, N) in enumerate(meshes):
        o.add_mesh(P, idx)
        if N is not None:
            o.set_normals(i, N, 0)
    o.build()
    prim, t, u, v = o.intersect(org, dr)
    first = np.cumsum([0] + [len(idx) // 3 for _, idx, _ in meshes])

    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<I", len(meshes)))
        for P, idx, N in meshes:
            f.write(struct.pack("<I", len(P))); f.write(_vec4(P).tobytes())
            f.write(struct.pack("<I", len(idx))); f.write(np.ascontiguousarray(idx, np.uint32).tobytes())
            f.write(struct.pack("<I", 0 if N is None else 1))
            if N is not None:
                f.write(_vec4(N).tobytes())
        f.write(struct.pack("<I", nrays)); f.write(org.tobytes()); f.write(np.ascontiguousarray(dr).tobytes())
        f.write(struct.pack("<I", len(borg))); f.write(borg.tobytes()); f.write(np.ascontiguousarray(bdir).tobytes())
    res = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    assert "BVH traversal statistiscs" in res.stdout and "# of rays                    %d" % (2 * nrays) in res.stdout

    rec = np.dtype([("hit", "<i4"), ("geom", "<i4"), ("index", "<u4"), ("d", "<f8", 12), ("inside", "<i4")])
    raw = open(fout, "rb").read()
    recs = np.frombuffer(raw, rec, 2 * nrays)
    off = 2 * nrays * rec.itemsize
    beams = np.frombuffer(raw, "<i4", 2 * len(borg), off).reshape(-1, 2); off += 8 * len(borg)
    stat = np.frombuffer(raw, "<u8", 5, off); off += 40
    rc_unknown, empty_hit = np.frombuffer(raw, "<i4", 2, off); off += 8
    tile_c = np.frombuffer(raw, "<f4", 24 * 16 * 3, off).reshape(16, 24, 3); off += 4 * 24 * 16 * 3
    nraster = int(np.frombuffer(raw, "<u4", 1, off)[0]); off += 4
    raster = []
    for _ in range(nraster):
        ordn = int(np.frombuffer(raw, "<i4", 1, off)[0]); off += 4
        raster.append((ordn, np.frombuffer(raw, "<f8", 256, off).reshape(16, 16))); off += 8 * 256
    assert off == len(raw)

    assert rc_unknown == -1 and empty_hit == 0
    hit = prim != po.MISS
    L = po.lib(); dp = C.POINTER(C.c_double)
    for leg, part in (("ri_raytrace", recs[:nrays]), ("ri_raytrace_batch", recs[nrays:])):
        assert np.array_equal(part["hit"] != 0, hit), leg
        h = np.nonzero(hit)[0]
        mesh = np.searchsorted(first, prim[h], side="right") - 1
        assert np.array_equal(part["geom"][h], mesh), leg
        assert np.array_equal(part["index"][h], 3 * (prim[h] - first[mesh])), leg        # bvh.c:1813
        assert np.array_equal(part["d"][h, 0], t[h]) and np.array_equal(part["d"][h, 1], u[h]) and np.array_equal(part["d"][h, 2], v[h]), leg
        for i in h[:800]:                                            # hit epilogue vs the oracle's restatement
            Pp = np.zeros(3); Ng = np.zeros(3); Ns = np.zeros(3); inside = C.c_int(0)
            oo = np.ascontiguousarray(org[i]); dd = np.ascontiguousarray(dr[i])
            L.lo_state_build(o.h, int(prim[i]), float(t[i]), float(u[i]), float(v[i]), oo.ctypes.data_as(dp), dd.ctypes.data_as(dp),
                             Pp.ctypes.data_as(dp), Ng.ctypes.data_as(dp), Ns.ctypes.data_as(dp), C.byref(inside))
            assert np.array_equal(part["d"][i, 3:6], Pp) and np.array_equal(part["d"][i, 6:9], Ng), leg
            assert np.array_equal(part["d"][i, 9:12], Ns) and part["inside"][i] == inside.value, leg

    exp_b = o.beam_visibility(borg, bdir)
    assert np.array_equal(beams[:, 0], np.where(exp_b < 0, -1, 0))
    ok = exp_b >= 0
    assert np.array_equal(beams[ok, 1], exp_b[ok])
    assert ok.sum() > 50 and len(set(exp_b[ok].tolist())) >= 2

    # the beam-raster path through the mirror (ri_raster_plane_setup + ri_hipbvh_intersect_beam): plane->t == the oracle's
    assert nraster == 8
    written = 0
    for ordn, plane_t in raster:
        rc, t_exp, fl = o.beam_raster(borg[ordn], bdir[ordn], 16, 16, np.eye(3), bdir[ordn][0], borg[ordn], 45.0)
        assert rc == 0 and np.array_equal(plane_t, t_exp)
        written += int((plane_t != 0).sum())
    assert written > 0

    # statistics: both legs counted, hits agree with the oracle; work counters are plausible
    assert stat[0] == 2 * nrays and stat[4] == 2 * int(hit.sum())
    assert stat[1] > 0 and stat[2] >= stat[4] and stat[3] >= stat[4] // 2

    # the tile-level entry point of the plain-C mirror == the same tile through the ctypes plumbing
    import lucille_amd as la
    acc = la.HipAccel(0)
    for i, (P, idx, N) in enumerate(meshes):
        acc.add_mesh(P, idx)
        if N is not None:
            acc.set_normals(i, N, 0)
    acc.commit()
    c2w = np.eye(4); c2w[3, :3] = (0.5, 0.5, 3.0)
    cam = la.Camera.make(32, 32, 2.0, c2w.reshape(16), 1)
    img, st = acc.render_ao_tile(cam, 4, 8, 24, 16, 2, 16, seed=77)
    assert np.array_equal(img.cpu().numpy(), tile_c)
    assert st["primary_hits"] > 50 and 0.0 < float(tile_c.mean()) < 1.0
    acc.close()
