"""Regroup / triangle-pass thresholds of the two frames, finer: the path-traced config-4 frame and the config-5 AO frame, each pair timed twice in two passes.   python tools/experiments/knob_sweep6.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_ps.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    acc.add_mesh(g["pos%d" % k], g["idx%d" % k])
    if ("nrm%d" % k) in g.files: acc.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
acc.commit()
c = g["camera"]; cam = la.Camera.make(2048, 2048, c[16], c[:16], int(c[19]))
for rep in range(2):
    row = []
    for ma, tb in ((32, 12), (24, 8), (24, 12), (20, 8), (28, 8), (24, 6), (16, 8), (16, 6), (24, 4), (32, 12)):
        acc.set_param("min_active", ma); acc.set_param("tri_batch", tb); ts = []
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            img, st = render.render_pt_frame_sharded(acc, cam, 256, 0, 1, tile=2048, spp_chunk=256, kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=8, seed=7)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        row.append("(%d, %d) %.2f" % (ma, tb, min(ts[1:])))
    print("config-4 frame, ms: " + "  ".join(row), flush=True)
acc.close(); del img
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P_, I_ = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 8); acc.add_mesh(P_, I_); del P_, I_
acc.commit()
c = g["camera"]; cam = la.Camera.make(4096, 4096, c[16], c[:16], int(c[19]))
for rep in range(2):
    row = []
    for ma, tb in ((32, 12), (24, 12), (28, 12), (24, 16), (28, 16), (20, 12), (32, 16), (24, 10), (32, 12)):
        acc.set_param("min_active", ma); acc.set_param("tri_batch", tb); ts = []
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            img, st = render.render_ao_frame(acc, cam, 1, 64, tile=4096)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        row.append("(%d, %d) %.2f" % (ma, tb, min(ts[1:])))
    print("config-5 AO frame, ms: " + "  ".join(row), flush=True)
