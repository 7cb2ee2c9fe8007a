"""BASELINE config 4's frame, nothing else: python tools/pt_frames.py [size] [spp] [spp_chunk] [frames]
(for rocprofv3 --kernel-trace --stats: every kernel total / frames = its share of one frame; the first frame is warm-up
but counted by the profiler, so use frames >= 4 and the printed wall time of the later ones)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 64
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 4
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "ao_ps.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    acc.add_mesh(g["pos%d" % k], g["idx%d" % k])
    if ("nrm%d" % k) in g.files:
        acc.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
acc.commit(build="host")
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
ts = []
for it in range(frames):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img, st = render.render_pt_frame_sharded(acc, cam, spp, 0, 1, tile=size, spp_chunk=chunk, kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=8, seed=7)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("frames_ms", ["%.2f" % t for t in ts], "rays", st["rays"], "paths", st["paths"], "mean", float(img.mean().item()))
