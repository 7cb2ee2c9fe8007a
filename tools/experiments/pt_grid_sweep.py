"""round 5: the path-traced config-4 frame against the number of persistent workgroups (occupancy).   python tools/experiments/pt_grid_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_ps.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    acc.add_mesh(g["pos%d" % k], g["idx%d" % k])
    if ("nrm%d" % k) in g.files: acc.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
acc.commit()
c = g["camera"]; cam = la.Camera.make(2048, 2048, c[16], c[:16], int(c[19]))
def frame(mv):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img, st = render.render_pt_frame_sharded(acc, cam, 256, 0, 1, tile=2048, spp_chunk=256, kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=mv, seed=7)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for grid in (256, 512, 768, 1024):
    acc.set_param("grid", grid)
    frame(2); cam_only = min(frame(2) for _ in range(3)); frame(8); full = min(frame(8) for _ in range(3))
    print("grid %4d workgroups (%d per CU): camera rays + first decision %.2f ms, whole frame %.2f ms" % (grid, grid // 256, cam_only, full), flush=True)
