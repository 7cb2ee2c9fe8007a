import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
size, spp = 2048, 256
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_ps.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    acc.add_mesh(g["pos%d" % k], g["idx%d" % k])
    if ("nrm%d" % k) in g.files:
        acc.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
acc.commit()
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
acc.set_param("pt_fused", 1)
W = int(sys.argv[1])
for grid in (256 * W, 256 * (W + 1)):
    acc.set_param("pt_grid", grid)
    best = 1e9
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        img, st = render.render_pt_frame_sharded(acc, cam, spp, 0, 1, tile=size, spp_chunk=256, kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=8, seed=7)
        torch.cuda.synchronize()
        if it: best = min(best, time.perf_counter() - t0)
    print("waves %d grid %d: %.1f ms mean %.9f rays %d" % (W, grid, best * 1e3, float(img.mean()), st["rays"]), flush=True)
