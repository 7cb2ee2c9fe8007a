"""one path-traced tile, three times (for a rocprofv3 --kernel-trace timeline): python tools/pt_tile_trace.py x0 y0 w h [spp] [size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lucille_amd as la
x0, y0, w, h = [int(v) for v in sys.argv[1:5]]
spp = int(sys.argv[5]) if len(sys.argv) > 5 else 256
size = int(sys.argv[6]) if len(sys.argv) > 6 else 2048
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "ao_ps.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    acc.add_mesh(g["pos%d" % k], g["idx%d" % k])
    if ("nrm%d" % k) in g.files:
        acc.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
acc.commit(build="host")
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
out = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, st = acc.render_pt_tile(cam, x0, y0, w, h, 0, spp, spp, kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=8, seed=7, out=out)
    torch.cuda.synchronize(); print("tile ms %.3f" % ((time.perf_counter() - t0) * 1e3), st)
