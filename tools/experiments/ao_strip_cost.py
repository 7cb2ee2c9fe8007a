"""round 4: what the AO rays of each 128-line strip of the config-5 frame cost (counting kernels): where the slow ranges of the
fused AO launch come from.  python tools/experiments/ao_strip_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
size, tess, ns, rows = 4096, 8, 64, 128
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I)
acc.commit(build="device")
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
acc.trace_statistics(True)
for y0 in range(0, size, rows):
    acc.statistics(clear=True)
    img, st = acc.render_ao_tile(cam, 0, y0, size, rows, 1, ns, seed=1); torch.cuda.synchronize()
    cs = acc.statistics(clear=True)
    nr = max(1, cs["rays"])
    print("lines %4d-%4d  hits %7d  rays %9d  nodes/ray %6.2f  tris/ray %5.2f  retraced %d" % (y0, y0 + rows - 1, st["primary_hits"], cs["rays"], cs["nodes"] / nr, cs["tris"] / nr, acc.last_retraced() if hasattr(acc, "last_retraced") else -1), flush=True)
