"""round 4: the regroup threshold (min_active) and the triangle-pass threshold (tri_batch) of the fused AO stage, now that the stage is
VALU-bound (four workgroups per CU): config-5 frame.  python tools/ao_knob_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes
size, tess, ns = 4096, 8, 64
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I)
acc.commit(build="device")
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
base = None
for ma, tb in ((32, 12), (48, 12), (40, 12), (24, 12), (16, 12), (8, 12), (32, 8), (32, 20), (32, 28), (32, 40), (24, 24), (16, 24), (16, 40), (32, 12)):
    acc.set_param("min_active", ma); acc.set_param("tri_batch", tb)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fr, st = render.render_ao_frame(acc, cam, 1, ns, tile=size); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    if base is None: base = fr.clone()
    print("min_active %2d tri_batch %2d  %.2f ms  frame %s" % (ma, tb, min(ts), "equal" if torch.equal(fr, base) else "DIFFERS"), flush=True)
