"""round 4: the order of the fused AO stage's work items (set_param "ao_group"): config-5 frame (21.1 M triangles, 4096^2, 64 AO
samples) and the config-2 frame, every setting's frame compared with the plain order's bit for bit.  python tools/ao_group_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes, render
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ao_c1.npz"))
c = g["camera"]
for tess, size in ((8, 4096), (0, 1024)):
    acc = la.HipAccel(0)
    for k in range(int(g["ngeoms"])):
        Pk, Ik = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(Pk, Ik)
    acc.commit()
    cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
    base = None
    for grp in (0, 64, 16, 256, 1024, 64, 0):
        acc.set_param("ao_group", grp)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fr, st = render.render_ao_frame(acc, cam, 1, 64, tile=size); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        if base is None: base = fr.clone()
        print("tess %d %dx%d  ao_group %4d  %.2f ms  frame %s" % (tess, size, size, grp, min(ts), "equal" if torch.equal(fr, base) else "DIFFERS"), flush=True)
    acc.close()
