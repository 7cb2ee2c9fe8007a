"""tri_batch x min_active around the defaults: S-soup-1M closest-hit dump, then the config-5 AO frame (python tools/knob_sweep2.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes, render
n = 100_000_000
P, idx, st = scenes.soup_triangles(1000000, 0.005)
ho, hd, _ = scenes.soup_rays(n, st)
o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda(); del ho, hd
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build="host")
out = acc.intersect_device(o, d); torch.cuda.synchronize()
def t():
    ts = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(); acc.intersect_device(o, d, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return n / min(ts) / 1e3
for tb in (8, 10, 12, 14):
    for ma in (32, 36, 40):
        acc.set_param("tri_batch", tb); acc.set_param("min_active", ma)
        print("soup tri_batch %2d min_active %2d  %.1f Mrays/s" % (tb, ma, t()), flush=True)
acc.close(); del o, d, out
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    Pk, Ik = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 8); acc.add_mesh(Pk, Ik)
acc.commit(build="host")
c = g["camera"]; cam = la.Camera.make(4096, 4096, c[16], c[:16], int(c[19]))
for tb in (8, 12):
    for ma in (32, 40):
        acc.set_param("tri_batch", tb); acc.set_param("min_active", ma)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); render.render_ao_frame(acc, cam, 1, 64, tile=4096); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print("AO   tri_batch %2d min_active %2d  %.2f ms" % (tb, ma, min(ts)), flush=True)
