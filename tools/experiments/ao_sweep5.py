"""config-5 AO frame against the kernel's regroup / triangle-batch / range thresholds (one scene build): python tools/ao_sweep5.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
c = g["camera"]; cam = la.Camera.make(4096, 4096, c[16], c[:16], int(c[19]))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    Pk, Ik = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 8); acc.add_mesh(Pk, Ik)
acc.commit(build="host")
def frame():
    render.render_ao_frame(acc, cam, 1, 64, tile=4096); torch.cuda.synchronize(); best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); img, st = render.render_ao_frame(acc, cam, 1, 64, tile=4096); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, float(img.mean())
print("defaults (min_active 32, tri_batch 12): %.2f ms mean %.6f" % frame(), flush=True)
for ma in (24, 32, 40):
    for tb in (8, 12, 16, 24):
        acc.set_param("min_active", ma); acc.set_param("tri_batch", tb)
        print("min_active %2d tri_batch %2d: %.2f ms" % ((ma, tb) + frame()[:1]), flush=True)
acc.set_param("min_active", 32); acc.set_param("tri_batch", 12)
for rc in (64, 128, 256, 512, 1024):
    acc.set_param("ray_chunk", rc)
    print("ray_chunk %4d: %.1f ms" % ((rc,) + frame()[:1]), flush=True)
acc.set_param("ray_chunk", 256)
for gb in (512, 768, 1024, 1280):
    acc.set_param("grid", gb)
    print("grid %4d workgroups: %.1f ms" % ((gb,) + frame()[:1]), flush=True)
