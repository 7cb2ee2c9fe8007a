import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
c = g["camera"]; cam = la.Camera.make(4096, 4096, c[16], c[:16], int(c[19]))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    Pk, Ik = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 8); acc.add_mesh(Pk, Ik)
t0 = time.perf_counter(); info = acc.commit(on_device=True); tc = time.perf_counter() - t0
acc.wait_exact()
render.render_ao_frame(acc, cam, 1, 64, tile=4096); torch.cuda.synchronize(); best = 1e9
for _ in range(2):
    t0 = time.perf_counter(); img, st = render.render_ao_frame(acc, cam, 1, 64, tile=4096); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print("LH_DEVICE_LEAF=%s: commit %.2f s, %d nodes depth %d, AO frame %.1f ms mean %.6f" % (os.environ.get("LH_DEVICE_LEAF", "1"), tc, info["nnodes_traversal"], info["max_depth"], best * 1e3, float(img.mean())))
