"""BASELINE config 5 probe: example scene tessellated 4^n, big AO frame on 1 GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes
tess = int(sys.argv[1]); size = int(sys.argv[2]); ns = int(sys.argv[3]); tile = int(sys.argv[4])
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "ao_c1.npz"))
t0 = time.time()
acc = la.HipAccel(0); ntri = 0
for k in range(int(g["ngeoms"])):
    P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I); ntri += len(I) // 3
t1 = time.time(); info = acc.commit(build="host"); t2 = time.time()
print("tris", ntri, "tessellate s %.2f commit s %.2f" % (t1 - t0, t2 - t1), info, flush=True)
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    img, st = render.render_ao_frame(acc, cam, 1, ns, tile=tile)
    torch.cuda.synchronize(); dt = time.time() - t0
    rays = st["primary_rays"] + st["ao_rays"]
    print("frame %d: %.1f ms, %d rays, %.1f Mrays/s, mean %.6f, mem %.1f GB" % (it, dt * 1e3, rays, rays / dt / 1e6, img.mean().item(), torch.cuda.max_memory_allocated() / 1e9), flush=True)
