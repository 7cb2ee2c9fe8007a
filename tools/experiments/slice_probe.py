"""does the traversal kernel lose efficiency on smaller launches?  config 5 scene, the frame's 16.8 M camera rays and
the any-hit version of the same rays, traced in 1 / 4 / 16 / 64 / 256 slices (no host sync in between)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
tess = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I)
acc.commit(build="host")
c = g["camera"]; cam = la.Camera.make(4096, 4096, c[16], c[:16], int(c[19]))
org, dr = acc.primary_rays(cam, 0, 0, 4096, 4096, 1)
n = org.shape[0]
for mode in (la.MODE_CLOSEST, la.MODE_ANY):
    out = acc.intersect_device(org, dr, mode=mode); torch.cuda.synchronize()
    for parts in (1, 4, 16, 64, 256):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for p in range(parts):
                b, e = n * p // parts, n * (p + 1) // parts
                acc.intersect_device(org[b:e], dr[b:e], out=tuple(x[b:e] for x in out), mode=mode)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print("mode %d, %3d slices: %.3f ms (%.0f Mrays/s)" % (mode, parts, best * 1e3, n / best / 1e6), flush=True)
