"""config 5 frame as one batch, then as full-width bands of `rows` rows (for rocprofv3 --kernel-trace):
python tools/band_probe.py [rows ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
rows_list = [int(x) for x in sys.argv[1:]] or [4096, 1024, 256, 64]
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 8); acc.add_mesh(P, I)
acc.commit(build="host")
c = g["camera"]; cam = la.Camera.make(4096, 4096, c[16], c[:16], int(c[19]))
out = torch.empty((4096, 4096, 3), dtype=torch.float32, device="cuda")
acc.render_ao_tile(cam, 0, 0, 4096, 4096, 1, 64, seed=1, out=out); torch.cuda.synchronize()
for rows in rows_list:
    torch.cuda.synchronize(); t0 = time.perf_counter(); rays = 0
    for y0 in range(0, 4096, rows):
        _, st = acc.render_ao_tile(cam, 0, y0, 4096, rows, 1, 64, seed=1, out=out[:rows]); rays += st["primary_rays"] + st["ao_rays"]
    torch.cuda.synchronize()
    print("bands of %4d rows: frame %.2f ms, %d rays" % (rows, (time.perf_counter() - t0) * 1e3, rays), flush=True)
