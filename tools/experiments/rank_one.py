"""One rank's share (world, rank, band rows from argv) of the BASELINE config-5 AO frame, a few times: the command to put under
rocprofv3 --kernel-trace --stats / --pmc when looking at what a rank's batch costs per kernel.
  python tools/rank_one.py [world] [rank] [band_rows] [repeats] [size] [tess] [samples]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes, shard
a = [int(x) for x in sys.argv[1:]] + [None] * 7
world, rank, brow, reps, size, tess, ns = a[0] or 8, a[1] or 0, a[2] or 4, a[3] or 5, a[4] or 4096, a[5] or 8, a[6] or 64
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I)
acc.commit(build="host")
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
if world == 1:
    brow = size
y0s = list(range(0, size, brow))
mine = [y0s[b] for b in shard.tiles_of_rank(len(y0s), rank, world)]
out = torch.zeros((len(mine), brow, size, 3), dtype=torch.float32, device="cuda")
for r in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    acc.render_ao_bands(cam, mine, brow, 1, ns, seed=1, out=out); torch.cuda.synchronize()
    print("world %d rank %d rows %d: %.2f ms" % (world, rank, brow, (time.perf_counter() - t0) * 1e3), flush=True)
