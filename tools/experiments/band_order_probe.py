"""Round 6: is the spread between the ranks' shares (7.2 .. 7.7 ms at equal hit counts) the bands they hold, or the ORDER they are measured in?
The product's band rule, every rank's batch alone, best of 4: ranks 0..7, ranks 7..0, ranks 0..7 with half a second of rest before each.
python tools/experiments/band_order_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes, shard
size, tess, ns, W = 4096, 8, 64, 8
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I); del P, I
acc.commit()
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
brow, y0s = render.bands_for(size, W)
batches = [[y0s[b] for b in shard.bands_of_rank(len(y0s), r, W)] for r in range(W)]
slab = torch.zeros((len(batches[0]), brow, size, 3), dtype=torch.float32, device="cuda")
def t_of(r, reps=4):
    acc.render_ao_bands(cam, batches[r], brow, 1, ns, seed=1, out=slab); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); acc.render_ao_bands(cam, batches[r], brow, 1, ns, seed=1, out=slab); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), float(np.median(ts))
for r in range(W): t_of(r, 1)
for name, order, rest in (("ranks 0..7", list(range(W)), 0.0), ("ranks 7..0", list(range(W - 1, -1, -1)), 0.0), ("ranks 0..7, 0.5 s of rest before each", list(range(W)), 0.5),
                          ("ranks 7..0, 0.5 s of rest before each", list(range(W - 1, -1, -1)), 0.5), ("ranks 0..7 again", list(range(W)), 0.0)):
    res = {}
    for r in order:
        if rest: time.sleep(rest)
        res[r] = t_of(r)
    print("%-40s best of 4 by rank: %s | median: %s" % (name, " ".join("%.2f" % res[r][0] for r in range(W)), " ".join("%.2f" % res[r][1] for r in range(W))), flush=True)
