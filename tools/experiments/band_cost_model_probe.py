"""Round 6: can a band's cost be read off the PREVIOUS frame?  Per band of the config-5 frame: camera-ray hits (pixels with geometry) and the sum of the
pixels' values (= unoccluded AO rays / N: an unoccluded ray walks to the end, an occluded one stops at its occluder); per rank (the product's band
rule) those sums against the rank's measured batch time; then a band's own time, each band as a batch of its own (launch overheads included).
python tools/experiments/band_cost_model_probe.py"""
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
brow, y0s = render.bands_for(size, W); nb = len(y0s)
img, st = render.render_ao_frame(acc, cam, 1, ns, tile=size); torch.cuda.synchronize()
v = img[:, :, 0].flip(0)                                   # frame line 0 = bottom: the band rule counts lines from there
# which pixels have geometry: a second frame with 1 AO sample would do; here: the primary-hit mask through the scratch of a tile render is not
# per pixel, so use value > 0 or ... (a fully occluded hit reads 0 like a miss: rare) -- and the frame's own statistics for the total
vis = v.view(nb, brow * size).sum(1).double().cpu().numpy()            # sum of values per band = unoccluded rays / N
geo = (v > 0).view(nb, brow * size).sum(1).double().cpu().numpy()      # pixels that see geometry and some light
batches = [shard.bands_of_rank(nb, r, W) for r in range(W)]
slab = torch.zeros((len(batches[0]), brow, size, 3), dtype=torch.float32, device="cuda")
tr = []
for r in range(W):
    mine = [y0s[b] for b in batches[r]]
    acc.render_ao_bands(cam, mine, brow, 1, ns, seed=1, out=slab); torch.cuda.synchronize(); ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); acc.render_ao_bands(cam, mine, brow, 1, ns, seed=1, out=slab); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    tr.append(min(ts))
tr = np.array(tr); G = np.array([geo[b].sum() for b in batches]); V = np.array([vis[b].sum() for b in batches])
print("rank   batch ms   lit pixels (k)   sum of values (k)")
for r in range(W): print("%4d   %7.2f   %10.0f   %10.0f" % (r, tr[r], G[r] / 1e3, V[r] / 1e3))
A = np.stack([np.ones(W), G, V], 1); coef, res, *_ = np.linalg.lstsq(A, tr, rcond=None)
print("least squares  t = %.3f + %.3e x lit + %.3e x values: residuals (ms) %s" % (coef[0], coef[1], coef[2], " ".join("%+.3f" % x for x in (tr - A @ coef))))
print("correlation of the batch time with lit pixels %.3f, with the sum of values %.3f" % (np.corrcoef(tr, G)[0, 1], np.corrcoef(tr, V)[0, 1]))
# a band by itself: groups of 8 adjacent bands (128 lines) as one batch each -- coarse, but the launch overhead stays a small part
tg = []
for gp in range(nb // W):
    mine = [y0s[gp * W + k] for k in range(W)]
    out = slab[:W]
    acc.render_ao_bands(cam, mine, brow, 1, ns, seed=1, out=out); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); acc.render_ao_bands(cam, mine, brow, 1, ns, seed=1, out=out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    tg.append(min(ts))
tg = np.array(tg); Gg = geo.reshape(-1, W).sum(1); Vg = vis.reshape(-1, W).sum(1)
A = np.stack([np.ones(len(tg)), Gg, Vg], 1); coef, *_ = np.linalg.lstsq(A, tg, rcond=None)
print("groups of 128 lines as batches of their own (ms): " + " ".join("%.2f" % x for x in tg))
print("least squares  t = %.3f + %.3e x lit + %.3e x values; correlation of the residual-free fit %.3f; largest residual %.3f ms of a mean of %.2f" % (coef[0], coef[1], coef[2], np.corrcoef(tg, A @ coef)[0, 1], np.abs(tg - A @ coef).max(), tg.mean()))
