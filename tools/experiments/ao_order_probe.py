"""round 4: the fused AO stage's work order "lowest stratum first" (set_param("ao_group", -1)): whole config-5 frame and shares of it,
frames compared bit for bit.  python tools/ao_order_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes, shard
size, tess, ns = 4096, 8, 64
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I)
acc.commit(build="device")
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
for world in (1, 2, 8, 32):
    brow, y0s = render.bands_for(size, world, None)
    mine = [y0s[b] for b in shard.tiles_of_rank(len(y0s), 0, world)]
    out = torch.zeros((len(mine), brow, size, 3), dtype=torch.float32, device="cuda")
    base = None
    for grp, ab in ((0, 384), (-1, 384), (-1, 1024), (-1, 4096), (0, 384), (-1, 2048)):
        acc.set_param("ao_group", grp); acc.set_param("ao_budget", ab)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            acc.render_ao_bands(cam, mine, brow, 1, ns, seed=1, out=out); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        if base is None: base = out.clone()
        print("world %2d rank 0  order %2d  AO budget %4d  batch ms: best %.2f  median %.2f  frame %s" % (world, grp, ab, min(ts), sorted(ts)[2], "equal" if torch.equal(out, base) else "DIFFERS"), flush=True)
