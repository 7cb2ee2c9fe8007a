"""round 4: visit budget of the fused AO stage at four workgroups per CU (config 5): the whole frame and rank 0's share of an 8-rank
frame.  python tools/ao_budget_probe.py"""
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
ncu = torch.cuda.get_device_properties(0).multi_processor_count
per_cu = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if per_cu: acc.set_param("grid", ncu * per_cu); acc.set_param("stack_cap", cap)
for world in (1, 2, 4):
    brow, y0s = render.bands_for(size, world, None)
    mine = [y0s[b] for b in shard.bands_of_rank(len(y0s), 0, world)]
    out = torch.zeros((len(mine), brow, size, 3), dtype=torch.float32, device="cuda")
    for rb, ab in ((128, 384), (128, 512), (128, 768), (128, 1024), (128, 384)):
        acc.set_param("ray_budget", rb); acc.set_param("ao_budget", ab)   # "ray_budget" sets both: the AO stage's after it
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            acc.render_ao_bands(cam, mine, brow, 1, ns, seed=1, out=out); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("world %d rank 0  camera-ray budget %4d  AO budget %4d  batch ms: best %.2f  median %.2f" % (world, rb, ab or rb, min(ts), sorted(ts)[2]), flush=True)
