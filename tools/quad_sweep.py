"""the quad-per-ray walk (variant 7) against the default on S-soup-1M, with its regroup / triangle-batch thresholds swept:
python tools/quad_sweep.py [nrays]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 20000000
P, idx, st = scenes.soup_triangles(1000000, 0.005)
ho, hd, _ = scenes.soup_rays(nr, st)
o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda()
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
def rate(mode, v):
    out = acc.intersect_device(o, d, mode=mode, variant=v); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); acc.intersect_device(o, d, out=out, mode=mode, variant=v); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return nr / best / 1e3
print("default walk (variant 4): closest %.0f any %.0f Mrays/s" % (rate(0, 4), rate(1, 4)), flush=True)
print("quad walk, defaults: closest %.0f any %.0f" % (rate(0, 7), rate(1, 7)), flush=True)
for ma in (4, 16, 32, 48):
    acc.set_param("min_active", ma)
    for tb in (4, 8, 16, 32):
        acc.set_param("tri_batch", tb)
        print("quad walk min_active %2d (rays %2d) tri_batch %2d (rays %d): closest %.0f any %.0f" % (ma, max(1, ma // 4), tb, max(1, tb // 4), rate(0, 7), rate(1, 7)), flush=True)
acc.set_param("min_active", 32); acc.set_param("tri_batch", 8)
for g in (256 * 2, 256 * 3, 256 * 4, 256 * 6, 256 * 8):
    acc.set_param("quad_grid", g)
    print("quad walk grid %4d workgroups: closest %.0f any %.0f" % (g, rate(0, 7), rate(1, 7)), flush=True)
