"""S-soup-1M closest-hit dump on both builders' trees at several persistent-grid sizes: python tools/grid_probe.py [nrays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
P, idx, st = scenes.soup_triangles(1000000, 0.005)
ho, hd, _ = scenes.soup_rays(n, st)
o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda(); del ho, hd
for build in ("host", "device"):
    acc = la.HipAccel(0); acc.add_mesh(P, idx); info = acc.commit(build=build)
    out = acc.intersect_device(o, d); torch.cuda.synchronize()
    def t():
        ts = []
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(); acc.intersect_device(o, d, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        return n / min(ts) / 1e3
    print("%s tree (depth %d): default grid %.1f Mrays/s" % (build, info["max_depth"], t()), flush=True)
    for g in (512, 768, 1024, 1280):
        acc.set_grid(g); print("   grid %4d  %.1f Mrays/s" % (g, t()), flush=True)
    acc.close()
