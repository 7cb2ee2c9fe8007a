"""S-soup-1M closest-hit dump: one knob at a time around the defaults (set_param), 3 timed launches each, best of 3.
python tools/knob_sweep.py [nrays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
P, idx, st = scenes.soup_triangles(1000000, 0.005)
ho, hd, _ = scenes.soup_rays(n, st)
o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda(); del ho, hd
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build="host")
out = acc.intersect_device(o, d); torch.cuda.synchronize()
def t():
    ts = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(); acc.intersect_device(o, d, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return n / min(ts) / 1e3
DEF = {"min_active": 32, "tri_batch": 12, "ray_chunk": 256, "dump_budget": 2048}
print("defaults %.1f Mrays/s" % t(), flush=True)
for name, vals in (("dump_budget", (512, 1024, 4096, 16384, 1 << 30)), ("min_active", (16, 24, 40, 48)), ("tri_batch", (4, 6, 12, 16, 24)),
                   ("ray_chunk", (64, 128, 512, 1024))):
    for v in vals:
        acc.set_param(name, v)
        print("%-12s %10d  %.1f Mrays/s" % (name, v, t()), flush=True)
    acc.set_param(name, DEF[name])
for g in (768, 1024, 1280):
    acc.set_grid(g); print("grid %d  %.1f Mrays/s" % (g, t()), flush=True)
print("defaults again %.1f" % t())
