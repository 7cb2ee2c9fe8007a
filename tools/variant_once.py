"""one closest-hit launch of a kernel variant over S-soup-1M (for counter passes): python tools/variant_once.py <variant> [nrays] [param=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
v = int(sys.argv[1]); nr = int(sys.argv[2]) if len(sys.argv) > 2 else 20000000
P, idx, st = scenes.soup_triangles(1000000, 0.005)
ho, hd, _ = scenes.soup_rays(nr, st)
o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda()
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build="host")
for kv in sys.argv[3:]:
    k, val = kv.split("="); acc.set_param(k, int(val))
out = acc.intersect_device(o[:1000000], d[:1000000], variant=v); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); acc.intersect_device(o, d, variant=v); e1.record(); torch.cuda.synchronize()
print("variant %d %s: %.0f Mrays/s" % (v, " ".join(sys.argv[3:]), nr / e0.elapsed_time(e1) / 1e3))
