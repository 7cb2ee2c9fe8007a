"""In-process A/B of kernel variants on S-soup: python tools/variant_ab.py [nrays] [variants...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lucille_amd as la
from oracle import pyoracle as po
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 50000000
vs = [int(x) for x in sys.argv[2:]] or [4, 5]
P, idx, org, dr = po.soup(1000000, nr)
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
o = torch.from_numpy(org).cuda(); d = torch.from_numpy(dr).cuda()
def t(mode, v):
    out = acc.intersect_device(o, d, mode=mode, variant=v); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); acc.intersect_device(o, d, out=out, mode=mode, variant=v); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return nr / min(ts) / 1e3
for rnd in range(2):
    for v in vs:
        print("variant", v, "closest %.1f any %.1f" % (t(0, v), t(1, v)), flush=True)
