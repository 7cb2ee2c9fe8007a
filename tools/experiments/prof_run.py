"""One small traced workload for PMC passes: S-soup-1M, N rays, variant 2 closest + any."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lucille_amd as la
from oracle import pyoracle as po
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 20000000
P, idx, org, dr = po.soup(int(os.environ.get("NTRI", "1000000")), nt)
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build="host")
o = torch.from_numpy(org).cuda(); d = torch.from_numpy(dr).cuda()
for mode in (0, 1):
    out = acc.intersect_device(o, d, mode=mode)
    torch.cuda.synchronize()
    out = acc.intersect_device(o, d, out=out, mode=mode)
    torch.cuda.synchronize()
acc.close()
