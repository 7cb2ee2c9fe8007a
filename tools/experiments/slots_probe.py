"""lane-slot diagnostics of the COUNT build: python tools/slots_probe.py [nrays]  (LH_MIN_ACTIVE / LH_TRI_BATCH sweepable)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["LH_DEBUG_COUNTERS"] = "1"
import numpy as np, torch
import lucille_amd as la
from oracle import pyoracle as po
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 8000000
P, idx, org, dr = po.soup(1000000, nr)
o = torch.from_numpy(org).cuda(); d = torch.from_numpy(dr).cuda()
for ma, tb, ch in ((32, 8, 64), (32, 8, 256), (32, 8, 512), (32, 8, 2048), (40, 8, 512), (48, 8, 512), (56, 8, 512), (40, 16, 512), (48, 4, 512)):
    os.environ["LH_MIN_ACTIVE"] = str(ma); os.environ["LH_TRI_BATCH"] = str(tb); os.environ["LH_RAY_CHUNK"] = str(ch)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build="host")
    for mode in (0,):
        print("min_active", ma, "tri_batch", tb, "chunk", ch, flush=True)
        out, cnt = acc.intersect_device(o, d, mode=mode, counters=True)
        torch.cuda.synchronize()
        outs = acc.intersect_device(o, d, mode=mode); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); acc.intersect_device(o, d, out=outs, mode=mode); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        print("   %.1f Mrays/s" % (nr / min(ts) / 1e3), flush=True)
    acc.close()
