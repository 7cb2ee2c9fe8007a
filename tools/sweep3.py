import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lucille_amd as la
from oracle import pyoracle as po
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 20000000
vals = [int(x) for x in sys.argv[2].split(",")]
P, idx, org, dr = po.soup(1000000, nt)
d_org = torch.from_numpy(org).cuda(); d_dir = torch.from_numpy(dr).cuda()
def timeit(acc, mode, variant, reps=3):
    outs = acc.intersect_device(d_org, d_dir, mode=mode, variant=variant); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); acc.intersect_device(d_org, d_dir, out=outs, mode=mode, variant=variant); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return nt / min(ts) / 1e3
for m in vals:
    os.environ["LH_MIN_ACTIVE"] = str(m)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
    print("min_active", m, "v3 closest %.1f any %.1f" % (timeit(acc, 0, 3), timeit(acc, 1, 3)), flush=True)
    acc.close()
