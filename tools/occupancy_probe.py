"""round 4: workgroups per CU of the incoherent walk, properly (the persistent grid follows): S-soup-1M closest-hit dump of 100 M rays,
grid = CUs x {2, 3, 4}, LDS stack rows capped so that many fit (4 per CU: 128 VGPRs allow no more), records compared with the default's.
python tools/occupancy_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
P, idx, st = scenes.soup_triangles(1000000, 0.005)
ho, hd, _ = scenes.soup_rays(n, st)
o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda(); del ho, hd
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build="host")
ref = [x.clone() for x in acc.intersect_device(o, d)]; torch.cuda.synchronize()
out = acc.intersect_device(o, d); torch.cuda.synchronize()
ncu = torch.cuda.get_device_properties(0).multi_processor_count
for per_cu, cap in ((3, 0), (4, 40), (4, 36), (4, 34), (4, 32), (4, 28), (5, 28), (3, 0)):
    acc.set_param("grid", ncu * per_cu); acc.set_param("stack_cap", cap); acc.set_param("top_nodes", 0)
    ts = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(); acc.intersect_device(o, d, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    same = all(torch.equal(a, b) for a, b in zip(out, ref))
    print("workgroups per CU %d  stack_cap %2d  %.1f Mrays/s  records %s" % (per_cu, cap, n / min(ts) / 1e3, "equal" if same else "DIFFER"), flush=True)
