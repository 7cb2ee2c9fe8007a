"""Sweep of the lean walk's knobs on S-soup (GPU box): python tools/t2_sweep.py [nrays] [ntris] [half_extent]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lucille_amd as la
from lucille_amd import scenes
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 50000000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
he = float(sys.argv[3]) if len(sys.argv) > 3 else 0.005
P, idx, st = scenes.soup_triangles(nt, he)
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
o = torch.empty((nr, 3), dtype=torch.float64, device="cuda"); d = torch.empty_like(o)
for b in range(0, nr, 10000000):
    m = min(10000000, nr - b)
    ho, hd, st = scenes.soup_rays(m, st)
    o[b:b + m].copy_(torch.from_numpy(ho)); d[b:b + m].copy_(torch.from_numpy(hd))
cus = torch.cuda.get_device_properties(0).multi_processor_count
def t(mode, v):
    out = acc.intersect_device(o, d, mode=mode, variant=v); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); acc.intersect_device(o, d, out=out, mode=mode, variant=v); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return nr / min(ts) / 1e3
print("variant 4: closest %.0f any %.0f" % (t(0, 4), t(1, 4)), flush=True)
for wg in (2, 3, 4, 5, 6):
    acc.set_param("t2_grid", cus * wg)
    print("lean, %d workgroups/CU: closest %.0f any %.0f" % (wg, t(0, 6), t(1, 6)), flush=True)
acc.set_param("t2_grid", cus * 5)
for ma in (16, 24, 32, 40, 48):
    acc.set_param("min_active", ma)
    print("lean, min_active %d: closest %.0f any %.0f" % (ma, t(0, 6), t(1, 6)), flush=True)
acc.set_param("min_active", 32)
for tb in (4, 8, 12, 16, 24):
    acc.set_param("tri_batch", tb)
    print("lean, tri_batch %d: closest %.0f any %.0f" % (tb, t(0, 6), t(1, 6)), flush=True)
