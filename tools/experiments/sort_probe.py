"""would sorting an incoherent ray dump pay?  S-soup-1M, n rays: trace time in dump order against the same rays ordered by
(origin cell Morton code, direction octant) with torch: python tools/sort_probe.py [nrays] [bits]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 8
P, idx, st = scenes.soup_triangles(1000000, 0.005)
ho, hd, _ = scenes.soup_rays(n, st)
o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda()
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build="host")
def t(o, d):
    out = acc.intersect_device(o, d); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(); acc.intersect_device(o, d, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts), out
t0, out0 = t(o, d)
print("dump order: %.2f ms = %.0f Mrays/s" % (t0, n / t0 / 1e3))
def spread(x):
    x = x.to(torch.int64); r = torch.zeros_like(x)
    for b in range(bits):
        r |= ((x >> b) & 1) << (3 * b)
    return r
for mode in ("origin", "origin+octant", "entry"):
    lo = o.min(0).values; hi = o.max(0).values
    if mode == "entry":          # where the ray enters the unit box of the soup
        inv = 1.0 / d; t1 = (0.0 - o) * inv; t2 = (1.0 - o) * inv
        tn = torch.minimum(t1, t2).max(1).values.clamp(min=0.0)
        pnt = o + d * tn[:, None]; lo = pnt.min(0).values; hi = pnt.max(0).values
    else:
        pnt = o
    q = ((pnt - lo) / (hi - lo) * (2 ** bits - 1)).clamp(0, 2 ** bits - 1)
    key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    if mode != "origin":
        octant = (d[:, 0] < 0).to(torch.int64) | ((d[:, 1] < 0).to(torch.int64) << 1) | ((d[:, 2] < 0).to(torch.int64) << 2)
        key = (octant << (3 * bits)) | key
    torch.cuda.synchronize(); ts = time.perf_counter()
    perm = torch.argsort(key); torch.cuda.synchronize(); tsort = (time.perf_counter() - ts) * 1e3
    os_ = o[perm].contiguous(); ds_ = d[perm].contiguous()
    t1_, out1 = t(os_, ds_)
    same = bool(torch.equal(out1[0], out0[0][perm]))
    print("%-14s %d bits: trace %.2f ms = %.0f Mrays/s (torch argsort %.1f ms), same hits %s" % (mode, bits, t1_, n / t1_ / 1e3, tsort, same))
    del perm, os_, ds_, out1
