"""round 5: what would sorting an incoherent ray dump buy?  S-soup-1M / S-soup-10M, the bench's random rays: the same batch traced in
the given order and sorted (on the host, untimed) by a key of origin cell (Morton, 2^b cells per axis) and direction octant.
python tools/experiments/sorted_rays_probe.py [ntri] [nrays]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from oracle import pyoracle as po
ntri = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nrays = int(sys.argv[2]) if len(sys.argv) > 2 else 50000000
P, idx, org, dr = po.soup(ntri, nrays)
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
def part1by2(x):
    x = x.astype(np.uint64) & 0x3FF
    x = (x | (x << 16)) & 0x30000FF; x = (x | (x << 8)) & 0x300F00F; x = (x | (x << 4)) & 0x30C30C3; x = (x | (x << 2)) & 0x9249249
    return x
def key(org, dr, bits):
    lo = org.min(0); hi = org.max(0) + 1e-9
    c = np.minimum(((org - lo) / (hi - lo) * (1 << bits)).astype(np.int64), (1 << bits) - 1) if bits else np.zeros(org.shape, np.int64)
    m = part1by2(c[:, 0]) | (part1by2(c[:, 1]) << 1) | (part1by2(c[:, 2]) << 2)
    octant = (dr[:, 0] < 0).astype(np.uint64) | ((dr[:, 1] < 0).astype(np.uint64) << 1) | ((dr[:, 2] < 0).astype(np.uint64) << 2)
    return (octant << np.uint64(3 * bits)) | m
def timeit(o, d, mode):
    od = torch.from_numpy(o).cuda(); dd = torch.from_numpy(d).cuda()
    out = acc.intersect_device(od, dd, mode=mode); torch.cuda.synchronize(); ts = []
    for _ in range(4):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(); acc.intersect_device(od, dd, out=out, mode=mode); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return o.shape[0] / min(ts) / 1e3
print("%d triangles, %d rays" % (ntri, nrays), flush=True)
print("given order: closest %.1f Mrays/s, any-hit %.1f" % (timeit(org, dr, 0), timeit(org, dr, 1)), flush=True)
for bits in (0, 1, 2, 3):
    k = key(org, dr, bits); p = np.argsort(k, kind="stable")
    o2 = np.ascontiguousarray(org[p]); d2 = np.ascontiguousarray(dr[p])
    print("sorted by octant + %d-bit Morton cell (%d-bit key): closest %.1f Mrays/s, any-hit %.1f" % (bits, 3 + 3 * bits, timeit(o2, d2, 0), timeit(o2, d2, 1)), flush=True)
