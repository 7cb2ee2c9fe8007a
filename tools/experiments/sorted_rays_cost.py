"""round 5: what a physically sorted ray dump would cost end to end on S-soup-1M, 100 M rays, piece by piece (torch ops stand in for the
copy kernels): key + argsort are NOT what is measured here (a counting sort's passes took < 2 ms, lh_sort.hip's first version);
the gather of the rays into bin order, the walk of the sorted copy, the scatter of the records back."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from oracle import pyoracle as po
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000000
P, idx, org, dr = po.soup(1000000, n)
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
od = torch.from_numpy(org).cuda(); dd = torch.from_numpy(dr).cuda()
def ev(f, reps=3):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
lo = od.min(0).values; hi = od.max(0).values
c = ((od - lo) / (hi - lo) * 8).clamp(0, 7).to(torch.int64)
key = ((dd[:, 0] < 0).long() | ((dd[:, 1] < 0).long() << 1) | ((dd[:, 2] < 0).long() << 2)) * 512 + c[:, 0] * 64 + c[:, 1] * 8 + c[:, 2]
perm = torch.argsort(key); del key, c
out = acc.intersect_device(od, dd, mode=0)
print("walk, given order: %.2f ms" % ev(lambda: acc.intersect_device(od, dd, out=out, mode=0)), flush=True)
os_ = torch.empty_like(od); ds_ = torch.empty_like(dd)
def gather():
    torch.index_select(od, 0, perm, out=os_); torch.index_select(dd, 0, perm, out=ds_)
print("gather of the rays into bin order (2 x index_select of [n,3] fp64): %.2f ms" % ev(gather), flush=True)
outs = acc.intersect_device(os_, ds_, mode=0)
print("walk of the sorted copy, records in sorted order: %.2f ms" % ev(lambda: acc.intersect_device(os_, ds_, out=outs, mode=0)), flush=True)
back = [torch.empty_like(x) for x in outs]
def scatter():
    for b, s in zip(back, outs): b.index_copy_(0, perm, s)
print("scatter of the records back (prim, t, u, v): %.2f ms" % ev(scatter), flush=True)
print("records equal: %s" % all(torch.equal(a, b) for a, b in zip(back, out)), flush=True)
