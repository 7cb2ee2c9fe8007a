"""Regroup threshold x triangle-pass threshold x rays per cursor atomic of the headline dump (S-soup-1M, 50 M rays, closest hit, the device builder's
tree, four workgroups per CU) at the round-5 kernels: the last sweep of these knobs (r03) ran at three workgroups per CU.   python tools/experiments/knob_sweep3.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from oracle import pyoracle as po
P, idx, org, dr = po.soup(1000000, 50000000)
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
o = torch.from_numpy(org).cuda(); d = torch.from_numpy(dr).cuda()
out = acc.intersect_device(o, d); torch.cuda.synchronize()


def rate():
    ts = []
    for _ in range(4):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); acc.intersect_device(o, d, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return o.shape[0] / min(ts) / 1e3


print("default (min_active 32, tri_batch 12): %.1f Mrays/s" % rate(), flush=True)
for ma in (16, 24, 32, 40, 48, 56):
    row = []
    for tb in (4, 8, 12, 16, 24, 32):
        acc.set_param("min_active", ma); acc.set_param("tri_batch", tb); row.append("%7.1f" % rate())
    print("min_active %2d  tri_batch 4 / 8 / 12 / 16 / 24 / 32: %s" % (ma, " ".join(row)), flush=True)
acc.set_param("min_active", 32); acc.set_param("tri_batch", 12)
for rc in (256, 512, 1024, 2048, 4096):
    acc.set_param("ray_chunk", rc); print("ray_chunk %4d: %.1f Mrays/s" % (rc, rate()), flush=True)
