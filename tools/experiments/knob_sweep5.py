"""More knobs of the headline dump at the new thresholds (S-soup-1M, 50 M rays, closest hit): visit budget, LDS copy of the tree's top, persistent grid.   python tools/experiments/knob_sweep5.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from oracle import pyoracle as po
P, idx, org, dr = po.soup(1000000, 50000000)
o = torch.from_numpy(org).cuda(); d = torch.from_numpy(dr).cuda()


def rate(acc, out):
    ts = []
    for _ in range(4):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); acc.intersect_device(o, d, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return o.shape[0] / min(ts) / 1e3


def sweep(name, values):
    row = []
    for v in values:
        acc = la.HipAccel(0); acc.add_mesh(P, idx)
        try:
            if v is not None: acc.set_param(name, v)
            acc.commit(); out = acc.intersect_device(o, d); torch.cuda.synchronize(); row.append("%s %.1f" % (v, rate(acc, out)))
        except Exception as e:
            row.append("%s ERR %s" % (v, str(e)[:40]))
        acc.close()
    print("%s: %s" % (name, "  ".join(row)), flush=True)


sweep("dump_budget", [None, 512, 1024, 2048, 4096, 16384])
sweep("top_nodes", [None, 0, 16, 64, 144, 256, 400])
sweep("grid", [None, 512, 768, 1024, 1280])
sweep("ray_chunk", [None, 64, 128, 192, 256, 384])
