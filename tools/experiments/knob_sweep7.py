"""Regroup / triangle-pass thresholds of the dumps over the 8-wide nodes (S-soup-10M, 50 M rays as in the bench's HBM leg), both modes.   python tools/experiments/knob_sweep7.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from oracle import pyoracle as po
PAIRS = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(32, 12), (40, 12), (48, 12), (40, 16), (48, 16), (40, 10), (36, 12), (44, 14), (32, 12)]
P, idx, org, dr = po.soup(10000000, 50000000); acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
o = torch.from_numpy(org).cuda(); d = torch.from_numpy(dr).cuda()
for mode in (0, 1):
    out = acc.intersect_device(o, d, mode=mode); torch.cuda.synchronize(); row = []
    for ma, tb in PAIRS:
        acc.set_param("min_active", ma); acc.set_param("tri_batch", tb); ts = []
        for _ in range(4):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); acc.intersect_device(o, d, out=out, mode=mode); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        row.append("(%d, %d) %.1f" % (ma, tb, o.shape[0] / min(ts) / 1e3))
    print("S-soup-10M %s (node bytes %d): %s" % ("closest" if mode == 0 else "any hit", acc.dump_node_bytes(), "  ".join(row)), flush=True)
