"""A/B of the quad-coalesced node fetch (set_param "coop_fetch") against the default walk: python tools/coop_probe.py [nrays]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes

def rate(acc, o, d, mode, reps=4):
    out = acc.intersect_device(o, d, mode=mode); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); acc.intersect_device(o, d, out=out, mode=mode); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return o.shape[0] / best / 1e3, out

nr = int(sys.argv[1]) if len(sys.argv) > 1 else 50000000
for name, nt, he, n in (("S-soup-1M", 1000000, 0.005, nr), ("S-soup-10M (4-wide nodes forced)", 10000000, 0.002, nr // 2)):
    P, idx, st = scenes.soup_triangles(nt, he)
    ho, hd, _ = scenes.soup_rays(n, st)
    o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda()
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(); acc.set_param("wide8", 0)
    ref = None
    for rnd in range(2):
        for coop in (0, 1):
            acc.set_param("coop_fetch", coop)
            rc, oc = rate(acc, o, d, 0); ra, oa = rate(acc, o, d, 1)
            if ref is None:
                ref = ([x.clone() for x in oc], oa[0].clone())
            same = all(torch.equal(x, y) for x, y in zip(oc, ref[0])) and torch.equal(oa[0], ref[1])
            print("%s coop_fetch %d: closest %.0f any %.0f Mrays/s  records equal: %s" % (name, coop, rc, ra, same), flush=True)
    acc.close()
