"""A/B of liblucille_hip.so builds (gpurun_variants/*.so) on the ray-dump legs only, one subprocess per library, interleaved rounds:
S-soup-1M closest hit and any hit (50 M rays, the device builder's tree as in the bench, then the host builder's), node / triangle
records per ray from the counting launch, and a digest of the records (hits, sum of t) that must not depend on the library.
    python tools/experiments/ab_h16.py [rounds] [big]      big: S-soup-10M, 20 M rays (the 8-wide walk) as well"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
big = len(sys.argv) > 2 and sys.argv[2] == "big"
libs = sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "*.so")))
code = r'''
import sys, os, time; sys.path.insert(0, %r)
import numpy as np, torch; import lucille_amd as la; from oracle import pyoracle as po
def leg(ntri, nray, build):
    P, idx, org, dr = po.soup(ntri, nray); acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build=build)
    o = torch.from_numpy(org).cuda(); d = torch.from_numpy(dr).cuda(); res = []
    for mode in (0, 1):
        out = acc.intersect_device(o, d, mode=mode); torch.cuda.synchronize(); ts = []
        for _ in range(4):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(); acc.intersect_device(o, d, out=out, mode=mode); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        m = 2000000
        _, c = acc.intersect_device(o[:m].contiguous(), d[:m].contiguous(), mode=mode, counters=True)
        if mode == 0: dig = "hits %%d sum_t %%.6f" %% (int((out[0] >= 0).sum()), float(out[1][out[0] >= 0].sum()))
        else: dig = "occluded %%d" %% int(out[0].sum())
        res.append("%%s %%.1f Mrays/s (%%.2f + %%.2f rec/ray; %%s)" %% ("closest" if mode == 0 else "any", nray / min(ts) / 1e3, c["nodes"] / c["rays"], c["tris"] / c["rays"], dig))
    acc.close(); return "%%s-%%dM: " %% (build, ntri // 1000000) + "  ".join(res)
print(leg(1000000, 50000000, "device")); print(leg(1000000, 50000000, "host"))
if %r: print(leg(10000000, 20000000, "device"))
''' % (ROOT, big)
for r in range(rounds):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LH_LIBRARY=l), capture_output=True, text=True)
        lines = [x for x in out.stdout.strip().splitlines() if "Mrays/s" in x]
        for x in lines or ["ERR " + out.stderr[-600:]]: print(os.path.basename(l), "round", r, x, flush=True)
