"""lh_accel_intersect_host by batch size: the plain path (pageable hipMemcpy, one launch) against the pipelined one (pinned ring, pool copies), S-soup-1M,
closest hit, best of 5.  python tools/hostpath_sizes.py   (sets LH_PIPE_MIN=1 so that every size CAN take the pipelined path; LH_HOST_SIMPLE=1 forces the plain one)"""
import os, sys, time
os.environ["LH_PIPE_MIN"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
P, idx, st = scenes.soup_triangles(1000000, 0.005)
ho, hd, _ = scenes.soup_rays(8 << 20, st)
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
nmax = len(ho); hp = np.zeros(nmax, np.uint32); ht = np.zeros(nmax); hu = np.zeros(nmax); hv = np.zeros(nmax)
def run(n, simple):
    if simple: os.environ["LH_HOST_SIMPLE"] = "1"
    else: os.environ.pop("LH_HOST_SIMPLE", None)
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter()
        rc = acc.L.lh_accel_intersect_host(acc.h, n, ho.ctypes.data, hd.ctypes.data, hp.ctypes.data, ht.ctypes.data, hu.ctypes.data, hv.ctypes.data, None, 0)
        best = min(best, time.perf_counter() - t0); assert rc == 0
    return best, hp[:n].copy(), ht[:n].copy()
run(1 << 22, False); run(1 << 20, True)
print("rays      plain ms  Mrays/s   pipelined ms  Mrays/s   records equal")
for lg in range(int(os.environ.get('HP_LG0', 14)), 24):
    n = 1 << lg
    ts, ps, tts = run(n, True); tp, pp, ttp = run(n, False)
    print("%8d  %8.3f  %7.1f   %8.3f      %7.1f   %s" % (n, ts * 1e3, n / ts / 1e6, tp * 1e3, n / tp / 1e6, bool(np.array_equal(ps, pp) and np.array_equal(tts, ttp))))
