"""host-array path (lh_accel_intersect_host) throughput with caller-owned, already-touched buffers"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lucille_amd as la
from oracle import pyoracle as po
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 20000000
P, idx, org, dr = po.soup(1000000, nr)
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build="host")
prim = np.zeros(nr, np.uint32); t = np.zeros(nr); u = np.zeros(nr); v = np.zeros(nr); occ = np.zeros(nr, np.uint8)
L = acc.L
for label, env in (("pipelined", None), ("simple", "1")):
    if env: os.environ["LH_HOST_SIMPLE"] = env
    for rep in range(3):
        t0 = time.perf_counter()
        rc = L.lh_accel_intersect_host(acc.h, nr, org.ctypes.data, dr.ctypes.data, prim.ctypes.data, t.ctypes.data, u.ctypes.data, v.ctypes.data, None, 0)
        t1 = time.perf_counter()
        rc2 = L.lh_accel_intersect_host(acc.h, nr, org.ctypes.data, dr.ctypes.data, None, None, None, None, occ.ctypes.data, 1)
        t2 = time.perf_counter()
    print(label, "closest %.1f Mrays/s (%.1f GB/s over the link), any-hit %.1f Mrays/s" % (nr / (t1 - t0) / 1e6, nr * 76 / (t1 - t0) / 1e9, nr / (t2 - t1) / 1e6), rc, rc2, flush=True)
