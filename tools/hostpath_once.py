"""the bench's host_path leg by itself: S-soup-1M, <nrays> rays in pageable host arrays through lh_accel_intersect_host (the pipelined path), best
of 3, records compared with the device path's.  python tools/hostpath_once.py [nrays]   (LH_PIPE_CHUNK / LH_PIPE_DEPTH / LH_COPY_THREADS from the
environment: tools/r06_hostpath_sweep.sh)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 20000000
P, idx, st = scenes.soup_triangles(1000000, 0.005)
ho, hd, _ = scenes.soup_rays(nr, st)
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
hp = np.zeros(nr, np.uint32); ht = np.zeros(nr); hu = np.zeros(nr); hv = np.zeros(nr)
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    rc = acc.L.lh_accel_intersect_host(acc.h, nr, ho.ctypes.data, hd.ctypes.data, hp.ctypes.data, ht.ctypes.data, hu.ctypes.data, hv.ctypes.data, None, 0)
    ts.append(time.perf_counter() - t0); assert rc == 0
o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda()
dev = acc.intersect_device(o, d); torch.cuda.synchronize()
same = bool(np.array_equal(hp, dev[0].cpu().numpy().view(np.uint32)) and all(np.array_equal(x, dev[k].cpu().numpy()) for k, x in ((1, ht), (2, hu), (3, hv))))
occ = np.zeros(nr, np.uint8)
t0 = time.perf_counter(); rc = acc.L.lh_accel_intersect_host(acc.h, nr, ho.ctypes.data, hd.ctypes.data, None, None, None, None, occ.ctypes.data, 1); ta = time.perf_counter() - t0
same_any = bool(np.array_equal(occ.astype(bool), hp != 0xFFFFFFFF))
best = min(ts[1:])
print("chunk %s depth %s threads %s: %.0f Mrays/s closest (%.1f GB/s of 76 B per ray; calls %s ms), any-hit %.0f Mrays/s; records == device path: %s, any == closest's hits: %s" % (
    os.environ.get("LH_PIPE_CHUNK", "default"), os.environ.get("LH_PIPE_DEPTH", "default"), os.environ.get("LH_COPY_THREADS", "default"),
    nr / best / 1e6, nr * 76 / best / 1e9, " ".join("%.1f" % (x * 1e3) for x in ts), nr / ta / 1e6, same, same_any))
