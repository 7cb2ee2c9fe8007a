"""round 4: the top of the tree in LDS (set_param "top_nodes") and workgroups per CU (set_param "stack_cap"):
S-soup-1M closest-hit dump of 100 M rays on the host builder's and the device builder's tree, then the config-5 AO frame.
Every setting's records are compared with the default's, bit for bit.  python tools/top_probe.py [nrays]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes, render
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
P, idx, st = scenes.soup_triangles(1000000, 0.005)
ho, hd, _ = scenes.soup_rays(n, st)
o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda(); del ho, hd
def bench(acc, out, reps=3):
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(); acc.intersect_device(o, d, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return n / min(ts) / 1e3
for build in ("host", "device"):
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build=build)
    ref = acc.intersect_device(o, d); torch.cuda.synchronize()
    ref = [x.clone() for x in ref]
    out = acc.intersect_device(o, d); torch.cuda.synchronize()
    print("== S-soup-1M, %d rays, %s tree, LDS rows %d" % (n, build, -1), flush=True)
    for top, cap in ((0, 0), (16, 0), (32, 0), (96, 0), (144, 0), (0, 36), (0, 30), (96, 36), (352, 30), (416, 26), (224, 34)):
        acc.set_param("top_nodes", top); acc.set_param("stack_cap", cap)
        r = bench(acc, out)
        same = all(torch.equal(a, b) for a, b in zip(out, ref))
        print("top_nodes %3d stack_cap %2d  %.1f Mrays/s  records %s" % (top, cap, r, "equal" if same else "DIFFER"), flush=True)
    acc.close(); del ref, out
del o, d
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    Pk, Ik = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 8); acc.add_mesh(Pk, Ik)
acc.commit(build="host")
c = g["camera"]; cam = la.Camera.make(4096, 4096, c[16], c[:16], int(c[19]))
base = None
for top in (0, 32, 96, 0):
    acc.set_param("top_nodes", top)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fr, _ = render.render_ao_frame(acc, cam, 1, 64, tile=4096); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    if base is None: base = fr.clone()
    print("AO config 5  top_nodes %3d  %.2f ms  frame %s" % (top, min(ts), "equal" if torch.equal(fr, base) else "DIFFERS"), flush=True)
