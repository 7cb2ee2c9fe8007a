"""the host builder's SAH constants (LH_BVH_CT: a triangle test against a node step, LH_BVH_CI) on the soup and on config 5:
python tools/host_sah_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes, render
n = 50_000_000
P, idx, st = scenes.soup_triangles(1000000, 0.005)
ho, hd, _ = scenes.soup_rays(n, st)
o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda(); del ho, hd
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "ao_c1.npz"))
meshes = [scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 8) for k in range(int(g["ngeoms"]))]
c = g["camera"]; cam = la.Camera.make(4096, 4096, c[16], c[:16], int(c[19]))
for ct in sys.argv[1:] or ["1.0", "1.2", "1.5", "2.0"]:
    os.environ["LH_BVH_CT"] = ct
    acc = la.HipAccel(0); acc.add_mesh(P, idx); info = acc.commit(build="host")
    out = acc.intersect_device(o, d); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(); acc.intersect_device(o, d, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    _, cnt = acc.intersect_device(o[:2000000], d[:2000000], counters=True)
    print("CT %s soup: %.0f Mrays/s, %d nodes, %.1f nodes + %.2f tris per ray" % (ct, n / min(ts) / 1e3, info["nnodes_traversal"], cnt["nodes"] / 2e6, cnt["tris"] / 2e6), flush=True)
    acc.close()
    acc = la.HipAccel(0)
    for Pk, Ik in meshes:
        acc.add_mesh(Pk, Ik)
    info = acc.commit(build="host"); fr = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); render.render_ao_frame(acc, cam, 1, 64, tile=4096); torch.cuda.synchronize(); fr.append((time.perf_counter() - t0) * 1e3)
    print("CT %s config 5: frame %.2f ms, %d nodes" % (ct, min(fr), info["nnodes_traversal"]), flush=True)
    acc.close()
