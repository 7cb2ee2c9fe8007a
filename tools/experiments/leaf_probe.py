import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes
def rate(acc, o, d, mode):
    out = acc.intersect_device(o, d, mode=mode); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); acc.intersect_device(o, d, out=out, mode=mode); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return o.shape[0] / best / 1e3
P, idx, st = scenes.soup_triangles(1000000, 0.005)
ho, hd, _ = scenes.soup_rays(30000000, st)
o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda()
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
c = g["camera"]; cam = la.Camera.make(2048, 2048, c[16], c[:16], int(c[19]))
for leaf in (1, 2, 3, 4):
    os.environ["LH_DEVICE_LEAF"] = str(leaf)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); info = acc.commit(on_device=True)
    _, cnt = acc.intersect_device(o[:2000000], d[:2000000], counters=True)
    print("soup-1M leaf<=%d: %d nodes depth %d build %.3f s; closest %.0f any %.0f; %.1f nodes + %.1f tris per ray" % (leaf, info["nnodes_traversal"], info["max_depth"],
          info["build_seconds"], rate(acc, o, d, 0), rate(acc, o, d, 1), cnt["nodes"] / 2e6, cnt["tris"] / 2e6), flush=True)
    acc.close()
    acc = la.HipAccel(0)
    for k in range(int(g["ngeoms"])):
        Pk, Ik = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 6); acc.add_mesh(Pk, Ik)
    info = acc.commit(on_device=True)
    render.render_ao_frame(acc, cam, 1, 64, tile=2048); torch.cuda.synchronize()
    t0 = time.perf_counter(); img, stt = render.render_ao_frame(acc, cam, 1, 64, tile=2048); torch.cuda.synchronize(); tf = time.perf_counter() - t0
    print("   AO tess-6 (1.3 M tris) 2048^2: %d nodes depth %d build %.3f s; frame %.1f ms" % (info["nnodes_traversal"], info["max_depth"], info["build_seconds"], tf * 1e3), flush=True)
    acc.close()
os.environ.pop("LH_DEVICE_LEAF")
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    Pk, Ik = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 6); acc.add_mesh(Pk, Ik)
info = acc.commit(build="host")
render.render_ao_frame(acc, cam, 1, 64, tile=2048); torch.cuda.synchronize()
t0 = time.perf_counter(); img, stt = render.render_ao_frame(acc, cam, 1, 64, tile=2048); torch.cuda.synchronize(); tf = time.perf_counter() - t0
print("   AO tess-6 HOST build: %d nodes depth %d build %.3f s; frame %.1f ms" % (info["nnodes_traversal"], info["max_depth"], info["build_seconds"], tf * 1e3))
