"""config-5 frame on the device-built tree: python tools/devtree_frame.py [tess] [size]   (env LH_DEVICE_CUT, LH_DEVICE_LEAF, LH_BUILD_TIMING)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes
tess = int(sys.argv[1]) if len(sys.argv) > 1 else 8
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "ao_c1.npz"))
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
keep = None
if os.environ.get("WITH_HOST_ACCEL"):                  # bench.py's situation: the host-built accelerator of the same scene stays alive
    keep = la.HipAccel(0)
    for k in range(int(g["ngeoms"])):
        P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); keep.add_mesh(P, I)
    keep.commit(build="host")
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); render.render_ao_frame(keep, cam, 1, 64, tile=size); torch.cuda.synchronize()
        print("host-tree frame %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    if os.environ.get("WITH_HOST_ACCEL") == "close":
        keep.close(); keep = None
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I)
t0 = time.perf_counter(); info = acc.commit(on_device=True); tc = time.perf_counter() - t0
acc.wait_exact()
ts = []
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img, st = render.render_ao_frame(acc, cam, 1, 64, tile=size); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
acc.trace_statistics(True); acc.statistics(clear=True)
render.render_ao_frame(acc, cam, 1, 64, tile=size); torch.cuda.synchronize()
cnt = acc.statistics(clear=True)
print("frames", ["%.1f" % t for t in ts], "rows", acc.L.lh_accel_trace_rows(acc.h) if hasattr(acc.L, "lh_accel_trace_rows") else "?")
print("cut %s: commit %.3f s (tree %.3f s), %d nodes depth %d, frame %.2f ms, nodes/ray %.2f tris/ray %.2f, mean %.6f" % (
    os.environ.get("LH_DEVICE_CUT", "default"), tc, info["build_seconds"], info["nnodes_traversal"], info["max_depth"], min(ts),
    cnt["nodes"] / cnt["rays"], cnt["tris"] / cnt["rays"], float(img.mean().item())))
