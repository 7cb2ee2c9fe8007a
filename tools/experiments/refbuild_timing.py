"""commit of the config-5 scene with both trees on the device, phase by phase: python tools/refbuild_timing.py [tess]   (LH_REF_BUILD=host for the round-2 path)"""
import os, sys, time
os.environ["LH_BUILD_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import lucille_amd as la
from lucille_amd import scenes
tess = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "ao_c1.npz"))
meshes = [scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess) for k in range(int(g["ngeoms"]))]
for it in range(3):
    acc = la.HipAccel(0)
    for P, I in meshes:
        acc.add_mesh(P, I)
    t1 = time.perf_counter(); info = acc.commit(on_device=True); t2 = time.perf_counter()
    acc.wait_exact(); t3 = time.perf_counter()
    print("commit %.3f s (traversal tree %.3f s, lucille's own tree %.3f s), exact after another %.3f s" % (t2 - t1, info["build_seconds"], info["ref_build_seconds"], t3 - t2), flush=True)
    acc.close()
P, idx, st = scenes.soup_triangles(10000000, 0.002)
for it in range(2):
    acc = la.HipAccel(0); acc.add_mesh(P, idx)
    t1 = time.perf_counter(); info = acc.commit(on_device=True); t2 = time.perf_counter()
    acc.wait_exact(); t3 = time.perf_counter()
    print("S-soup-10M: commit %.3f s (traversal tree %.3f s, lucille's own tree %.3f s), exact after another %.3f s" % (t2 - t1, info["build_seconds"], info["ref_build_seconds"], t3 - t2), flush=True)
    acc.close()
