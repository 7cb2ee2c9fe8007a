"""device commit of the config-5 scene, phase by phase (LH_BUILD_TIMING=1): python tools/devbuild_timing.py [tess]"""
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
    t0 = time.perf_counter()
    for P, I in meshes:
        acc.add_mesh(P, I)
    t1 = time.perf_counter(); info = acc.commit(on_device=True); t2 = time.perf_counter()
    print("add_mesh %.3f s  commit %.3f s  (tree %.3f s, %d nodes, depth %d)" % (t1 - t0, t2 - t1, info["build_seconds"], info["nnodes_traversal"], info["max_depth"]), flush=True)
    acc.wait_exact(); acc.close()
