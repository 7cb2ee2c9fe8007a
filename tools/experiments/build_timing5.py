import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import lucille_amd as la
from lucille_amd import scenes
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    Pk, Ik = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 8); acc.add_mesh(Pk, Ik)
t0 = time.perf_counter(); info = acc.commit(build="host"); print("POOL %s PARMIN %s: commit %.2f s, tree %.2f s, ref tree %.2f s" % (os.environ.get("LH_POOL_THREADS"), os.environ.get("LH_PAR_MIN_LOG2"), time.perf_counter() - t0, info["build_seconds"], info["ref_build_seconds"]))
