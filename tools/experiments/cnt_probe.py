import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
P, idx, st = scenes.soup_triangles(1000000, 0.005)
ho, hd, _ = scenes.soup_rays(8000000, st)
o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda()
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build="host")
for mode in (0, 1):
    _, c = acc.intersect_device(o, d, mode=mode, counters=True)
    print(mode, c, flush=True)
