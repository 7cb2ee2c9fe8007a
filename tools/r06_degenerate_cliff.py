"""What ONE numerically collinear triangle of |e1|_1 |e2|_1 > 1 costs a ray dump: S-soup-1M in MILLIMETRES (the unit cube x 1000),
20 M rays, with and without one zero-area triangle appended (three different points on a line: it stays in the tree) -- a sliver of 30 x 50 mm in
a corner of the scene (s2 ~ 2 300: the cap 1 / s2 is far below every unit direction), then one that runs through the whole scene.
LH_DANGER_BOXES=0: round 5's rule (every ray beyond the cap takes the reference walk).   python tools/r06_degenerate_cliff.py [nrays]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
P, idx, st = scenes.soup_triangles(1_000_000, 0.005)
ho, hd, _ = scenes.soup_rays(nr, st)
P = P * 1000.0; ho = ho * 1000.0
o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda()
u = np.array([0.6, 0.1, 0.79]); u /= np.linalg.norm(u)
a = np.array([60.0, 80.0, 40.0]); Zs = np.stack([a, a + 30.0 * u, a + 1.6 * 30.0 * u])
a = np.array([500.0, 500.0, 500.0]); Zl = np.stack([a - 900.0 * u, a + 1100.0 * u, a - 900.0 * u + 1.75 * (2000.0 * u)])
for tag, PP in (("the soup", P), ("+ one sliver of 30 x 50 in a corner", np.concatenate([P, Zs])), ("+ one collinear triangle through the scene", np.concatenate([P, Zl]))):
    acc = la.HipAccel(0); acc.add_mesh(PP, np.arange(PP.shape[0], dtype=np.uint32)); acc.commit(); acc.wait_exact()
    for mode, name in ((la.MODE_CLOSEST, "closest"), (la.MODE_ANY, "any hit")):
        out = acc.intersect_device(o[:1000000], d[:1000000], mode=mode); torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter(); out = acc.intersect_device(o, d, mode=mode); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        hits = int((out[0] != (-1 if mode == la.MODE_CLOSEST else 0)).sum().item())
        print("%-44s %-8s %9.1f Mrays/s   hits %d" % (tag, name, nr / min(ts) / 1e6, hits), flush=True)
    acc.close()
