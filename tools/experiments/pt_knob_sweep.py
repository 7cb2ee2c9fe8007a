"""round 5: the path-traced config-4 frame against the walk's knobs (min_active: lanes below which a wave regroups; tri_batch: parked
leaves a triangle pass waits for).   python tools/experiments/pt_knob_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_ps.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    acc.add_mesh(g["pos%d" % k], g["idx%d" % k])
    if ("nrm%d" % k) in g.files: acc.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
acc.commit()
c = g["camera"]; cam = la.Camera.make(2048, 2048, c[16], c[:16], int(c[19]))
def frame():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img, st = render.render_pt_frame_sharded(acc, cam, 256, 0, 1, tile=2048, spp_chunk=256, kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=8, seed=7)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, img
frame(); ref = frame()[1].clone()
for ma in (16, 24, 32, 40, 48, 56, 62):
    row = []
    for tb in (2, 4, 8, 12, 24):
        acc.set_param("min_active", ma); acc.set_param("tri_batch", tb)
        frame(); ms = min(frame()[0] for _ in range(3)); ok = bool(torch.equal(frame()[1], ref))
        row.append("%6.1f%s" % (ms, "" if ok else "!"))
    print("min_active %2d | tri_batch 2 4 8 12 24: %s" % (ma, " ".join(row)), flush=True)
