"""path-traced frame (BASELINE config 4 shape) against the paths per pass: python tools/pt_chunk_probe.py"""
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
    if ("nrm%d" % k) in g.files:
        acc.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
acc.commit(build="host")
c = g["camera"]; size = 2048; spp = 256
cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
ref = None
for chunk in (16, 32, 64, 128, 256):
    best = 1e9
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        img, st = render.render_pt_frame_sharded(acc, cam, spp, 0, 1, tile=size, spp_chunk=chunk, kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=8, seed=7)
        torch.cuda.synchronize()
        if it: best = min(best, time.perf_counter() - t0)
    if ref is None: ref = img.clone()
    print("spp per pass %3d (%4d M paths, %.1f GB of path state): frame %.1f ms, %.0f Mrays/s, image equal to the 16-spp passes': %s, max |diff| %.2e"
          % (chunk, size * size * chunk >> 20, size * size * chunk * 170 / 1e9, best * 1e3, st["rays"] / best / 1e6, bool(torch.equal(img, ref)), float((img - ref).abs().max())), flush=True)
