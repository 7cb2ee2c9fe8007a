"""round 4: regroup / triangle-pass thresholds on the config-2 frame (322 triangles, 1024^2, PixelSamples 3 3, 64 AO samples: short rays,
the refill dominates).  python tools/config2_knob_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, rib
sc = rib.RibScene(os.path.join(ROOT, "tests", "golden", "rib", "ambient_occlusion.rib"))
acc = la.HipAccel(0); sc.add_to(acc); acc.commit()
ps = int(sc.info.pixel_samples[0])
cam = la.Camera.make(1024, 1024, sc.camera.flength, list(sc.camera.cam2world), sc.camera.rh)
base = None
for ma, tb in ((32, 12), (48, 12), (24, 12), (16, 12), (8, 12), (4, 12), (16, 4), (16, 8), (16, 24), (8, 8), (8, 24), (32, 12)):
    acc.set_param("min_active", ma); acc.set_param("tri_batch", tb)
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fr, st = render.render_ao_frame(acc, cam, ps, 64, tile=1024); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    if base is None: base = fr.clone()
    print("min_active %2d tri_batch %2d  %.2f ms  frame %s" % (ma, tb, min(ts), "equal" if torch.equal(fr, base) else "DIFFERS"), flush=True)
