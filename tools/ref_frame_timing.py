"""BASELINE config 2 (1024 x 1024, 64 AO samples, 3x3 pixel samples as the example RIB asks) through the REFERENCE's
own renderer with the HIP accelerator bound, per mode of integration/ri_render_hip.c, next to lsh_hip.  GPU box.
  python tools/ref_frame_timing.py [size] [gather] [pixel_samples]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import ref_rib
from tests.helpers import load_golden
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
gather = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = load_golden("ao_c1")
c2w = np.asarray(g["camera"][:16]).reshape(4, 4)
w2c = np.linalg.inv(c2w @ np.linalg.inv(np.diag([1.0, 1.0, -1.0, 1.0])))
scene = {"ngeoms": int(g["ngeoms"]), "w2c": w2c, "fov": 45.0}
for k in range(int(g["ngeoms"])):
    scene["pos%d" % k] = g["pos%d" % k]; scene["idx%d" % k] = g["idx%d" % k]
tmp = tempfile.mkdtemp(); sp = os.path.join(tmp, "scene.npz"); np.savez(sp, **scene)
kw = dict(width=size, height=size, gather_nsamples=gather, pixel_samples=ps, lib="liblucille_ref_hip.so", record=False)
for mode, method in (("batched", 2), ("replay", 2)):
    t0 = time.time()
    o = ref_rib.render_scene_subprocess(sp, os.path.join(tmp, mode + ".npz"), accel_method=method, env={"RI_HIP_RENDER": mode}, **kw)
    print("reference renderer, accel hip, RI_HIP_RENDER=%s: %.2f s wall incl. process start, Ri ingest, BVH build (image mean %.4f)" % (mode, time.time() - t0, o["image"].mean()), flush=True)
small = dict(kw); small.update(width=128, height=128, gather_nsamples=16)
t0 = time.time(); ref_rib.render_scene_subprocess(sp, os.path.join(tmp, "rays.npz"), accel_method=2, env={"RI_HIP_RENDER": "rays"}, **small)
t1 = time.time(); ref_rib.render_scene_subprocess(sp, os.path.join(tmp, "cpu.npz"), accel_method=1, env={"RI_HIP_RENDER": "rays"}, **small)
t2 = time.time()
print("128x128 / 16 samples: one-ray HIP vtable %.2f s, reference CPU BVH %.2f s" % (t1 - t0, t2 - t1))
rib = os.path.join(ROOT, "tests", "golden", "rib", "ambient_occlusion.rib")
t0 = time.time()
r = subprocess.run([os.path.join(ROOT, "lucille_amd", "csrc", "lsh_hip"), "--resolution", "%dx%d" % (size, size), "--gather", str(gather),
                    "--pixelsamples", str(ps), "--output", os.path.join(tmp, "o.hdr"), rib], capture_output=True, text=True)
print("lsh_hip: %.2f s wall;" % (time.time() - t0), [l for l in r.stdout.splitlines() if "Rendering" in l])
