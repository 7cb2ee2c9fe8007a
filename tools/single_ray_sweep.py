"""threads x knobs of the coalesced one-ray path (tests/c/single_ray_threads.c): python tools/single_ray_sweep.py"""
import os, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lucille_amd import scenes
CSRC = os.path.join(ROOT, "lucille_amd", "csrc")
tmp = tempfile.mkdtemp()
exe = os.path.join(tmp, "srt")
subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "single_ray_threads.c"), "-o", exe,
                       "-L" + CSRC, "-llucille_hip", "-lpthread", "-Wl,-rpath," + CSRC])
def write(name, P, idx, org, dr):
    f = open(os.path.join(tmp, name), "wb")
    f.write(struct.pack("<I", len(P))); f.write(np.ascontiguousarray(P, np.float64).tobytes())
    f.write(struct.pack("<I", len(idx))); f.write(np.ascontiguousarray(idx, np.uint32).tobytes())
    f.write(struct.pack("<I", len(org))); f.write(np.ascontiguousarray(org).tobytes()); f.write(np.ascontiguousarray(dr).tobytes()); f.close()
P, idx, st = scenes.soup_triangles(200000, 0.01)
org, dr, _ = scenes.soup_rays(16000, st)
write("soup.bin", P, idx, org, dr)
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
Pa = np.concatenate([g["pos%d" % k] for k in range(int(g["ngeoms"]))]); off = np.cumsum([0] + [len(g["pos%d" % k]) for k in range(int(g["ngeoms"]))])
Ia = np.concatenate([g["idx%d" % k] + off[k] for k in range(int(g["ngeoms"]))]).astype(np.uint32)
lo, hi = Pa.min(0), Pa.max(0); rng = np.random.default_rng(1)
oa = rng.uniform(lo - 1, hi + 1, (16000, 3)); da = rng.uniform(lo, hi, (16000, 3)) - oa
write("ao.bin", Pa, Ia, oa, da)
for scene in ("ao.bin", "soup.bin"):
    for env in ({}, {"LH_COMB_SPIN_US": "0"}, {"LH_COMB_GATHER_US": "0"}, {"LH_COMB_GATHER_US": "100"}):
        for th, comb in ((1, 1), (4, 1), (16, 1), (16, 0)):
            if comb == 0 and env: continue
            r = subprocess.run([exe, os.path.join(tmp, scene), os.path.join(tmp, "out.bin"), str(th), str(comb)], capture_output=True, text=True, env=dict(os.environ, **env))
            print(scene, env, r.stdout.strip() or r.stderr[-300:], flush=True)
