"""A/B of differently built liblucille_hip.so files (gpurun_variants/*.so) on the same box, one subprocess per library, interleaved
rounds: S-soup-1M closest-hit dump (Mrays/s), the path-traced config-4 frame and the config-5 AO frame (ms).   python tools/experiments/ab_frames.py [rounds]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
libs = sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "*.so")))
code = r'''
import sys, os, time; sys.path.insert(0, %r)
import numpy as np, torch; import lucille_amd as la; from lucille_amd import render; from oracle import pyoracle as po
P, idx, org, dr = po.soup(1000000, 50000000); acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build='host')
o = torch.from_numpy(org).cuda(); d = torch.from_numpy(dr).cuda()
out = acc.intersect_device(o, d, mode=0); torch.cuda.synchronize(); ts = []
for _ in range(4):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(); acc.intersect_device(o, d, out=out, mode=0); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
soup = o.shape[0] / min(ts) / 1e3
acc.close(); del o, d, out
g = np.load(os.path.join(%r, "tests", "golden", "ao_ps.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    acc.add_mesh(g["pos%%d" %% k], g["idx%%d" %% k])
    if ("nrm%%d" %% k) in g.files: acc.set_normals(k, g["nrm%%d" %% k], int(g["two_side%%d" %% k]))
acc.commit()
c = g["camera"]; cam = la.Camera.make(2048, 2048, c[16], c[:16], int(c[19]))
ts = []
for it in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img, st = render.render_pt_frame_sharded(acc, cam, 256, 0, 1, tile=2048, spp_chunk=256, kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=8, seed=7)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
pt = min(ts[1:]); ptmean = float(img.mean()); acc.close(); del img
from lucille_amd import scenes
g = np.load(os.path.join(%r, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P_, I_ = scenes.tessellate(g["pos%%d" %% k], g["idx%%d" %% k], 8); acc.add_mesh(P_, I_); del P_, I_
acc.commit()
c = g["camera"]; cam = la.Camera.make(4096, 4096, c[16], c[:16], int(c[19]))
ts = []
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img, st = render.render_ao_frame(acc, cam, 1, 64, tile=4096)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("soup %%.1f Mrays/s   pt frame %%.2f ms (mean %%.9f)   ao frame %%.2f ms (mean %%.9f)" %% (soup, pt, ptmean, min(ts[1:]), float(img.mean())))
''' % (ROOT, ROOT, ROOT)
for r in range(rounds):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LH_LIBRARY=l), capture_output=True, text=True)
        print(os.path.basename(l), "round", r, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "ERR " + out.stderr[-400:], flush=True)
