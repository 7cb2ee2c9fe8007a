"""A/B of the libraries in gpurun_variants/ on what a rank's share of the config-5 frame costs (rank 0 and rank 7 of 8, one batch each), on the whole
frame and on the S-soup-1M dump: one subprocess per library, interleaved rounds.   python tools/r06_ab_share.py [rounds]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
libs = sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "*.so")))
code = r'''
import sys, os, time; sys.path.insert(0, %r)
import numpy as np, torch; import lucille_amd as la; from lucille_amd import render, scenes, shard; from oracle import pyoracle as po
P, idx, org, dr = po.soup(1000000, 50000000); acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
o = torch.from_numpy(org).cuda(); d = torch.from_numpy(dr).cuda()
out = acc.intersect_device(o, d, mode=0); torch.cuda.synchronize(); ts = []
for _ in range(4):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(); acc.intersect_device(o, d, out=out, mode=0); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
soup = o.shape[0] / min(ts) / 1e3
acc.close(); del o, d, out
g = np.load(os.path.join(%r, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P_, I_ = scenes.tessellate(g["pos%%d" %% k], g["idx%%d" %% k], 8); acc.add_mesh(P_, I_); del P_, I_
acc.commit()
c = g["camera"]; cam = la.Camera.make(4096, 4096, c[16], c[:16], int(c[19]))
ts = []
for it in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img, st = render.render_ao_frame(acc, cam, 1, 64, tile=4096)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
brow, y0s = render.bands_for(4096, 8); sh = []
for r in (0, 3, 7):
    mine = [y0s[b] for b in shard.bands_of_rank(len(y0s), r, 8)]
    outb = torch.zeros((len(mine), brow, 4096, 3), dtype=torch.float32, device="cuda"); tt = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); acc.render_ao_bands(cam, mine, brow, 1, 64, seed=1, out=outb); torch.cuda.synchronize(); tt.append((time.perf_counter() - t0) * 1e3)
    sh.append(min(tt[1:]))
print("soup %%.1f Mrays/s   ao frame %%.2f ms (mean %%.9f)   shares of 8 (rank 0 / 3 / 7) %%.2f %%.2f %%.2f ms" %% (soup, min(ts[1:]), float(img.mean()), sh[0], sh[1], sh[2]))
''' % (ROOT, ROOT)
for r in range(rounds):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LH_LIBRARY=l), capture_output=True, text=True)
        print(os.path.basename(l), "round", r, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "ERR " + out.stderr[-400:], flush=True)
