"""Second pass of knob_sweep3: the neighbourhood of (min_active 24, tri_batch 8) on the S-soup-1M dump in both modes, the S-soup-10M dump (8-wide nodes), and what the
same pair does to the path-traced config-4 frame and the config-5 AO frame.   python tools/experiments/knob_sweep4.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes
from oracle import pyoracle as po


def rate(acc, o, d, out, mode):
    ts = []
    for _ in range(4):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); acc.intersect_device(o, d, out=out, mode=mode); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return o.shape[0] / min(ts) / 1e3


def dump(ntri, nray, pairs):
    P, idx, org, dr = po.soup(ntri, nray); acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
    o = torch.from_numpy(org).cuda(); d = torch.from_numpy(dr).cuda()
    for mode in (0, 1):
        out = acc.intersect_device(o, d, mode=mode); torch.cuda.synchronize(); row = []
        for ma, tb in pairs:
            acc.set_param("min_active", ma); acc.set_param("tri_batch", tb); row.append("(%d, %d) %.1f" % (ma, tb, rate(acc, o, d, out, mode)))
        print("S-soup-%dM %s: %s" % (ntri // 1000000, "closest" if mode == 0 else "any hit", "  ".join(row)), flush=True)
    acc.close()


dump(1000000, 50000000, [(32, 12), (20, 6), (20, 8), (20, 10), (24, 6), (24, 8), (24, 10), (28, 6), (28, 8), (28, 10), (32, 8), (32, 12)])
dump(10000000, 20000000, [(32, 12), (24, 8), (24, 12), (32, 8), (40, 12), (32, 12)])
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_ps.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    acc.add_mesh(g["pos%d" % k], g["idx%d" % k])
    if ("nrm%d" % k) in g.files: acc.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
acc.commit()
c = g["camera"]; cam = la.Camera.make(2048, 2048, c[16], c[:16], int(c[19]))
for ma, tb in ((32, 12), (24, 8), (32, 8), (32, 12)):
    acc.set_param("min_active", ma); acc.set_param("tri_batch", tb); ts = []
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        img, st = render.render_pt_frame_sharded(acc, cam, 256, 0, 1, tile=2048, spp_chunk=256, kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=8, seed=7)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("config-4 frame (%d, %d): %.2f ms (mean %.9f)" % (ma, tb, min(ts[1:]), float(img.mean())), flush=True)
acc.close(); del img
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P_, I_ = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 8); acc.add_mesh(P_, I_); del P_, I_
acc.commit()
c = g["camera"]; cam = la.Camera.make(4096, 4096, c[16], c[:16], int(c[19]))
for ma, tb in ((32, 12), (24, 8), (32, 8), (24, 12), (32, 12)):
    acc.set_param("min_active", ma); acc.set_param("tri_batch", tb); ts = []
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        img, st = render.render_ao_frame(acc, cam, 1, 64, tile=4096)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("config-5 AO frame (%d, %d): %.2f ms (mean %.9f)" % (ma, tb, min(ts[1:]), float(img.mean())), flush=True)
