"""host vs device build of the traversal tree, and what the device-built tree costs in traversal: python tools/build_probe.py [--soups-only]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes
def rate(acc, o, d, mode):
    out = acc.intersect_device(o, d, mode=mode); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); acc.intersect_device(o, d, out=out, mode=mode); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return o.shape[0] / best / 1e3
for name, nt, he, nr in (("S-soup-1M", 1000000, 0.005, 30000000), ("S-soup-10M", 10000000, 0.002, 20000000)):
    P, idx, st = scenes.soup_triangles(nt, he)
    ho, hd, _ = scenes.soup_rays(nr, st)
    o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda()
    for on_dev in (False, True):
        acc = la.HipAccel(0); acc.add_mesh(P, idx)
        t0 = time.perf_counter(); info = acc.commit(build="device" if on_dev else "host"); tc = time.perf_counter() - t0
        _, cnt = acc.intersect_device(o[:2000000], d[:2000000], counters=True)
        print("%s %s build: commit %.3f s (tree %.3f s, ref tree %.3f s%s), %d 4-wide nodes, depth %d; closest %.0f any %.0f Mrays/s; %.1f nodes + %.1f tris per ray"
              % (name, "DEVICE" if on_dev else "host  ", tc, info["build_seconds"], info["ref_build_seconds"], " in the background" if on_dev else "",
                 info["nnodes_traversal"], info["max_depth"], rate(acc, o, d, 0), rate(acc, o, d, 1), cnt["nodes"] / 2e6, cnt["tris"] / 2e6), flush=True)
        acc.close()
if "--soups-only" in sys.argv:
    sys.exit(0)
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
c = g["camera"]; cam = la.Camera.make(4096, 4096, c[16], c[:16], int(c[19]))
for on_dev in (False, True):
    acc = la.HipAccel(0)
    for k in range(int(g["ngeoms"])):
        Pk, Ik = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 8); acc.add_mesh(Pk, Ik)
    t0 = time.perf_counter(); info = acc.commit(build="device" if on_dev else "host"); tc = time.perf_counter() - t0
    render.render_ao_frame(acc, cam, 1, 64, tile=4096); torch.cuda.synchronize()
    t0 = time.perf_counter(); img, st = render.render_ao_frame(acc, cam, 1, 64, tile=4096); torch.cuda.synchronize(); tf = time.perf_counter() - t0
    t0 = time.perf_counter(); acc.wait_exact(); tw = time.perf_counter() - t0
    render.render_ao_frame(acc, cam, 1, 64, tile=4096); torch.cuda.synchronize()
    t0 = time.perf_counter(); img2, st2 = render.render_ao_frame(acc, cam, 1, 64, tile=4096); torch.cuda.synchronize(); tf2 = time.perf_counter() - t0
    print("   after waiting %.2f s for lucille's own tree: AO frame %.1f ms (fused AO stage), image mean %.6f, equal %s" % (tw, tf2 * 1e3, float(img2.mean()), bool(torch.equal(img, img2))))
    print("config 5 scene %s build: commit %.3f s (tree %.3f s), %d nodes, depth %d; AO frame %.1f ms, image mean %.6f"
          % ("DEVICE" if on_dev else "host  ", tc, info["build_seconds"], info["nnodes_traversal"], info["max_depth"], tf * 1e3, float(img.mean())), flush=True)
    acc.close()
