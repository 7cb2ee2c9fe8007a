"""the fused path-tracing pass (set_param "pt_fused": camera ray to last vertex inside the walk) against the wavefront passes:
same image bit for bit?  frame time?   python tools/pt_fused_probe.py [size] [spp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_ps.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    acc.add_mesh(g["pos%d" % k], g["idx%d" % k])
    if ("nrm%d" % k) in g.files:
        acc.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
acc.commit()
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
def frame(chunk):
    best = 1e9
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        img, st = render.render_pt_frame_sharded(acc, cam, spp, 0, 1, tile=size, spp_chunk=chunk, kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=8, seed=7)
        torch.cuda.synchronize()
        if it: best = min(best, time.perf_counter() - t0)
    return img, st, best
chunk = max(1, min(spp, (256 << 20) // (size * size)))
ref, st0, t0 = frame(chunk)
print("wavefront: %.1f ms, %d rays, depth %d, mean %.9f" % (t0 * 1e3, st0["rays"], max(st0.values())*0+st0.get("max_depth", -1), float(ref.mean())), flush=True)
acc.set_param("pt_fused", 1)
for grid in (0, 256, 512, 768, 1024):
    acc.set_param("pt_grid", grid)
    for ch in (chunk, spp):
        img, st, t = frame(ch)
        print("fused grid %4d spp/pass %3d: %.1f ms, %d rays, depth %d, image equal %s (max |diff| %.2e), stats equal %s"
              % (grid, ch, t * 1e3, st["rays"], st.get("max_depth", -1), bool(torch.equal(img, ref)), float((img - ref).abs().max()), st == st0), flush=True)
