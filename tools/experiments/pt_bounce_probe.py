"""round 5: what one bounce of the path-traced config-4 frame costs, by difference: frames with max_vertices = 2 (camera rays only),
3 (one bounce), 4 (two) -- frame time, and with the counting kernels the node / triangle steps, lane slots, regroup iterations
and rays through the reference walk of each.   python tools/experiments/pt_bounce_probe.py [size] [spp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    if ("nrm%d" % k) in g.files: acc.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
acc.commit()
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
def frame(mv):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img, st = render.render_pt_frame_sharded(acc, cam, spp, 0, 1, tile=size, spp_chunk=spp, kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=mv, seed=7)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, st
prev = None
for mv in (2, 3, 4, 8):
    frame(mv); ms = min(frame(mv)[0] for _ in range(3)); st = frame(mv)[1]
    acc.trace_statistics(True); acc.statistics(clear=True); acc.slot_statistics(clear=True)
    frame(mv)
    s = acc.statistics(clear=True); sl = acc.slot_statistics(clear=True); acc.trace_statistics(False)
    row = dict(ms=ms, **s, **sl, retraced=int(acc.L.lh_accel_last_retraced(acc.h)))
    print("max_vertices %d: %s" % (mv, row), flush=True)
    if prev:
        d = {k: row[k] - prev[k] for k in row}
        r = max(1, d["rays"])
        print("   difference: %.2f ms for %d rays; per ray: nodes %.2f tris %.2f exact %.3f; lane use node steps %.3f triangle steps %.3f; regroup iterations per ray %.3f; ref walk %d"
              % (d["ms"], d["rays"], d["nodes"] / r, d["tris"] / r, d["exact"] / r, d["nodes"] / max(1, d["node_slots"]), d["tris"] / max(1, d["tri_slots"]), d["regroups"] / r, d["retraced"]), flush=True)
    prev = row
