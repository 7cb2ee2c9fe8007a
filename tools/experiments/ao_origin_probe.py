"""round 4: the origin-first walk of the fused AO stage (set_param "ao_origin_first"): config-5 frame (21.1 M triangles, 4096^2,
64 AO samples), a device-built twin, and the config-2 frame -- frame time, node / triangle records per ray from the counting
kernels, and the frame compared with the from-the-root walk's bit for bit.  python tools/ao_origin_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes, render
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ao_c1.npz"))
c = g["camera"]
for tess, size, build in ((8, 4096, "host"), (8, 4096, "device"), (4, 2048, "host"), (0, 1024, "host")):
    acc = la.HipAccel(0)
    for k in range(int(g["ngeoms"])):
        Pk, Ik = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(Pk, Ik)
    acc.commit(build=build)
    cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
    base = None
    for up in (0, 1, 0, 1):
        acc.set_param("ao_origin_first", up)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fr, st = render.render_ao_frame(acc, cam, 1, 64, tile=size); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        if base is None: base = fr.clone()
        acc.trace_statistics(True); acc.statistics(clear=True)
        render.render_ao_frame(acc, cam, 1, 64, tile=size); torch.cuda.synchronize()
        cs = acc.statistics(clear=True); acc.trace_statistics(False)
        nr = max(1, cs["rays"])
        print("tess %d %dx%d %-6s origin_first %d  %.2f ms  nodes/ray %.2f tris/ray %.2f exact/ray %.4f  frame %s" % (
            tess, size, size, build, up, min(ts), cs["nodes"] / nr, cs["tris"] / nr, cs["exact"] / nr, "equal" if torch.equal(fr, base) else "DIFFERS"), flush=True)
    acc.close()
