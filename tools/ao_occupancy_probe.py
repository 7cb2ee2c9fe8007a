"""round 4: workgroups per CU of the fused AO stage, with the persistent grid following (the r03 experiment changed LH_WG_PER_CU but
never the grid): config-5 frame (21.1 M triangles, 4096^2, 64 AO samples), grid = CUs x {3, 4, 5}, LDS stack rows capped so that
many fit; every frame compared with the default's bit for bit.  python tools/ao_occupancy_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes, render
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ao_c1.npz"))
c = g["camera"]
ncu = torch.cuda.get_device_properties(0).multi_processor_count
for tess, size, build in ((8, 4096, "device"),):
    acc = la.HipAccel(0)
    for k in range(int(g["ngeoms"])):
        Pk, Ik = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(Pk, Ik)
    info = acc.commit(build=build)
    cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
    base = None
    print("tess %d: depth %d" % (tess, info["max_depth"]), flush=True)
    for per_cu, cap, top, budget in ((4, 34, 0, 128), (4, 34, 0, 384), (4, 34, 0, 512), (4, 34, 0, 1024), (4, 34, 0, 2048), (4, 34, 0, 8192)):
        acc.set_param("grid", int(ncu * per_cu)); acc.set_param("stack_cap", cap); acc.set_param("top_nodes", top); acc.set_param("ray_budget", budget)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fr, st = render.render_ao_frame(acc, cam, 1, 64, tile=size); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        if base is None: base = fr.clone()
        print("tess %d %dx%d  workgroups per CU %s  stack_cap %2d  top_nodes %3d  budget %3d  %.2f ms  frame %s" % (tess, size, size, per_cu, cap, top, budget, min(ts), "equal" if torch.equal(fr, base) else "DIFFERS"), flush=True)
    acc.close()
