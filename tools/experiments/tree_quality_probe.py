"""host-built vs device-built traversal tree on the BASELINE config-5 scene: commit time, frame time, node visits / triangle
tests per ray (COUNT build).  python tools/tree_quality_probe.py [size] [tess] [samples]   (LH_DEVICE_* env knobs apply)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tess = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 64
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
meshes = [scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess) for k in range(int(g["ngeoms"]))]
ref = None
for on_device in (False, True):
    acc = la.HipAccel(0)
    for P, I in meshes:
        acc.add_mesh(P, I)
    t0 = time.perf_counter(); info = acc.commit(build="device" if on_device else "host"); tc = time.perf_counter() - t0
    acc.wait_exact()
    if os.environ.get("LH_STACK_CAP"):
        acc.set_param("stack_cap", int(os.environ["LH_STACK_CAP"])); acc.set_param("grid", 256 * int(os.environ.get("LH_WG_PER_CU", "3")))
    render.render_ao_frame(acc, cam, 1, ns, tile=size); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        img, st = render.render_ao_frame(acc, cam, 1, ns, tile=size); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    acc.trace_statistics(True); acc.statistics(clear=True)
    render.render_ao_frame(acc, cam, 1, ns, tile=size); torch.cuda.synchronize()
    s = acc.statistics(clear=True); acc.trace_statistics(False)
    nr = max(1, s["rays"])
    same = None if ref is None else bool(torch.equal(img, ref))
    if ref is None:
        ref = img.clone()
    print("%s tree: commit %.3f s (tree %.3f s), nodes %d depth %d | frame %.2f ms | per ray: %.2f node visits, %.2f triangle tests, %.3f fp64 | image equal: %s"
          % ("device" if on_device else "host", tc, info["build_seconds"], info["nnodes_traversal"], info["max_depth"], best * 1e3,
             s["nodes"] / nr, s["tris"] / nr, s["exact"] / nr, same), flush=True)
    acc.close()
