"""round 4: where a rank's share of a sharded config-5 frame goes (one GPU plays rank 0 of `world`): wall time of the batch and, with
LH_STAGE_TIMING=1 in the environment, the library's own stage times (stderr).  python tools/rank_stage_probe.py [world] [band_rows] [build] [workgroups per CU] [stack cap]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes, shard
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
want = int(sys.argv[2]) if len(sys.argv) > 2 else None
build = sys.argv[3] if len(sys.argv) > 3 else "device"
per_cu = float(sys.argv[4]) if len(sys.argv) > 4 else 0       # persistent workgroups per CU (0: the library's own grid)
cap = int(sys.argv[5]) if len(sys.argv) > 5 else 0             # cap of the LDS stack rows
size, tess, ns = 4096, 8, 64
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I)
acc.commit(build=build)
if per_cu > 0:
    acc.set_param("grid", int(torch.cuda.get_device_properties(0).multi_processor_count * per_cu)); acc.set_param("stack_cap", cap)
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
brow, y0s = render.bands_for(size, world, want)
ranks = [int(x) for x in os.environ["LH_PROBE_RANKS"].split(",")] if os.environ.get("LH_PROBE_RANKS") else (0, world - 1)
for r in ranks:
    mine = [y0s[b] for b in shard.bands_of_rank(len(y0s), r, world)]
    out = torch.zeros((len(mine), brow, size, 3), dtype=torch.float32, device="cuda")
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        acc.render_ao_bands(cam, mine, brow, 1, ns, seed=1, out=out); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("world %d rank %d: %d bands of %d rows, batch wall ms: %s" % (world, r, len(mine), brow, " ".join("%.2f" % t for t in ts)), flush=True)
