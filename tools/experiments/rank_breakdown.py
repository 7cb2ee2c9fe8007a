"""Where a rank's share of the BASELINE config-5 AO frame spends its time (one GPU): stage times (LH_STAGE_TIMING HIP events)
of the whole frame and of every rank's batch at world = 8 under three shard layouts -- interleaved 4-line bands (round 2),
contiguous strips, interleaved 64-line bands -- plus the histogram of node visits per AO ray (COUNT build) of the whole frame.
  LH_STAGE_TIMING=1 python tools/rank_breakdown.py [size] [tess] [samples] 2> gpurun_out/rank_breakdown.log"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes, shard
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tess = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 64
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I)
acc.commit(build="host")
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
slab = torch.zeros(size * size * 3 + 64 * size * 3, dtype=torch.float32, device="cuda")
def say(msg):
    sys.stderr.write(msg + "\n"); sys.stderr.flush()
def batch(mine, brow):
    out = slab[:len(mine) * brow * size * 3].view(len(mine), brow, size, 3)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        acc.render_ao_bands(cam, mine, brow, 1, ns, seed=1, out=out); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best
say("== whole frame")
say("wall %.2f ms" % (batch([0], size) * 1e3))
world = 8
for name, brow in (("interleaved 4-line bands", 4), ("interleaved 64-line bands", 64), ("contiguous strips", size // world)):
    y0s = list(range(0, size, brow))
    say("== world 8, %s" % name)
    for r in range(world):
        mine = [y0s[b] for b in shard.tiles_of_rank(len(y0s), r, world)]
        say("rank %d wall %.2f ms" % (r, batch(mine, brow) * 1e3))
say("== node visits per ray, whole frame (COUNT build)")
os.environ["LH_DEBUG_COUNTERS"] = "1"
acc.trace_statistics(True)
batch([0], size)
say("statistics %s" % (acc.statistics(),))
