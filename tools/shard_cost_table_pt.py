"""Per-shard cost table of the BASELINE config 4 path-traced frame (examples/plane_sphere, 2048 x 2048, 256 spp) from a
ONE-GPU run, and the max-over-ranks frame time it predicts for 2 / 4 / 8 GPUs under the shard assignment bench.py uses
(render.render_pt_frame_sharded: tiles of size/4, tile_id % world).  Prediction = busiest rank's tiles + gather of the other
ranks' slabs at a stated link rate.   python tools/shard_cost_table_pt.py [size] [spp] [tile ...] >> profiles/<round>_shard_cost_table.md"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, shard
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tiles_arg = [int(x) for x in sys.argv[3:]] or [size // 4, size // 8]
LINK_GBPS = 45.0
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_ps.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    acc.add_mesh(g["pos%d" % k], g["idx%d" % k])
    if ("nrm%d" % k) in g.files:
        acc.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
info = acc.commit(build="host")
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
KW = dict(kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=8, seed=7)
def t_tile(x0, y0, w, h):
    out = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    best = 1e9
    for _ in range(3):
        out.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        acc.render_pt_tile(cam, x0, y0, w, h, 0, spp, spp, out=out, **KW); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best
t1 = t_tile(0, 0, size, size)
print("\n# Shard cost table: BASELINE config 4 path-traced frame (plane_sphere, %d triangles, %dx%d, %d spp, <= 8 vertices), one MI355X\n" % (info["ntriangles"], size, size, spp))
print("Whole frame as ONE pass (2^30 paths): **%.2f ms**.  Shards = square tiles, `tile_id %% world == rank` (render_pt_frame_sharded); a tile is one pass "
      "(15 launches).  Times: best of 3 per tile, one after the other on one GPU.  Gather: (N-1)/N of the %d MB frame over N-1 links at %.0f GB/s.\n" % (t1 * 1e3, size * size * 12 // 1000000, LINK_GBPS))
print("| tile | tiles | ranks | sum over ranks (ms) | busiest rank (ms) | least busy (ms) | imbalance | gather (ms) | predicted frame (ms) | predicted speed-up |")
print("|---|---|---|---|---|---|---|---|---|---|")
for tile in tiles_arg:
    tiles = shard.tile_grid(size, size, tile)
    cost = [t_tile(*t) for t in tiles]
    for world in (1, 2, 4, 8):
        per = [sum(cost[i] for i in shard.tiles_of_rank(len(tiles), r, world)) for r in range(world)]
        gather = 0.0 if world == 1 else (size * size * 12 / world) / (LINK_GBPS * 1e9)
        pred = max(per) + gather
        print("| %d | %d | %d | %.2f | %.2f | %.2f | %.1f %% | %.2f | %.2f | %.2fx |" % (tile, len(tiles), world, sum(per) * 1e3, max(per) * 1e3, min(per) * 1e3,
              100.0 * (max(per) / (sum(per) / world) - 1.0), gather * 1e3, pred * 1e3, t1 / pred))
    print("\nPer-tile cost, tile %d (ms, row-major from the frame's first line): " % tile + " ".join("%.2f" % (x * 1e3) for x in cost) + "\n")

# a rank's interleaved 4-line bands as ONE pass (what bench.py does at N > 1: lh_render_pt_bands)
print("\nBands: full-width bands of `rows` lines, `band_id % world == rank`, all of a rank's bands as ONE pass (`lh_render_pt_bands`).\n")
print("| band rows | ranks | sum over ranks (ms) | busiest rank (ms) | least busy (ms) | imbalance | gather (ms) | predicted frame (ms) | predicted speed-up |")
print("|---|---|---|---|---|---|---|---|---|")
mat = la.Material.make(kd=(0.8,) * 3)
acc.set_environment((1.0, 1.0, 1.0), None)
for rows in (4, 16):
    nb = size // rows
    for world in (2, 4, 8):
        per = []
        for r in range(world):
            cnt = len(range(r, nb, world))
            out = torch.zeros((cnt, rows, size, 3), dtype=torch.float32, device="cuda"); best = 1e9
            for _ in range(3):
                out.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
                acc.render_pt_bands(cam, rows * r, rows, rows * world, cnt, 0, spp, spp, max_vertices=8, override=mat, seed=7, out=out); torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            per.append(best); del out
        gather = (size * size * 12 / world) / (LINK_GBPS * 1e9)
        pred = max(per) + gather
        print("| %d | %d | %.2f | %.2f | %.2f | %.1f %% | %.2f | %.2f | %.2fx |" % (rows, world, sum(per) * 1e3, max(per) * 1e3, min(per) * 1e3,
              100.0 * (max(per) / (sum(per) / world) - 1.0), gather * 1e3, pred * 1e3, t1 / pred))
