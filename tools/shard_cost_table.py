"""Per-shard cost table of the BASELINE config 5 AO frame from a ONE-GPU run, and the max-over-ranks frame time it
predicts for 2 / 4 / 8 GPUs under the shard assignment bench.py uses (render.bands_for + shard.bands_of_rank: full-width bands
dealt out in serpentine order) -- SURVEY 8e / VERDICT r01 item 2c, r04 item 1.  No multi-GPU hardware is involved: the prediction is
sum-of-my-bands + the gather of the other ranks' slabs at a stated link rate.
  python tools/shard_cost_table.py [size] [tess] [samples] > profiles/<round>_shard_cost_table.md"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes, shard
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tess = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 64
LINK_GBPS = 45.0          # one xGMI link, one direction, achievable (MI355X_MICROARCH.md: 7 links x ~153 GB/s bidirectional peak per GPU)
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0); ntri = 0
for k in range(int(g["ngeoms"])):
    P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I); ntri += I.shape[0] // 3
info = acc.commit()          # lh_accel_commit's own choice of builder (the device builders at this size)
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
def t_of(x0, y0, w, h, out):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        acc.render_ao_tile(cam, x0, y0, w, h, 1, ns, seed=1, out=out); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best
full = torch.empty((size, size, 3), dtype=torch.float32, device="cuda")
t1 = t_of(0, 0, size, size, full)
print("# Shard cost table: BASELINE config 5 AO frame (%d triangles, %dx%d, %d AO samples), one MI355X\n" % (ntri, size, size, ns))
print("Whole frame as ONE device batch: **%.2f ms** (tree %.2f s + reference-order tree %.2f s, once per scene: lh_accel_commit's own choice of builder).\n" % (t1 * 1e3, info["build_seconds"], info["ref_build_seconds"]))
SKEW_MS = {1: 0.0, 2: 0.10, 4: 0.16, 8: 0.17}      # what a frame's two barriers cost between real processes (profiles/r04_skew.txt, lh_dist_host_barrier, p50)
print("Shards = `render.bands_for(H, world)`: full-width bands (column 3: lines per band; the default first), dealt out in serpentine order (`shard.bands_of_rank`: groups of `world` bands, even groups in rank order, odd groups reversed); a rank's bands are ONE "
      "`lh_render_ao_bands` call (one device batch).  Times are best-of-4 wall times of every rank's batch (after one untimed pass over all ranks), run one after the other on one "
      "GPU.  Prediction for N ranks = max over ranks of its batch + gather, where the gather moves "
      "(N-1)/N of the frame (%d MB: one fp32 per pixel -- an AO frame is grey, rank 0 writes the value three times) to rank 0 over N-1 xGMI links in parallel at %.0f GB/s per link; the last column adds what the two barriers "
      "around a timed frame cost between real processes (p50: %s ms at 2 / 4 / 8 ranks, profiles/r04_skew.txt).\n" % (size * size * 4 // 1000000, LINK_GBPS, " / ".join("%.2f" % SKEW_MS[k] for k in (2, 4, 8))))
print("| ranks | bands | band rows | sum over ranks (ms) | busiest rank (ms) | least busy (ms) | imbalance | gather (ms) | predicted frame (ms) | predicted speed-up | with the barriers' skew |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
rows = {}; hits_of = {}; per_of = {}
slab = torch.zeros(size * size * 3 + 64 * size * 3, dtype=torch.float32, device="cuda")
for world, want_rows in ((1, None), (2, None), (4, None), (8, None), (8, 4), (8, 8), (8, 12), (8, 32), (8, 64)):
    brow, y0s = render.bands_for(size, world, want_rows)
    per = []
    # one untimed pass over every rank's batch first: the first batches after a change of band layout run 0.3-0.8 ms slow for several
    # repetitions (scratch buffers re-grown, clocks), which best-of-3 of the FIRST rank alone did not shake off (r05: rank 0 read
    # 7.68 ms where rank 1, with the same number of hits, read 7.28)
    for r in range(world if world > 1 else 0):
        mine = [y0s[b] for b in shard.bands_of_rank(len(y0s), r, world)]
        acc.render_ao_bands(cam, mine, brow, 1, ns, seed=1, out=slab[:len(mine) * brow * size * 3].view(len(mine), brow, size, 3)); torch.cuda.synchronize()
    for r in range(world):
        mine = [y0s[b] for b in shard.bands_of_rank(len(y0s), r, world)]
        out = slab[:len(mine) * brow * size * 3].view(len(mine), brow, size, 3)
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _, st_ = acc.render_ao_bands(cam, mine, brow, 1, ns, seed=1, out=out); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        per.append(best); hits_of.setdefault((world, brow), []).append(st_["primary_hits"])
    gather = 0.0 if world == 1 else (size * size * 4 / world) / (LINK_GBPS * 1e9)       # each peer sends its 1/N of the frame over its own link: ONE float per pixel (an AO frame is grey: render.py / lh_dist.hip k_take_channel0)
    pred = max(per) + gather
    rows.setdefault(world, (per, pred)); per_of[(world, brow)] = per
    print("| %d | %d | %d | %.2f | %.2f | %.2f | %.1f %% | %.2f | %.2f | %.2fx | %.2fx |" % (world, len(y0s), brow, sum(per) * 1e3, max(per) * 1e3, min(per) * 1e3,
          100.0 * (max(per) / (sum(per) / world) - 1.0), gather * 1e3, pred * 1e3, rows[1][1] / pred, rows[1][1] / (pred + SKEW_MS[world] * 1e-3)))
print("\nPer-rank batch time at 8 ranks, default bands (ms): " + " ".join("%.2f" % (x * 1e3) for x in rows[8][0]))
print("\nPer rank, 8 ranks, by band height -- batch ms (camera-ray hits of the rank, thousands):\n")
for (w_, b_), per in per_of.items():
    if w_ == 8:
        print("* %d lines: " % b_ + "  ".join("%.2f (%d)" % (x * 1e3, h // 1000) for x, h in zip(per, hits_of[(w_, b_)])))
print("\nReading: a rank renders ALL of its bands as one device batch (`lh_render_ao_bands`), so the per-launch drain of the persistent "
      "traversal kernel (as long as its slowest ray: bounded by the visit budget since round 3) is paid "
      "once per rank and frame.  The sum over ranks exceeds the one-batch frame by (ranks - 1) drains plus what the finer interleave costs in "
      "coherence; the imbalance column is the busiest rank against the mean (the extra 8-rank rows show other band heights: the "
      "default is the first).  Rendering the same bands one launch at a time costs +1.7 ms per band "
      "(`profiles/r02_shard_cost_table_bands_v1.md`: 3.75x predicted at 8 ranks).")
