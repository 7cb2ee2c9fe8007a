"""bench.py: BASELINE config 4 -- plane_sphere path traced, 2048^2, 256 spp."""
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403 -- the constants and helpers every leg shares
from .common import ROOT, gather_ceiling, pmc_source, host_cores


def pt_frame_leg(la, acc_device, rank, world, size, spp, dev):
    """Secondary leg (BASELINE config 4): examples/plane_sphere (the 1 986 triangles + vertex normals
    the reference's RIB ingest produced, tests/golden/ao_ps.npz), size x size, spp paths per pixel,
    diffuse wavefront path tracer, tiles sharded tile_id % world + gather of tile slabs to rank 0."""
    import torch
    from lucille_amd import render
    g = np.load(os.path.join(ROOT, "tests", "golden", "ao_ps.npz"))
    from lucille_amd import shard
    acc = la.HipAccel(acc_device)
    def add_meshes(a):
        for k in range(int(g["ngeoms"])):
            a.add_mesh(g["pos%d" % k], g["idx%d" % k])
            if ("nrm%d" % k) in g.files:
                a.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
    shard.commit_shared(acc, add_meshes, rank, world, build="host")
    c = g["camera"]
    cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
    times = []; st = None; img = None; first = None; repeat = True
    pt_tile = size if world == 1 else max(128, size // 4)
    # paths per pass (decided once: the first frame's buffers stay allocated): as many as 70 % of the free HBM holds (164 B of
    # path state each; a 2048^2 x 256 spp frame is 2^30 paths = 176 GB of the 288): every pass costs one kernel ramp + drain per
    # bounce, so fewer, larger wavefronts are faster (tools/experiments/pt_frames.py: 166.3 / 154.7 / 148.6 ms per frame as 4 / 2 / 1 passes;
    # the image does not change by a bit)
    torch.cuda.empty_cache()
    free_b = torch.cuda.mem_get_info(dev)[0]
    per_pass = max(64 << 20, min(1 << 30, int(free_b * 7 // 10 // 164)))
    # sharded: a rank's interleaved 4-line bands (1 / world of the frame) are one pass per sample chunk (render_pt_frame_sharded)
    area = pt_tile * pt_tile if world == 1 else max(1, size * size // world)
    chunk = max(1, min(spp, per_pass // area))
    while spp % chunk:            # whole passes
        chunk -= 1
    for it in range(3):
        shard.barrier()
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        img, st = render.render_pt_frame_sharded(acc, cam, spp, rank, world, tile=pt_tile, spp_chunk=chunk,
                                                 kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=8, seed=7)
        torch.cuda.synchronize(dev)
        shard.barrier()
        if it > 0:
            times.append(time.perf_counter() - t0)
        if rank == 0:
            if it == 0:
                first = img.clone()
            else:
                repeat = repeat and bool(torch.equal(img, first))
    retiled = None
    if world == 1:
        # the same frame cut into four tiles (other wavefront sizes, other compaction orders): every pixel's paths are keyed by
        # (pixel, sample), so the image must not change by a bit
        t2 = size // 2
        img2, _ = render.render_pt_frame_sharded(acc, cam, spp, rank, world, tile=t2, spp_chunk=max(1, min(spp, (64 << 20) // (t2 * t2))),
                                                 kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=8, seed=7)
        torch.cuda.synchronize(dev)
        retiled = bool(torch.equal(img2, img)); del img2
    roof = None
    if world == 1:
        # one more frame (untimed) through the counting instantiation of the trace kernel.  Every ray of a bounce goes through
        # HBM as fp64 records: 48 B written by the shader, 48 B read by the trace kernel (camera rays: generated in the kernel,
        # nothing), 28 B of hit record written and read again by the shader; plus 64 B per node visit and 40 B per triangle test
        acc.trace_statistics(True); acc.statistics(clear=True)
        render.render_pt_frame_sharded(acc, cam, spp, rank, world, tile=size, spp_chunk=max(1, min(spp, (64 << 20) // (size * size))),
                                       kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=8, seed=7)
        torch.cuda.synchronize(dev)
        c = acc.statistics(clear=True); acc.trace_statistics(False)
        nr = max(1, c["rays"])
        b_frame = 64.0 * c["nodes"] + 40.0 * c["tris"] + 2 * 28.0 * c["rays"] + 2 * 48.0 * (c["rays"] - st["paths"])
        roof = {"bound": "hbm", "achieved": round(b_frame / min(times) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(b_frame / min(times) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                "nodes_per_ray": round(c["nodes"] / nr, 2), "tris_per_ray": round(c["tris"] / nr, 2), "exact_per_ray": round(c["exact"] / nr, 4),
                "rays_counted": c["rays"],
                "note": "1 986 triangles: the tree is L2-resident.  Per pass: closest hit with the camera rays generated in the kernel, then per "
                        "bounce a decision kernel (streaming: miss / roulette, radiance into per-pixel 64-bit fixed-point sums, survivors compacted), "
                        "a scatter kernel over the survivors (hit epilogue, lobe, next ray) and one closest-hit launch: 22 launches, no host round "
                        "trip; kernel time = frame time, closest-hit kernels 71 % of it (profiles/r05_pt_timeline.txt)"}
    rays_all = shard.all_reduce_sum(float(st["rays"])) if world > 1 else float(st["rays"])
    t_all = shard.all_reduce_max(min(times)) if world > 1 else min(times)
    acc.close()
    if rank != 0:
        return None
    return {"workload": "examples/plane_sphere (1986 tris, vertex normals), %dx%d, %d spp, <=8 path vertices, kd 0.8, frame wall incl. ray gen, shading, compaction, tile gather"
                        % (size, size, spp),
            "rays_per_frame": int(rays_all), "frame_ms": round(t_all * 1e3, 3),
            "value": round(rays_all / t_all / 1e6, 1), "unit": "Mrays/s", "scaling": "strong",
            "spp_per_pass": chunk, "paths_per_pass": chunk * area,
            "shards": "one tile" if world == 1 else "full-width 4-line bands, band_id %% %d, a rank's bands = one pass per sample chunk (lh_render_pt_bands)" % world,
            "image_mean": float(img.mean().item()), "roofline": roof,
            "parity": "every bounce's closest-hit records are the pinned kernel's (bit-equal to the compiled reference on the same rays); the TRANSPORT "
                      "arithmetic (roulette, lobe choice, weights) is parity-UNPINNED: the reference's pathtrace.c is dead code that does not compile, "
                      "there is nothing to run it against (SURVEY 8f-3; HISTORY.md 11 lists the departures from its text)",
            # white furnace with albedo 0.8 under a unit environment: every pixel's radiance lies in (0, 1]
            "validation": {"frames_repeat": repeat, "retiled_frame_bit_equal": retiled,
                           "radiance_in_0_1": bool(float(img.min().item()) >= 0.0 and float(img.max().item()) <= 1.0 + 1e-6),
                           "ok": repeat and retiled is not False and 0.0 < float(img.mean().item()) <= 1.0}}
