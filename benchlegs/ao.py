"""bench.py: BASELINE config 5 -- the tessellated AO example scene, 4096^2, 64 AO samples; sharded in serpentine bands at N > 1."""
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403 -- the constants and helpers every leg shares
from .common import ROOT, gather_ceiling, pmc_source, host_cores, newest_pmc_summary, VALU_NODE_STEP_ANYHIT


def ao_frame_leg(la, acc_device, rank, world, size, nsamples, steps, dev, tess, build="auto", twin=True):
    """Secondary leg = BASELINE config 5 as stated (4096 x 4096, 64 AO samples, the AO example scene -- the
    322 triangles the reference's own RIB ingest produced, tests/golden/ao_c1.npz -- midpoint-tessellated
    `tess` = 8 times: 21.1 M triangles >= 10 M); --ao-size 1024 --ao-tess 0 is config 2.  Whole pipeline on
    the device (camera rays, hits, epilogue, AO rays, occlusion, radiance), tiles sharded over the ranks
    with one gather of tile slabs to rank 0 (strong scaling: the frame is fixed)."""
    import torch
    from lucille_amd import render, scenes
    g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
    from lucille_amd import shard
    acc = la.HipAccel(acc_device)
    ntri = sum(int(g["idx%d" % k].shape[0]) // 3 for k in range(int(g["ngeoms"]))) * 4 ** tess
    def add_meshes(a):
        for k in range(int(g["ngeoms"])):
            P_, I_ = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess)
            a.add_mesh(P_, I_)
            del P_, I_
    # ONE build (rank 0: tessellation + commit), then the broadcast of the flattened scene to every rank
    t0c = time.perf_counter(); info, commit_s, bcast_s = shard.commit_shared(acc, add_meshes, rank, world, build=build); commit_main_s = time.perf_counter() - t0c
    dev_built = info["nnodes"] == info["nnodes_traversal"]; other_b = "host" if dev_built else "device"
    c = g["camera"]
    cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
    # one GPU: the whole frame as one tile; sharded: full-width bands in serpentine order, ONE device batch per rank (render.bands_for / shard.bands_of_rank / lh_render_ao_bands)
    tile = None if world > 1 else min(size, 4096)
    times = []; st = None; img = None; stats = []
    for it in range(steps + 1):
        shard.barrier()
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        if world > 1:
            img, st = render.render_ao_frame_sharded(acc, cam, 1, nsamples, rank, world)
        else:
            img, st = render.render_ao_frame(acc, cam, 1, nsamples, tile=tile)
        torch.cuda.synchronize(dev)
        shard.barrier()
        stats.append(dict(st))
        if it > 0:
            times.append(time.perf_counter() - t0)
    # in-run validation: every timed frame produced the same counts; a differently tiled render of the
    # same frame (untimed) is bit-equal -- the RNG is keyed by absolute sample position
    ok = all(s == stats[0] for s in stats)
    if world == 1:
        img2, st2 = render.render_ao_frame(acc, cam, 1, nsamples, tile=max(256, size // 4))
        ok = ok and bool(torch.equal(img, img2)) and st2 == stats[0]
    # what the frame's rays cost: one more frame (untimed) through the counting instantiations of the same kernels
    roof = None
    if world == 1:
        acc.trace_statistics(True); acc.statistics(clear=True); acc.slot_statistics(clear=True)
        render.render_ao_frame(acc, cam, 1, nsamples, tile=tile); torch.cuda.synchronize(dev)
        c = acc.statistics(clear=True); sl = acc.slot_statistics(clear=True); acc.trace_statistics(False)
        nr = max(1, c["rays"])
        # The frame is NOT bandwidth-bound (coherent rays: 0.15 KB of fabric traffic per ray).  With four workgroups per CU the fused
        # any-hit kernel is bound by VALU ISSUE: the newest profiles/r*_pmc_ao_dense.txt (tools/pmc_cmd.sh: separate SQ / TCC passes of
        # this frame; its last block is the fused AO launch) -- SQ_INSTS_VALU wave instructions x 4 cycles / 1024 SIMDs against the
        # launch's cycles (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8).  `achieved` / `peak` are VALU lane operations per second: what
        # the walk's own steps need -- an ANY-HIT node step does not rank its children: VALU_NODE_STEP_ANYHIT (109; the ranked step's
        # 136 is the closest-hit walk's), 75 per triangle record through the fp32 filter (profiles/step_costs.json) -- x the counted
        # steps of the frame, against 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz.  The camera rays' closest-hit steps (4 % of the frame's
        # rays) are priced like the rest.  What `frac` leaves out is what the counters show the pipes are busy WITH besides: idle
        # lanes, the refill of finished lanes (ray generation + set-up, ~300 instructions a regroup), the fp64 resolves.
        recs = c["nodes"] + c["tris"]
        b_frame = 64.0 * c["nodes"] + 40.0 * c["tris"] + (48.0 + 28.0) * st["primary_rays"] + 4.0 * st["primary_hits"]
        lane_ops = VALU_NODE_STEP_ANYHIT * c["nodes"] + VALU_TRI_STEP * c["tris"]
        src, k = newest_pmc_summary("pmc_ao_dense.txt")
        sq = None
        if k is not None and k.get("SQ_INSTS_VALU") and k.get("GRBM_GUI_ACTIVE"):
            gv = k.get
            sq = {"source": "%s (tools/pmc_cmd.sh: separate SQ / TCC passes of this frame; the fused any-hit launch)" % src,
                  "valu_busy": round(gv("SQ_INSTS_VALU") * 4.0 / 1024.0 / (gv("GRBM_GUI_ACTIVE") / 8.0), 3),
                  "valu_lane_use": round(gv("SQ_THREAD_CYCLES_VALU", 0.0) / max(1.0, gv("SQ_ACTIVE_INST_VALU", 0.0) * 64.0), 3),
                  "wave_cycles_waiting": round(gv("SQ_WAIT_ANY", 0.0) / max(1.0, gv("SQ_WAVE_CYCLES", 0.0)), 3),
                  "l2_hit_rate": round(gv("TCC_HIT_sum", 0.0) / max(1.0, gv("TCC_HIT_sum", 0.0) + gv("TCC_MISS_sum", 0.0)), 3),
                  "valu_wave_instructions_per_ray": round(gv("SQ_INSTS_VALU") / max(1, st["ao_rays"]), 1),
                  "fabric_read_bytes_per_frame": gv("TCC_EA0_RDREQ_sum", 0.0) * 128.0}
        roof = {"bound": "valu issue", "achieved": round(lane_ops / min(times) / 1e12, 2), "peak": VALU_PEAK_TLANEOPS, "unit": "T lane-ops/s",
                "frac": round(lane_ops / min(times) / 1e12 / VALU_PEAK_TLANEOPS, 4), "traffic": sq["fabric_read_bytes_per_frame"] if sq else None,
                "formula": "achieved = (%d x node steps [the any-hit step: no ranking] + %d x triangle records of the counted frame) / frame time; peak = 256 CUs x 64 lanes x 2.4 GHz"
                           % (VALU_NODE_STEP_ANYHIT, VALU_TRI_STEP),
                "node_steps": int(c["nodes"]), "triangle_records": int(c["tris"]),
                "counters": sq, "records_per_s": round(recs / min(times) / 1e9, 1), "algorithmic_GBps": round(b_frame / min(times) / 1e9, 1),
                "nodes_per_ray": round(c["nodes"] / nr, 2), "tris_per_ray": round(c["tris"] / nr, 2), "exact_per_ray": round(c["exact"] / nr, 4),
                "lane_use_node_steps": round(c["nodes"] / max(1, sl["node_slots"]), 3),
                "lane_use_triangle_passes": round(c["tris"] / max(1, sl["tri_slots"]), 3),
                "rays_counted": c["rays"]}
    # the same scene through the OTHER builder (the device builders of lh_build.hip are lh_accel_commit's own choice from 1 M
    # triangles on, the host builder below that): commit time, frame time on that tree, and the image -- which must not change by a bit
    devb = None
    if world == 1 and twin:
        acc_d = la.HipAccel(acc_device)
        for k in range(int(g["ngeoms"])):
            P_, I_ = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc_d.add_mesh(P_, I_); del P_, I_
        t0 = time.perf_counter(); info_d = acc_d.commit(build=other_b); commit_other_s = time.perf_counter() - t0
        acc_d.wait_exact(); exact_s = time.perf_counter() - t0          # lucille's own tree attached: ties, fragile hits, beams follow the reference
        render.render_ao_frame(acc_d, cam, 1, nsamples, tile=tile); torch.cuda.synchronize(dev)
        tfd = []
        for _ in range(max(1, min(steps, 3))):                  # as the frames above: the best of the timed frames
            t0 = time.perf_counter(); img_d, st_d = render.render_ao_frame(acc_d, cam, 1, nsamples, tile=tile); torch.cuda.synchronize(dev)
            tfd.append(time.perf_counter() - t0)
        devb = {"builder": other_b, "commit_s": round(commit_other_s, 3), "tree_s": round(info_d["build_seconds"], 3),
                "reference_tree_s": round(acc_d.info()["ref_build_seconds"], 3), "commit_to_exact_s": round(exact_s, 3),
                "frame_ms": round(min(tfd) * 1e3, 3),
                "image_bit_equal": bool(torch.equal(img_d, img)) and dict(st_d) == stats[0]}
        ok = ok and devb["image_bit_equal"]
        acc_d.close(); del img_d
    rays_all = shard.all_reduce_sum(float(st["primary_rays"] + st["ao_rays"])) if world > 1 else float(st["primary_rays"] + st["ao_rays"])
    t_all = shard.all_reduce_max(min(times)) if world > 1 else min(times)
    ranks = None
    if world > 1:       # one more frame, untimed, with the device synchronised between a rank's batch and the gather: who did what
        tm = {}
        shard.barrier()
        render.render_ao_frame_sharded(acc, cam, 1, nsamples, rank, world, timing=tm)
        shard.barrier()
        d_ = shard.dist()
        tm.update(rank=rank, transport="rccl" if (d_ is not None and d_.transport == la.DIST_RCCL) else "shm", frame_ms_best=round(min(times) * 1e3, 3),
                  rays=int(st["primary_rays"] + st["ao_rays"]))
        print("[bench rank %d] ao_render %s" % (rank, json.dumps(tm)), file=sys.stderr, flush=True)
        ranks = shard.all_gather_object(tm)
    # N > 1: the gathered frame against the SAME frame rendered as one batch on rank 0's own replica (untimed): bit for bit
    # (the sample stream is keyed by absolute pixel and sample, so sharding must not move a bit)
    sharded_equal = None
    if world > 1 and rank == 0:
        one, st_one = render.render_ao_frame(acc, cam, 1, nsamples, tile=min(size, 4096)); torch.cuda.synchronize(dev)
        sharded_equal = bool(torch.equal(one, img)) and int(st_one["primary_rays"] + st_one["ao_rays"]) == int(rays_all)
        ok = ok and sharded_equal; del one
    ok_all = (shard.all_reduce_min(1.0 if ok else 0.0) if world > 1 else (1.0 if ok else 0.0)) > 0.5
    acc.close()
    if rank != 0:
        return None
    return {"workload": "BASELINE config 5: examples/ambient_occlusion scene tessellated to %d tris, %dx%d, %d AO samples, frame wall incl. ray gen + tile gather"
                        % (ntri, size, size, nsamples), "triangles": ntri,
            "tile": tile if tile is not None else "%d full-width bands of %d rows dealt out to %d ranks in serpentine order (shard.bands_of_rank), one device batch per rank, one byte per pixel gathered (the count of unoccluded rays; a float when samples per pixel > 1 or AO rays > 255)" % (
                len(render.bands_for(size, world)[1]), render.bands_for(size, world)[0], world),
            "device_bytes": info["device_bytes"], "build_s": round(info["build_seconds"], 3),
            "ref_tree_build_s": round(info["ref_build_seconds"], 3),
            "builder": ("device" if dev_built else "host") + (": lh_accel_commit's own choice at this size" if build == "auto" else ", asked for")
                       + "; `other_builder` is the same frame on the other builder's tree",
            "scene_load": {"rank0_tessellate_and_commit_s": round(commit_main_s, 3), "rank0_commit_s": round(commit_s, 3), "broadcast_s": round(bcast_s, 3) if world > 1 else None},
            "rays_per_frame": int(rays_all), "frame_ms": round(t_all * 1e3, 3),
            "value": round(rays_all / t_all / 1e6, 1), "unit": "Mrays/s", "scaling": "strong",
            "image_mean": float(img.mean().item()), "roofline": roof, "other_builder": devb, "ranks": ranks,
            "validation": {"frames_repeat": ok_all, "retiled_frame_bit_equal": bool(ok) if world == 1 else None, "sharded_frame_equals_one_batch": sharded_equal,
                           "primary_hits": int(stats[0]["primary_hits"]) if world == 1 else None, "ok": ok_all}}
