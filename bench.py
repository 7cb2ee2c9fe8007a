#!/usr/bin/env python
"""bench.py -- the driver's benchmark contract for the lucille hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE pass of the hot path (BVH traversal + ray/triangle intersection,
closest-hit records prim/t/u/v) over one synthetic ray batch that is already
resident in HBM.  Workload (BASELINE.json config 3, the one the north-star target
"~1M-tri scene" is quoted on): S-soup-1M = 1,000,000 random triangles
(SURVEY.md Appendix C generator, seed 88172645463325252), --rays incoherent rays
per GPU (default 100M).  Weak scaling: every rank traces its own ray batch against
a replicated BVH; no data-path collective (rays are independent).  Rank 0's batch
is the canonical Appendix C ray stream; rank r>0 continues from a rank-derived
xorshift state.

One JSON line on rank 0: value = total Mrays/s over all ranks (max-over-ranks
time), plus `roofline` (algorithmic bytes of the dominant kernel / its HIP-event
duration vs the 8 TB/s HBM peak) and `cpu_baseline` (the compiled reference, or
the oracle port, timed on this box's host cores on a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
B_IN, B_OUT, B_TRI = 32, 16, 40   # SURVEY.md 8(d) algorithmic bytes per ray / per triangle test
B_NODE = {"f32": 64, "q16": 32, "q16x4": 64}   # per node visit: SURVEY's 64-B fp32 2-wide node, 32-B 16-bit grid 2-wide, 64-B 16-bit grid 4-wide


def hip_event_timer():
    """HIP events on an explicit stream, straight from libamdhip64 (torch.cuda.Event only
    sees torch's current stream; the kernel is launched on the stream we pass)."""
    import torch
    hip = C.CDLL("libamdhip64.so")
    hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipEventSynchronize.argtypes = [C.c_void_p]
    hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
    return hip


def rank_state(seed_after_tris, rank):
    if rank == 0:
        return seed_after_tris
    x = (seed_after_tris ^ (0x9E3779B97F4A7C15 * (rank + 1))) & 0xFFFFFFFFFFFFFFFF
    x ^= (x >> 30); x = (x * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x ^= (x >> 27); x = (x * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    x ^= (x >> 31)
    return x or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rays", type=int, default=100_000_000, help="rays per GPU per step")
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--half-extent", type=float, default=0.005)
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--mode", choices=["closest", "any"], default="closest")
    ap.add_argument("--cpu-rays", type=int, default=1_500_000, help="cpu_baseline sample size")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-ao", action="store_true", help="skip the secondary AO-frame leg")
    ap.add_argument("--no-pt", action="store_true", help="skip the secondary path-traced leg")
    ap.add_argument("--pt-size", type=int, default=2048)      # BASELINE config 4: 2048 x 2048, 256 spp
    ap.add_argument("--pt-spp", type=int, default=256)
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--device-override", type=int, default=None,
                    help="testing only: put every rank on this device (2 ranks on a 1-GPU box, use with --backend gloo)")
    ap.add_argument("--ao-size", type=int, default=4096)
    ap.add_argument("--ao-samples", type=int, default=64)
    ap.add_argument("--ao-tess", type=int, default=7, help="midpoint-subdivision levels of the example scene (4^n x 322 triangles; 7 -> 5.3 M, 8 -> 21 M)")
    args = ap.parse_args()

    import torch
    import lucille_amd as la
    from lucille_amd import shard
    # bench needs the synthetic generator, which lives with the checkers; it is used
    # here only to MAKE inputs and (cpu_baseline leg) to time the CPU path
    from oracle import pyoracle as po

    if args.device_override is not None:
        os.environ["LH_DEVICE_OVERRIDE"] = str(args.device_override)
    rank, world, local = shard.init_process_group(backend=args.backend)
    assert world == args.gpus, "launch with --nproc-per-node == --gpus"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU fallback)")
    if args.device_override is not None:
        local = args.device_override
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # ---- synthetic inputs (host) -> HBM -----------------------------------------
    L = po.lib()
    st = C.c_uint64(po.SOUP_SEED)
    P = np.empty((3 * args.tris, 3), np.float64); idx = np.empty(3 * args.tris, np.uint32)
    L.lo_soup_triangles(C.byref(st), args.tris, args.half_extent, P.ctypes.data_as(C.POINTER(C.c_double)),
                        idx.ctypes.data_as(C.POINTER(C.c_uint32)))
    st = C.c_uint64(rank_state(st.value, rank))
    n = args.rays
    d_org = torch.empty((n, 3), dtype=torch.float64, device=dev)
    d_dir = torch.empty((n, 3), dtype=torch.float64, device=dev)
    chunk = 10_000_000
    ho = np.empty((min(chunk, n), 3)); hd = np.empty((min(chunk, n), 3))
    first_org = first_dir = None
    for b in range(0, n, chunk):
        m = min(chunk, n - b)
        L.lo_soup_rays(C.byref(st), m, ho.ctypes.data_as(C.POINTER(C.c_double)), hd.ctypes.data_as(C.POINTER(C.c_double)))
        d_org[b:b + m].copy_(torch.from_numpy(ho[:m])); d_dir[b:b + m].copy_(torch.from_numpy(hd[:m]))
        if b == 0:
            first_org = ho[:min(m, args.cpu_rays)].copy(); first_dir = hd[:min(m, args.cpu_rays)].copy()

    acc = la.HipAccel(local)
    acc.add_mesh(P, idx)
    info = acc.commit()

    mode = la.MODE_CLOSEST if args.mode == "closest" else la.MODE_ANY
    out = acc.intersect_device(d_org, d_dir, mode=mode, variant=args.variant)   # allocates outputs (untimed)
    torch.cuda.synchronize(dev)

    # ---- algorithmic bytes per ray: counted launch on a sample (untimed) ---------
    ns = min(n, 4_000_000)
    _, cnt = acc.intersect_device(d_org[:ns], d_dir[:ns], mode=mode, variant=args.variant, counters=True)
    n_nodes = cnt["nodes"] / ns; n_tris = cnt["tris"] / ns
    b_out = B_OUT if mode == la.MODE_CLOSEST else 4
    node_fmt = {"f32": "f32", "q16": "q16"}.get(os.environ.get("LH_NODE_FORMAT", ""), "q16x4")
    b_ray = B_IN + b_out + B_NODE[node_fmt] * n_nodes + B_TRI * n_tris

    # ---- timed region -------------------------------------------------------------
    hip = hip_event_timer()
    stream = torch.cuda.current_stream(dev)
    sptr = C.c_void_p(stream.cuda_stream)
    ev = [(C.c_void_p(), C.c_void_p()) for _ in range(args.steps)]
    for a, b in ev:
        hip.hipEventCreate(C.byref(a)); hip.hipEventCreate(C.byref(b))

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        acc.intersect_device(d_org, d_dir, out=out, mode=mode, variant=args.variant)
    barrier(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        hip.hipEventRecord(ev[k][0], sptr)
        acc.intersect_device(d_org, d_dir, out=out, mode=mode, variant=args.variant)
        hip.hipEventRecord(ev[k][1], sptr)
    torch.cuda.synchronize(dev); barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kms = []
    for a, b in ev:
        ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), a, b); kms.append(ms.value)
    kernel_ms = float(np.mean(kms))

    if world > 1:
        rdev = dev if torch.distributed.get_backend() == "nccl" else torch.device("cpu")
        tt = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- context figures (rank 0, untimed for `value`): the box's copy bandwidth and the host path ----
    copy_gbps = host_path = None
    if rank == 0 and world == 1 and not args.no_cpu:
        big = torch.empty(1 << 30, dtype=torch.uint8, device=dev); dst = torch.empty_like(big)
        dst.copy_(big); torch.cuda.synchronize(dev)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(big)
        e1.record(); torch.cuda.synchronize(dev)
        copy_gbps = 10 * 2 * big.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9       # read + write
        del big, dst
        nh = min(n, 20_000_000)
        h_org = np.ascontiguousarray(d_org[:nh].cpu().numpy()); h_dir = np.ascontiguousarray(d_dir[:nh].cpu().numpy())
        # caller-owned, already-touched result arrays (a fresh allocation would time page faults, not the path)
        hp = np.zeros(nh, np.uint32); ht = np.zeros(nh); hu = np.zeros(nh); hv = np.zeros(nh)
        best = None
        for _ in range(2):
            th = time.perf_counter()
            rc = acc.L.lh_accel_intersect_host(acc.h, nh, h_org.ctypes.data, h_dir.ctypes.data, hp.ctypes.data, ht.ctypes.data,
                                               hu.ctypes.data, hv.ctypes.data, None, 0)
            th = time.perf_counter() - th
            assert rc == 0
            best = th if best is None else min(best, th)
        host_path = {"value": round(nh / best / 1e6, 1), "unit": "Mrays/s", "link_GBps": round(nh * 76 / best / 1e9, 1),
                     "sample": "%d rays through lh_accel_intersect_host: pageable host arrays -> pinned staging in 2 M-ray chunks on two "
                               "streams, 48 B/ray up + 28 B/ray down over PCIe; never the headline value" % nh}

    ao = None
    if not args.no_ao:
        ao = ao_frame_leg(la, acc_device=local, rank=rank, world=world, size=args.ao_size, nsamples=args.ao_samples,
                          steps=max(2, args.steps), dev=dev, tess=args.ao_tess)

    pt = None
    if not args.no_pt:
        pt = pt_frame_leg(la, acc_device=local, rank=rank, world=world, size=args.pt_size, spp=args.pt_spp, dev=dev)

    if rank == 0:
        total_rays = n * world * args.steps
        value = total_rays / elapsed / 1e6
        achieved = b_ray * n / (kernel_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("rays_per_launch") == n and j.get("mode") == args.mode and j.get("kernel_tag") == node_fmt \
                        and args.variant in (-1, 4):
                    traffic = j.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "Mrays/s (primary+AO)", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 filter + f64 resolve (hit records f64)",
            "data": "synthetic",
            "config": {"workload": "S-soup-1M ray dump (BASELINE config 3): %d random triangles, %d incoherent rays per GPU, %s-hit"
                                   % (args.tris, n, args.mode),
                       "rays_per_gpu": n, "triangles": args.tris, "mode": args.mode,
                       "variant": args.variant, "parallelism": "replicated BVH, ray slices x%d" % world,
                       "bvh": {"nodes": info["nnodes"], "depth": info["max_depth"], "device_bytes": info["device_bytes"],
                               "build_s": round(info["build_seconds"], 3)}},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "kernel": "k_trace_persist_lane<walk=spec,%s nodes>" % node_fmt if args.variant in (-1, 4) else "k_trace_v%d" % args.variant,
                         "node_bytes": B_NODE[node_fmt],
                         "kernel_ms": round(kernel_ms, 3), "bytes_per_ray": round(b_ray, 1),
                         "nodes_per_ray": round(n_nodes, 3), "tris_per_ray": round(n_tris, 3)},
        }
        if copy_gbps is not None:
            # the box's own device-to-device copy rate next to the 8 TB/s datasheet peak
            res["roofline"]["measured_copy_GBps"] = round(copy_gbps, 1)
            res["roofline"]["frac_of_measured_copy"] = round(achieved / copy_gbps, 4)
        if host_path is not None:
            res["host_path"] = host_path
        if ao is not None:
            res["ao_render"] = ao
        if pt is not None:
            res["pt_render"] = pt
        if not args.no_cpu and world == 1:            # rank 0 at N = 1 only (the contract)
            res["cpu_baseline"] = cpu_baseline(po, P, idx, first_org, first_dir)
        print(json.dumps(res), flush=True)
    acc.close()
    if world > 1:
        torch.distributed.destroy_process_group()


def ao_frame_leg(la, acc_device, rank, world, size, nsamples, steps, dev, tess):
    """Secondary leg (BASELINE config 5's shape by default: 4096 x 4096, 64 AO samples, the example scene
    tessellated to 5.3 M triangles; --ao-size 1024 --ao-tess 0 is config 2): the reference's AO example scene (the 322 triangles
    its own RIB ingest produced, tests/golden/ao_c1.npz), midpoint-tessellated `tess` times,
    size x size pixels, `nsamples` AO rays per primary hit, whole pipeline on the device
    (camera rays, hits, epilogue, AO rays, occlusion, radiance), tiles sharded
    tile_id % world with one gather of tile slabs to rank 0 (strong scaling: the frame is fixed)."""
    import torch
    from lucille_amd import render, scenes
    g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
    acc = la.HipAccel(acc_device)
    ntri = 0
    for k in range(int(g["ngeoms"])):
        P_, I_ = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess)
        acc.add_mesh(P_, I_); ntri += I_.shape[0] // 3
    acc.commit()
    c = g["camera"]
    cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
    # one GPU: the whole frame as one tile (HBM holds it: 460 M rays x 49 B = 22 GB at 4096^2 x 64);
    # sharded: 64 tiles, tile_id % world
    tile = max(64, size // 8) if world > 1 else min(size, 4096)
    times = []; st = None; img = None
    for it in range(steps + 1):
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        if world > 1:
            img, st = render.render_ao_frame_sharded(acc, cam, 1, nsamples, rank, world, tile=tile)
        else:
            img, st = render.render_ao_frame(acc, cam, 1, nsamples, tile=tile)
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        if it > 0:
            times.append(time.perf_counter() - t0)
    rdev = dev if (world == 1 or torch.distributed.get_backend() == "nccl") else torch.device("cpu")
    rays = torch.tensor([st["primary_rays"] + st["ao_rays"]], dtype=torch.float64, device=rdev)
    tmax = torch.tensor([min(times)], dtype=torch.float64, device=rdev)
    if world > 1:
        torch.distributed.all_reduce(rays); torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    acc.close()
    if rank != 0:
        return None
    return {"workload": "examples/ambient_occlusion scene tessellated to %d tris, %dx%d, %d AO samples, frame wall incl. ray gen + tile gather"
                        % (ntri, size, size, nsamples), "triangles": ntri, "tile": tile,
            "rays_per_frame": int(rays.item()), "frame_ms": round(tmax.item() * 1e3, 3),
            "value": round(rays.item() / tmax.item() / 1e6, 1), "unit": "Mrays/s", "scaling": "strong",
            "image_mean": float(img.mean().item())}


def pt_frame_leg(la, acc_device, rank, world, size, spp, dev):
    """Secondary leg (BASELINE config 4): examples/plane_sphere (the 1 986 triangles + vertex normals
    the reference's RIB ingest produced, tests/golden/ao_ps.npz), size x size, spp paths per pixel,
    diffuse wavefront path tracer, tiles sharded tile_id % world + gather of tile slabs to rank 0."""
    import torch
    from lucille_amd import render
    g = np.load(os.path.join(ROOT, "tests", "golden", "ao_ps.npz"))
    acc = la.HipAccel(acc_device)
    for k in range(int(g["ngeoms"])):
        acc.add_mesh(g["pos%d" % k], g["idx%d" % k])
        if ("nrm%d" % k) in g.files:
            acc.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
    acc.commit()
    c = g["camera"]
    cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
    times = []; st = None; img = None
    for it in range(3):
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        pt_tile = size if world == 1 else max(128, size // 4)
        # paths per pass ~ 64 M (about 10 GB of path state): fewer, larger wavefronts -> fewer host syncs per tile
        chunk = max(1, min(spp, (64 << 20) // (pt_tile * pt_tile)))
        img, st = render.render_pt_frame_sharded(acc, cam, spp, rank, world, tile=pt_tile, spp_chunk=chunk,
                                                 kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=8, seed=7)
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        if it > 0:
            times.append(time.perf_counter() - t0)
    rdev = dev if (world == 1 or torch.distributed.get_backend() == "nccl") else torch.device("cpu")
    rays = torch.tensor([float(st["rays"])], dtype=torch.float64, device=rdev)
    tmax = torch.tensor([min(times)], dtype=torch.float64, device=rdev)
    if world > 1:
        torch.distributed.all_reduce(rays); torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    acc.close()
    if rank != 0:
        return None
    return {"workload": "examples/plane_sphere (1986 tris, vertex normals), %dx%d, %d spp, <=8 path vertices, kd 0.8, frame wall incl. ray gen, shading, compaction, tile gather"
                        % (size, size, spp),
            "rays_per_frame": int(rays.item()), "frame_ms": round(tmax.item() * 1e3, 3),
            "value": round(rays.item() / tmax.item() / 1e6, 1), "unit": "Mrays/s", "scaling": "strong",
            "image_mean": float(img.mean().item())}


def cpu_baseline(po, P, idx, org, dr):
    """The reference's CPU path on this box's host cores, bounded sample of the SAME
    workload (first rays of rank 0's batch).  kind "reference": the compiled reference
    itself (oracle/_ref, scalar double, single thread -- its own threading is a racy
    bucket queue that scales 1.36x on 8 cores); else kind "port": the
    bit-identical oracle.  Also reports the port on all host cores."""
    ncores = os.cpu_count() or 1
    out = {}
    if po.ref_available():
        ref = po.RefLib()
        ref.add_mesh(P, idx); ref.build()
        t0 = time.perf_counter(); ref.intersect(org, dr); dt = time.perf_counter() - t0
        out = {"value": round(org.shape[0] / dt / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "reference",
               "sample": "first %d rays of the same S-soup ray dump, ri_raytrace() per ray, %.1f s" % (org.shape[0], dt)}
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    if not out:
        t0 = time.perf_counter(); o.intersect(org, dr, nthreads=1); dt = time.perf_counter() - t0
        out = {"value": round(org.shape[0] / dt / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
               "sample": "first %d rays of the same S-soup ray dump, %.1f s" % (org.shape[0], dt)}
    reps = max(1, min(8, ncores // 8))
    big_o = np.concatenate([org] * reps); big_d = np.concatenate([dr] * reps)
    t0 = time.perf_counter(); o.intersect(big_o, big_d, nthreads=ncores); dt = time.perf_counter() - t0
    out["port_all_cores"] = {"value": round(big_o.shape[0] / dt / 1e6, 3), "unit": "Mrays/s", "cores": ncores,
                             "sample": "%d rays, contiguous slices per thread, %.1f s" % (big_o.shape[0], dt)}
    return out


if __name__ == "__main__":
    main()
