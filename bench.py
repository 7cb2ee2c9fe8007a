#!/usr/bin/env python
"""bench.py -- the driver's benchmark contract for the lucille hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE pass of the hot path (BVH traversal + ray/triangle intersection,
closest-hit records prim/t/u/v) over one synthetic ray dump that is already
resident in HBM.  Workload (BASELINE.json config 3, the one the north-star target
"~1M-tri scene" is quoted on): S-soup-1M = 1,000,000 random triangles
(SURVEY.md Appendix C generator, seed 88172645463325252) and the canonical dump of
--rays incoherent rays (default 100 M) that continues the same stream.

STRONG scaling: the dump is fixed.  Rank r owns the contiguous slice
[n r / N, n (r+1) / N) (it jump-aheads the generator to its first ray) and traces it against
its replica of the BVH.  The hit records (prim u32 + t, u, v f64 = 28 B/ray) stay in the HBM
of the rank that traced them: in lucille a rank's transport stage consumes the records of the
rays it shot and only PIXELS travel to the display owner ("every rank renders, rank 0 owns
the display", render.c:468-514, parallel.c:101-119) -- that exchange is the `ao_render` /
`pt_render` legs' gather of tile slabs.  The path has no per-ray exchange step, so the
headline has no data-path collective; what rank 0 collects is a digest per rank (hits,
sum of t).  value = n x steps / max-over-ranks time.  `with_record_gather` (N > 1) reports
the same dump with every record gathered to rank 0 chunk by chunk behind the tracing of the
next chunk (RCCL, lh_dist_gather) -- what a dump service that returns records to ONE host
would pay -- and `--gather-records` makes that the headline.

One JSON line on rank 0, with
  roofline      the dominant kernel on the headline workload: algorithmic bytes / HIP-event
                duration vs the 8 TB/s peak.  Its hot set (76 MB) lives in L2 + the 256 MiB
                Infinity Cache: `residency` says so -- this is NOT an HBM measurement;
  roofline_hbm  the same kernel on S-soup-10M (0.8 GB hot set, cannot live in the
                Infinity Cache): the HBM figure (N = 1 only);
  ao_render     BASELINE config 5 as stated: the AO example scene tessellated to 21.1 M
                triangles (4.8 GB of trees + triangles), 4096 x 4096, 64 AO samples, tiles sharded
                over the ranks, frame gathered to rank 0 (strong scaling);
  pt_render     BASELINE config 4 (plane_sphere, 2048^2, 256 spp);
  cpu_baseline  the compiled reference / the bit-identical port on this box's host cores.
Every timed launch is validated in-run (`validation`).
"""
import argparse
import ctypes as C
import json
import subprocess
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
B_IN, B_OUT, B_TRI = 32, 16, 40   # SURVEY.md 8(d) algorithmic bytes per ray / per triangle test
B_NODE_SURVEY = 64                # a node visit is priced at 64 B whatever the record
VALU_NODE_STEP, VALU_TRI_STEP = 136, 75      # VALU instructions of one 4-wide node step / one triangle record through the fp32 filter (lh_walk.h, lh_filter.h: counted in the disassembly)
VALU_PEAK_TLANEOPS = 256 * 64 * 2.4e9 / 1e12   # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane operations per second (one wave64 VALU instruction per SIMD every 4 cycles)
B_NODE = {"f32": 64, "q16": 32, "q16x4": 64}   # per node visit: SURVEY's 64-B fp32 2-wide node, 32-B 16-bit grid 2-wide, 64-B 16-bit grid 4-wide
# check values of the canonical S-soup-1M dump on the UNMODIFIED reference
SOUP1M_CHECK = {1_000_000: (821_596, 87998.6606), 2_000_000: (1_644_156, 176110.93)}


def hip_events():
    """HIP events on an explicit stream, straight from libamdhip64 (torch.cuda.Event only
    sees torch's current stream; the kernel is launched on the stream we pass)."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipEventSynchronize.argtypes = [C.c_void_p]
    hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
    return hip


class EventPairs:
    def __init__(self, hip, n):
        self.hip = hip
        self.ev = [(C.c_void_p(), C.c_void_p()) for _ in range(n)]
        for a, b in self.ev:
            hip.hipEventCreate(C.byref(a)); hip.hipEventCreate(C.byref(b))
        self.k = 0

    def begin(self, sptr):
        self.hip.hipEventRecord(self.ev[self.k][0], sptr)

    def end(self, sptr):
        self.hip.hipEventRecord(self.ev[self.k][1], sptr); self.k += 1

    def ms(self):
        out = []
        for a, b in self.ev[:self.k]:
            v = C.c_float(); self.hip.hipEventElapsedTime(C.byref(v), a, b); out.append(v.value)
        return out


def upload_rays(scenes, torch, dev, state, n, keep_first=0):
    """n rays of the stream starting at `state` -> HBM (generated on the host in 10 M-ray pieces)"""
    d_org = torch.empty((n, 3), dtype=torch.float64, device=dev)
    d_dir = torch.empty((n, 3), dtype=torch.float64, device=dev)
    chunk = 10_000_000
    ho = np.empty((min(chunk, max(n, 1)), 3)); hd = np.empty((min(chunk, max(n, 1)), 3))
    first = None
    for b in range(0, n, chunk):
        m = min(chunk, n - b)
        _, _, state = scenes.soup_rays(m, state, ho, hd)
        d_org[b:b + m].copy_(torch.from_numpy(ho[:m])); d_dir[b:b + m].copy_(torch.from_numpy(hd[:m]))
        if b == 0 and keep_first:
            first = (ho[:min(m, keep_first)].copy(), hd[:min(m, keep_first)].copy())
    return d_org, d_dir, first


def record_views(torch, buf, m):
    """SoA views (prim i32, t, u, v f64) over one byte buffer of m * 28 bytes: t | u | v | prim"""
    t = buf[0:8 * m].view(torch.float64); u = buf[8 * m:16 * m].view(torch.float64)
    v = buf[16 * m:24 * m].view(torch.float64); p = buf[24 * m:28 * m].view(torch.int32)
    return (p, t, u, v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rays", type=int, default=100_000_000, help="rays of the dump (all GPUs together)")
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--half-extent", type=float, default=0.005)
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--leg-build", choices=["auto", "host", "device"], default="auto",
                    help="builders of the roofline_hbm and ao_render legs' scenes (auto = lh_accel_commit's own choice; the other builder's tree is timed beside it at N = 1)")
    ap.add_argument("--no-other-builder", action="store_true", help="skip the launches on the other builder's tree (profiling runs: one tree per process)")
    ap.add_argument("--build", choices=["auto", "host", "device"], default="auto",
                    help="builders of the headline leg's scene: auto = lh_accel_commit's own choice (the device builders from 1 M triangles on)")
    ap.add_argument("--mode", choices=["closest", "any"], default="closest")
    ap.add_argument("--chunks", type=int, default=4, help="N>1: trace/gather pipeline depth per rank (with_record_gather)")
    ap.add_argument("--gather-records", action="store_true", help="accepted for compatibility: at N > 1 the gather of every hit record to rank 0 IS inside the headline's timed region (SURVEY 8e)")
    ap.add_argument("--no-gather-records", action="store_true", help="N>1: headline = the records stay with the rank that traced them (a digest travels); the gathered figure moves to `with_record_gather`")
    ap.add_argument("--cpu-rays", type=int, default=1_500_000, help="cpu_baseline sample size")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-hbm", action="store_true", help="skip the S-soup-10M HBM-roofline leg")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the gather microbenchmark (roofline.gather_ceiling)")
    ap.add_argument("--hbm-tris", type=int, default=10_000_000)
    ap.add_argument("--hbm-rays", type=int, default=50_000_000)
    ap.add_argument("--no-ao", action="store_true", help="skip the secondary AO-frame leg")
    ap.add_argument("--no-pt", action="store_true", help="skip the secondary path-traced leg")
    ap.add_argument("--no-config2", action="store_true", help="skip the BASELINE config 2 leg (the AO example RIB, 1024 x 1024, 64 AO samples)")
    ap.add_argument("--pt-size", type=int, default=2048)      # BASELINE config 4: 2048 x 2048, 256 spp
    ap.add_argument("--pt-spp", type=int, default=256)
    ap.add_argument("--backend", default=None, help="accepted for compatibility: torch.distributed is the launcher only (gloo control plane), device data moves through lh_dist_* = RCCL")
    ap.add_argument("--device-override", type=int, default=None,
                    help="testing only: put every rank on this device (2 ranks on a 1-GPU box: the shared-memory transport of lh_dist_*)")
    ap.add_argument("--ao-size", type=int, default=4096)
    ap.add_argument("--ao-samples", type=int, default=64)
    ap.add_argument("--ao-tess", type=int, default=8, help="midpoint-subdivision levels of the example scene (4^n x 322 triangles; 8 -> 21.1 M = BASELINE config 5's '>= 10 M', 7 -> 5.3 M)")
    ap.add_argument("--only", choices=["hbm", "ao", "pt", "config2"], default=None,
                    help="profiling aid: run one secondary leg (the headline shrinks to a 1 M-ray smoke pass)")
    args = ap.parse_args()
    if args.only:
        args.rays = 1_000_000; args.steps = max(1, min(args.steps, 2)); args.no_cpu = True
        args.no_hbm = args.only != "hbm"; args.no_ao = args.only != "ao"; args.no_pt = args.only != "pt"; args.no_config2 = args.only != "config2"

    import torch
    import lucille_amd as la
    from lucille_amd import scenes, shard

    if args.device_override is not None:
        os.environ["LH_DEVICE_OVERRIDE"] = str(args.device_override)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU fallback)")
    # torch.distributed is the launcher (gloo control plane); device data moves through lh_dist_* in the C ABI (RCCL over xGMI)
    rank, world, local = shard.init_process_group(backend=args.backend)
    assert world == args.gpus, "launch with --nproc-per-node == --gpus"
    if args.device_override is not None:
        local = args.device_override
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # ---- synthetic inputs: the scene on every rank, this rank's slice of the dump -> HBM ----
    P, idx, st_after_tris = scenes.soup_triangles(args.tris, args.half_extent)
    n_total = args.rays
    b0, b1 = shard.ray_slice(n_total, rank, world)
    n = b1 - b0
    d_org, d_dir, first = upload_rays(scenes, torch, dev, scenes.skip(st_after_tris, 5 * b0), n,
                                      keep_first=args.cpu_rays if rank == 0 else 0)

    # ONE host build (rank 0), then the flattened scene into every rank's HBM: ncclBroadcast
    acc = la.HipAccel(local)
    info, commit_s, bcast_s = shard.commit_shared(acc, lambda a: a.add_mesh(P, idx), rank, world, build=args.build)     # auto: what lh_accel_commit(accel, 0) does by itself
    device_built = info["nnodes"] == info["nnodes_traversal"]          # a device-built scene has no 2-wide nodes of its own

    mode = la.MODE_CLOSEST if args.mode == "closest" else la.MODE_ANY
    rec_bytes = 28 if mode == la.MODE_CLOSEST else 1

    # chunks of this rank's slice (N = 1: one launch); record buffers; rank 0's gather destination
    nchunks = 1 if world == 1 else max(1, args.chunks)
    per = shard.chunk_capacity(n_total, world, nchunks)     # equal chunk capacity on every rank
    cb = [(c * per, min(n, (c + 1) * per)) for c in range(nchunks)]
    bufs = [torch.empty(per * rec_bytes, dtype=torch.uint8, device=dev) for _ in range(nchunks)]

    def outs_of(c):
        m = cb[c][1] - cb[c][0]
        return record_views(torch, bufs[c], per)[:4] if mode == la.MODE_CLOSEST else (bufs[c][:per],), max(m, 0)

    gathered = None
    if world > 1 and rank == 0:
        gathered = [torch.empty((world, per * rec_bytes), dtype=torch.uint8, device=dev) for _ in range(nchunks)]
    gstream = torch.cuda.Stream(device=dev) if world > 1 else None       # the exchange step's own stream: chunk c on the links while chunk c + 1 is traced

    hip = hip_events()
    stream = torch.cuda.current_stream(dev)
    sptr = C.c_void_p(stream.cuda_stream)
    evp = EventPairs(hip, (args.steps + args.warmup + 2) * nchunks)

    def one_step(timed, gather=True):
        for c in range(nchunks):
            (o, m) = outs_of(c)
            if m > 0:
                if timed:
                    evp.begin(sptr)
                sl = slice(cb[c][0], cb[c][1])
                full = tuple(x[:m] for x in o)
                acc.intersect_device(d_org[sl], d_dir[sl], out=full, mode=mode, variant=args.variant)
                if timed:
                    evp.end(sptr)
            if world > 1 and gather:
                gstream.wait_stream(stream)
                shard.gather_bytes(bufs[c], gathered[c] if rank == 0 else None, stream=gstream)
        if world > 1 and gather:
            stream.wait_stream(gstream)

    def barrier():
        if world > 1:
            shard.barrier()

    one_step(False)                                   # allocations, lazy uploads (untimed)
    torch.cuda.synchronize(dev)

    # ---- algorithmic bytes per ray: counted launch on a sample (untimed) ---------
    ns = min(n, 4_000_000)
    cnt_out, cnt = acc.intersect_device(d_org[:ns], d_dir[:ns], mode=mode, variant=args.variant, counters=True)
    n_nodes = cnt["nodes"] / ns; n_tris = cnt["tris"] / ns
    b_out = B_OUT if mode == la.MODE_CLOSEST else 4
    node_fmt = "f32" if args.variant == 0 else "q16x4"
    b_ray = B_IN + b_out + B_NODE[node_fmt] * n_nodes + B_TRI * n_tris

    # ---- timed region -------------------------------------------------------------
    # ONE exchange step -- the hit-record slices gathered to GPU 0.  At N > 1 that gather is INSIDE the headline's timed
    # region (chunk c on the links while chunk c + 1 is traced), as in rounds 1-2; `records_stay_with_rank` reports the other
    # reading (a rank's transport stage consumes its own records, only a digest travels) beside it, --no-gather-records swaps them.
    head_gather = world == 1 or not args.no_gather_records
    whole = whole_out = None
    if world > 1:
        # no exchange step: a rank's slice is ONE launch into one record buffer (the chunks exist for the gather pipeline)
        whole = torch.empty(max(n, 1) * rec_bytes, dtype=torch.uint8, device=dev)
        whole_out = record_views(torch, whole, max(n, 1))[:4] if mode == la.MODE_CLOSEST else (whole[:max(n, 1)],)

    def stay_step(timed):
        if n > 0:
            if timed:
                evp.begin(sptr)
            acc.intersect_device(d_org[:n], d_dir[:n], out=tuple(x[:n] for x in whole_out), mode=mode, variant=args.variant)
            if timed:
                evp.end(sptr)

    def gather_step(timed):
        one_step(timed, world > 1)
    head_step = gather_step if head_gather else stay_step
    other_step = None if world == 1 else (stay_step if head_gather else gather_step)
    head_step(False)
    for _ in range(args.warmup):
        head_step(False)
    barrier(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        head_step(True)
    torch.cuda.synchronize(dev); barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    # N > 1: the other reading of the exchange step (secondary figure); the gathered pass also fills `gathered` for the validation
    other_elapsed = None
    gather_only_ms = None
    if world > 1:
        gsteps = max(1, min(args.steps, 3))
        other_step(False)
        barrier(); torch.cuda.synchronize(dev)
        tg = time.perf_counter()
        for k in range(gsteps):
            other_step(False)
        torch.cuda.synchronize(dev); barrier()
        other_elapsed = shard.all_reduce_max((time.perf_counter() - tg) / gsteps)
        if not head_gather:
            pass
        # this rank's share of the exchange alone (nothing traced): what the links cost when they are not hidden
        barrier(); torch.cuda.synchronize(dev)
        tg = time.perf_counter()
        for c in range(nchunks):
            shard.gather_bytes(bufs[c], gathered[c] if rank == 0 else None, stream=gstream)
        gstream.synchronize()
        gather_only_ms = (time.perf_counter() - tg) * 1e3
        barrier()
    gather_elapsed = None if world == 1 else (elapsed / args.steps if head_gather else other_elapsed)
    stay_elapsed = None if world == 1 else (other_elapsed if head_gather else elapsed / args.steps)
    # the same dump on the OTHER builder's tree (N = 1; secondary figure: what the choice of builder costs or gains)
    other = None
    if world == 1 and args.build == "auto" and n > 0 and not args.no_other_builder:
        acc_o = la.HipAccel(local); acc_o.add_mesh(P, idx)
        t0o = time.perf_counter(); info_o = acc_o.commit(build="host" if device_built else "device"); commit_o = time.perf_counter() - t0o
        buf_o = torch.empty(max(n, 1) * rec_bytes, dtype=torch.uint8, device=dev)
        out_o = record_views(torch, buf_o, max(n, 1))[:4] if mode == la.MODE_CLOSEST else (buf_o[:max(n, 1)],)
        acc_o.intersect_device(d_org[:n], d_dir[:n], out=tuple(x[:n] for x in out_o), mode=mode, variant=args.variant); torch.cuda.synchronize(dev)
        to = []
        for _ in range(3):
            t0o = time.perf_counter()
            acc_o.intersect_device(d_org[:n], d_dir[:n], out=tuple(x[:n] for x in out_o), mode=mode, variant=args.variant); torch.cuda.synchronize(dev)
            to.append(time.perf_counter() - t0o)
        (o_m, m_m) = outs_of(0)
        same = all(bool(torch.equal(a_[:min(n, m_m)], b_[:min(n, m_m)])) for a_, b_ in zip(out_o, o_m)) if (nchunks == 1 and m_m > 0) else None
        other = {"builder": "host" if device_built else "device", "value": round(n / min(to) / 1e6, 2), "unit": "Mrays/s", "commit_s": round(commit_o, 3),
                 "nodes": info_o["nnodes_traversal"], "depth": info_o["max_depth"], "records_bit_equal": same}
        acc_o.close(); del buf_o, out_o
    kms = evp.ms()
    launches_per_step = sum(1 for c in range(nchunks) if cb[c][1] > cb[c][0]) if head_gather else (1 if n > 0 else 0)
    kernel_ms = float(np.sum(kms)) / max(1, args.steps)            # per step, this rank's launches together

    if world > 1:
        elapsed = shard.all_reduce_max(elapsed)
        if gather_elapsed is not None and head_gather:
            gather_elapsed = elapsed / args.steps
        if stay_elapsed is not None and not head_gather:
            stay_elapsed = elapsed / args.steps
    # ---- per-rank diagnostics (N > 1): every rank says what it did, on stderr and -- collected by rank 0 -- in the line ----
    ranks = None
    if world > 1:
        d_ = shard.dist()
        mine = {"rank": rank, "device": local, "transport": "rccl" if (d_ is not None and d_.transport == la.DIST_RCCL) else "shm",
                "rccl_status": shard.rccl_status(), "rays": int(n), "launches_per_step": int(launches_per_step),
                "kernel_ms_per_step": round(kernel_ms, 3), "gather_only_ms": None if gather_only_ms is None else round(gather_only_ms, 3)}
        print("[bench rank %d] %s" % (rank, json.dumps(mine)), file=sys.stderr, flush=True)
        ranks = shard.all_gather_object(mine)

    # ---- the digest every rank sends instead of its records: hits and sum of t of its slice ------------------
    digest = None
    if mode == la.MODE_CLOSEST:
        if not head_gather:
            hp = whole_out[0][:n] != -1; ht = whole_out[1][:n]
            lh_, lt_ = float(hp.sum().item()), float(ht[hp].sum().item())
        else:
            lh_ = lt_ = 0.0
            for c in range(nchunks):
                (o_, m_) = outs_of(c)
                if m_ > 0:
                    hp = o_[0][:m_] != -1
                    lh_ += float(hp.sum().item()); lt_ += float(o_[1][:m_][hp].sum().item())
        digest = {"hits": int(shard.all_reduce_sum(lh_)) if world > 1 else int(lh_), "sum_t": round(shard.all_reduce_sum(lt_) if world > 1 else lt_, 3)}

    # ---- validation of the timed launches (rank 0) --------------------------------
    validation = None
    if rank == 0:
        if not head_gather:         # the headline's own launch: one buffer; the gathered chunks are checked against the chunked pass's
            validation = validate_dump(torch, la, args, mode, lambda c: (whole_out, n), [(0, n)], cnt_out, ns, gathered, world, per, n_total, bufs, nchunks)
        else:
            validation = validate_dump(torch, la, args, mode, outs_of, cb, cnt_out, ns, gathered, world, per, n_total, bufs, nchunks)

    # ---- context figures (rank 0, N = 1, untimed for `value`) ----------------------
    copy_gbps = host_path = None
    if rank == 0 and world == 1 and not args.no_cpu:
        copy_gbps = copy_rate(torch, dev)
        host_path = host_path_leg(acc, d_org, d_dir, n)

    hbm = None
    if rank == 0 and world == 1 and not args.no_hbm:
        del d_org, d_dir
        torch.cuda.empty_cache()
        hbm = hbm_leg(la, scenes, torch, dev, local, args, hip, sptr, node_fmt)

    ao = None
    if not args.no_ao:
        ao = ao_frame_leg(la, acc_device=local, rank=rank, world=world, size=args.ao_size, nsamples=args.ao_samples,
                          steps=max(2, args.steps), dev=dev, tess=args.ao_tess, build=args.leg_build, twin=not args.no_other_builder)

    c2 = None
    if rank == 0 and world == 1 and not args.no_config2:
        c2 = config2_leg(la, local, dev, max(3, args.steps))

    pt = None
    if not args.no_pt:
        pt = pt_frame_leg(la, acc_device=local, rank=rank, world=world, size=args.pt_size, spp=args.pt_spp, dev=dev)

    if rank == 0:
        value = n_total * args.steps / elapsed / 1e6
        achieved = b_ray * n / (kernel_ms * 1e-3) / 1e9
        traffic = traffic_source = traffic_ms = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc) and world == 1:
            try:
                j = json.load(open(pmc))
                if j.get("rays_per_launch") == n and j.get("mode") == args.mode and j.get("kernel_tag") == node_fmt \
                        and args.variant in (-1, 4):
                    traffic = j.get("hbm_bytes_per_launch"); traffic_ms = j.get("kernel_avg_ms_rocprof")
                    traffic_source = pmc_source("profiles/pmc_latest.json", j)
            except Exception:
                traffic = traffic_source = None
        hot_mb = (info["nnodes_traversal"] * 64 + info["ntriangles"] * 48) / 1e6
        res = {
            "metric": "Mrays/s (primary+AO)", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32 filter + f64 resolve (hit records f64)",
            "data": "synthetic",
            "config": {"workload": "S-soup-1M ray dump (BASELINE config 3): %d random triangles, one dump of %d incoherent rays cut into %d "
                                   "contiguous slice(s), %s-hit%s" % (args.tris, n_total, world, args.mode,
                                                                      "" if world == 1 else (", 28-B hit records gathered to rank 0 inside the timed region" if head_gather
                                                                                             else ", hit records stay with the rank that traced them (digest to rank 0)")),
                       "rays": n_total, "rays_per_gpu": n, "triangles": args.tris, "mode": args.mode,
                       "variant": args.variant, "parallelism": "replicated BVH, ray slices x%d%s" % (world, "" if world == 1 else (", %d-chunk trace/gather pipeline" % nchunks if head_gather else ", one launch per rank, no per-ray exchange")),
                       "scene_load": {"rank0_commit_s": round(commit_s, 3), "broadcast_s": round(bcast_s, 3) if world > 1 else None,
                                      "transport": None if world == 1 else ("rccl" if shard.dist().transport == la.DIST_RCCL else "shm (ranks share a device)"),
                                      "note": "one build on rank 0, flattened arrays broadcast to every rank (lh_dist_broadcast_scene)"},
                       "bvh": {"builder": ("device" if device_built else "host") + (": lh_accel_commit's own choice at this size (the device builders from 1 M triangles on)" if args.build == "auto" else ", asked for"),
                               "other_builder": other,
                               "nodes": info["nnodes_traversal"], "depth": info["max_depth"], "device_bytes": info["device_bytes"],
                               "build_s": round(info["build_seconds"], 3), "ref_tree_build_s": round(info["ref_build_seconds"], 3)}},
            "roofline": {"bound": "l2+mall" if hot_mb < 256.0 else "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_over_algorithmic": None if traffic is None else round(traffic / (b_ray * n), 3),
                         "traffic_frac_of_peak": None if traffic is None or not traffic_ms else round(traffic / (traffic_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "peak_note": "peak = the 8 TB/s HBM datasheet figure, kept as the common denominator; this workload's hot set is cache-resident, "
                                      "so `frac` is NOT a fraction of a bandwidth the data asked of HBM -- the HBM-bound figure is roofline_hbm.frac",
                         "formula": "bytes_per_ray = %d + %d + %d x nodes_per_ray + %d x tris_per_ray (SURVEY 8d); achieved = bytes_per_ray x rays_per_launch / kernel_ms; "
                                    "frac = achieved / peak" % (B_IN, b_out, B_NODE[node_fmt], B_TRI),
                         "residency": "hot set %.0f MB (4-wide nodes + tri32) < 256 MiB Infinity Cache: served by L2 + MALL, "
                                      "NOT an HBM measurement; see roofline_hbm" % hot_mb,
                         "kernel": "k_trace_persist_lane<walk=spec,%s nodes>" % node_fmt if args.variant in (-1, 4) else "k_trace_direct<%s nodes>" % node_fmt,
                         "node_bytes": B_NODE[node_fmt], "launches_per_step": launches_per_step,
                         "kernel_ms": round(kernel_ms, 3), "bytes_per_ray": round(b_ray, 1),
                         "nodes_per_ray": round(n_nodes, 3), "tris_per_ray": round(n_tris, 3)},
            "validation": validation,
        }
        if digest is not None:
            validation["digest_all_ranks"] = dict(digest, hit_rate=round(digest["hits"] / max(1, n_total), 4))
            validation["ok"] = bool(validation["ok"]) and 0.5 < digest["hits"] / max(1, n_total) < 0.99
        if validation is not None and other is not None and other["records_bit_equal"] is not None:
            validation["records_equal_on_the_other_builders_tree"] = other["records_bit_equal"]       # hit records do not depend on the tree
            validation["ok"] = bool(validation["ok"]) and bool(other["records_bit_equal"])
        if world > 1:
            res["exchange"] = {"headline_includes_record_gather": bool(head_gather),
                               "definition": "SURVEY 8e: hit-record slices gathered to GPU 0 (one exchange step); the gather is inside the headline's timed region "
                                             "unless --no-gather-records.  Rounds 1-2 timed it inside, round 3's headline did not: compare N > 1 values across rounds "
                                             "through `with_record_gather` / `records_stay_with_rank`, which every round from 4 on emits side by side"}
            res["with_record_gather"] = {"value": round(n_total / gather_elapsed / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(gather_elapsed * 1e3, 3),
                                         "is_headline": bool(head_gather), "bytes_to_rank0_per_step": int(rec_bytes * (n_total - n)),
                                         "note": "every hit record gathered to rank 0 in %d chunks behind the tracing of the next chunk (lh_dist_gather: RCCL "
                                                 "point-to-point, one xGMI link per peer)" % nchunks}
            res["records_stay_with_rank"] = {"value": round(n_total / stay_elapsed / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(stay_elapsed * 1e3, 3),
                                             "is_headline": not head_gather,
                                             "note": "one launch per rank, the records stay in the HBM of the rank that traced them (its transport stage consumes them), "
                                                     "rank 0 collects a digest per rank (hits, sum of t)"}
            res["ranks"] = ranks
            bad = [r_ for r_ in ranks if r_["transport"] != "rccl"]
            distinct = len(set(r_["device"] for r_ in ranks)) == world
            res["exchange"]["transport_ok"] = not (bad and distinct and os.environ.get("LH_DIST_TRANSPORT") != "shm")
        if world == 1 and not args.no_ceiling:
            # the ceiling that actually binds a cache-resident incoherent walk, measured now: random dependent 64-byte records at
            # this scene's footprint, four workgroups per CU like the walk; `frac_of_gather_ceiling` = the walk's records per second
            # (node visits + triangle records, both one request each) over it
            gc = gather_ceiling(max(16.0, hot_mb), 0)
            res["roofline"]["gather_ceiling"] = gc
            rec_s = (n_nodes + n_tris) * n / (kernel_ms * 1e-3)
            res["roofline"]["records_per_s"] = round(rec_s, 0)
            res["roofline"]["frac_of_gather_ceiling"] = round(rec_s / gc["records_per_s"], 4) if "records_per_s" in gc else None
        if copy_gbps is not None:
            # the box's own device-to-device copy rate next to the 8 TB/s datasheet peak
            res["roofline"]["measured_copy_GBps"] = round(copy_gbps, 1)
            res["roofline"]["frac_of_measured_copy"] = round(achieved / copy_gbps, 4)
        if hbm is not None:
            if copy_gbps is not None:
                hbm["measured_copy_GBps"] = round(copy_gbps, 1)
                hbm["frac_of_measured_copy"] = round(hbm["achieved"] / copy_gbps, 4)
            res["roofline_hbm"] = hbm
        if host_path is not None:
            res["host_path"] = host_path
        if ao is not None:
            res["ao_render"] = ao
        if pt is not None:
            res["pt_render"] = pt
        if c2 is not None:
            res["config2"] = c2
        if not args.no_cpu and world == 1:            # rank 0 at N = 1 only (the contract)
            res["cpu_baseline"] = cpu_baseline(P, idx, first[0], first[1])
            if not res["cpu_baseline"]["kind_ok"]:
                res["validation"]["cpu_baseline_is_the_compiled_reference"] = False
                res["validation"]["ok"] = False
                print("[bench] FAILED: cpu_baseline fell back to the port: oracle/_ref/liblucille_ref.so did not travel with the snapshot "
                      "(__graft_entry__.build() builds it where /root/reference exists; LH_ALLOW_PORT_BASELINE=1 accepts the port)", file=sys.stderr, flush=True)
        print(json.dumps(res), flush=True)
    acc.close()
    rc = 0
    if world > 1:
        # a rank that fell back to the shared-memory transport although every rank has its own device makes the scaling figure
        # meaningless (host staging instead of xGMI): the run fails, loudly, on every rank
        bad_local = 1.0 if (shard.dist().transport != la.DIST_RCCL and args.device_override is None
                            and os.environ.get("LH_DIST_TRANSPORT") != "shm") else 0.0
        if shard.all_reduce_max(bad_local) > 0.5:
            print("[bench rank %d] FAILED: the RCCL transport did not come up on every rank (%s); the figures above were taken over the "
                  "shared-memory fallback" % (rank, shard.rccl_status()), file=sys.stderr, flush=True)
            rc = 3
        shard.barrier()
        shard.dist().close()
        torch.distributed.destroy_process_group()
    if rc:
        sys.exit(rc)


def validate_dump(torch, la, args, mode, outs_of, cb, cnt_out, ns, gathered, world, per, n_total, bufs, nchunks):
    """the timed launches' own outputs: (1) bit-equal to the counted launch on the sample, (2) hits and
    sum(t) of the first 1 M / 2 M rays against the reference's check values for the canonical dump,
    (3) N > 1: the gathered records on rank 0 == the ranks' slices (own slice checked bit for bit,
    every slice by hit-rate bounds)"""
    v = {"ok": True}
    (o, m) = outs_of(0)
    k = min(ns, m)
    if mode == la.MODE_CLOSEST:
        same = all(torch.equal(a[:k], b[:k]) for a, b in zip(o, cnt_out))
        v["timed_equals_counted_launch"] = bool(same); v["ok"] &= bool(same)
        canonical = (args.tris == 1_000_000 and abs(args.half_extent - 0.005) < 1e-12 and args.variant in (-1, 4))
        for nn, (hits, sumt) in SOUP1M_CHECK.items():
            if canonical and m >= nn:
                hit = o[0][:nn] != -1
                h = int(hit.sum().item()); s = float(o[1][:nn][hit].sum().item())
                good = (h == hits) and abs(s - sumt) < 5e-3
                v["first_%dM" % (nn // 1_000_000)] = {"hits": h, "sum_t": round(s, 4), "reference_hits": hits, "reference_sum_t": sumt, "ok": good}
                v["ok"] &= good
        total_hits = int(sum(int((outs_of(c)[0][0][:outs_of(c)[1]] != -1).sum().item()) for c in range(len(cb))))
        v["hits_this_rank"] = total_hits
    else:
        same = torch.equal(o[0][:k], cnt_out[0][:k])
        v["timed_equals_counted_launch"] = bool(same); v["ok"] &= bool(same)
    if world > 1:
        ok = True
        for c in range(nchunks):
            ok &= bool(torch.equal(gathered[c][0], bufs[c]))
            if mode == la.MODE_CLOSEST:
                for r in range(world):
                    p = gathered[c][r][24 * per:28 * per].view(torch.int32)
                    frac = float((p != -1).float().mean().item())
                    ok &= (0.5 < frac < 0.99)
        v["gathered_records_ok"] = ok; v["ok"] &= ok
    return v


def copy_rate(torch, dev):
    big = torch.empty(1 << 30, dtype=torch.uint8, device=dev); dst = torch.empty_like(big)
    dst.copy_(big); torch.cuda.synchronize(dev)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dst.copy_(big)
    e1.record(); torch.cuda.synchronize(dev)
    return 10 * 2 * big.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9       # read + write


def host_path_leg(acc, d_org, d_dir, n):
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
    return {"value": round(nh / best / 1e6, 1), "unit": "Mrays/s", "link_GBps": round(nh * 76 / best / 1e9, 1),
            "sample": "%d rays through lh_accel_intersect_host: pageable host arrays -> pinned staging in 2 M-ray chunks on two "
                      "streams, 48 B/ray up + 28 B/ray down over PCIe; never the headline value" % nh}


def gather_ceiling(mb, mode, lds=40000, steps=200):
    """SURVEY 8d / VERDICT r04 item 4: what binds the incoherent walk is not the HBM datasheet figure but the memory system's rate
    for DEPENDENT random records -- tools/ubench/gather (one chain per lane, every link perturbed by the chain's own running
    sum), run here, on this box, at this leg's footprint and occupancy: mode 0 = 64-byte records (a 4-wide node), mode 5 =
    128-byte records (an 8-wide node).  -> {"records_per_s": G/s, ...} or {"error": ...}"""
    exe = os.path.join(ROOT, "tools", "ubench", "gather")
    cmd = [exe, str(int(mb)), str(steps), "4096", str(mode)]
    try:
        env = dict(os.environ, GATHER_LDS=str(lds))
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=env).stdout
        line = [l for l in out.splitlines() if l.startswith("array") and ("mode %d:" % mode) in l][-1]
        g = float(line.split("ms")[1].split("G chain-steps/s")[0])
        return {"records_per_s": round(g * 1e9, 0), "record_bytes": 64 if mode == 0 else 128, "footprint_MB": int(mb), "blocks_per_cu": int(160 * 1024 // lds),
                "cmd": "GATHER_LDS=%d tools/ubench/gather %d %d 4096 %d" % (lds, int(mb), steps, mode), "raw": line.strip(),
                "what": "dependent random gather, one chain per lane, 256 CUs x %d workgroups; the incoherent walk's binding ceiling (DESIGN 3.3)" % int(160 * 1024 // lds)}
    except Exception as e:                                  # noqa: BLE001 -- context, the leg stands without it
        return {"error": repr(e), "cmd": " ".join(cmd)}


def pmc_source(path, j):
    """where a `traffic` figure comes from: it is NOT measured inside this run (rocprofv3 counter passes re-run the whole
    command: tools/profile_round2.sh), it is the committed summary of the same command's last counter passes"""
    return {"file": path, "round": j.get("round"), "commit": j.get("commit"), "raw": j.get("source"),
            "FETCH_SIZE_KiB": j.get("FETCH_SIZE_KiB"), "WRITE_SIZE_KiB": j.get("WRITE_SIZE_KiB"),
            "formula": "2 x FETCH_SIZE x 1024 (gfx950: 128-B fabric requests tallied as 64 B) + WRITE_SIZE x 1024",
            "kernel_avg_ms_in_that_run": j.get("kernel_avg_ms_rocprof")}


def hbm_leg(la, scenes, torch, dev, local, args, hip, sptr, node_fmt):
    """the HBM roofline: the same closest-hit kernel on S-soup-10M (10 M triangles, half-extent 0.002: the
    SURVEY's config-5 stress soup).  Hot set = 4-wide nodes + tri32 ~ 0.8 GB >> the 256 MiB Infinity Cache."""
    P, idx, st = scenes.soup_triangles(args.hbm_tris, 0.002)
    n = args.hbm_rays
    d_org, d_dir, _ = upload_rays(scenes, torch, dev, st, n)
    acc = la.HipAccel(local); acc.add_mesh(P, idx)
    t0 = time.perf_counter(); info = acc.commit(build=args.leg_build); commit1 = time.perf_counter() - t0
    dev_built = info["nnodes"] == info["nnodes_traversal"]          # a device-built scene has no 2-wide nodes of its own
    other_b = "host" if dev_built else "device"
    del P, idx
    out = acc.intersect_device(d_org, d_dir); torch.cuda.synchronize(dev)
    ns = min(n, 4_000_000)
    cnt_out, cnt = acc.intersect_device(d_org[:ns], d_dir[:ns], counters=True)
    n_nodes = cnt["nodes"] / ns; n_tris = cnt["tris"] / ns
    node_bytes = acc.dump_node_bytes()              # 128: the 8-wide nodes (hot set beyond the Infinity Cache), else the 4-wide node's 64
    if node_bytes == 128:
        node_fmt = "q16x8"
    # SURVEY 8d prices EVERY node visit at 64 B (B_node), whatever record the walk really fetches: that is `bytes_per_ray`,
    # `achieved` and `frac` below.  The 8-wide walk fetches one 128-byte record per visit (and makes fewer visits); its record
    # bytes are reported as a plain number (`record_bytes_per_ray`), not as a bandwidth -- part of them is served by caches.
    b_ray = B_IN + B_OUT + B_NODE_SURVEY * n_nodes + B_TRI * n_tris
    # the same sample through the 4-wide walk: hit records do not depend on the tree
    cross = None
    if node_bytes == 128:
        acc.set_param("wide8", 0)
        alt = acc.intersect_device(d_org[:ns], d_dir[:ns]); torch.cuda.synchronize(dev)
        cross = all(torch.equal(a, b) for a, b in zip(alt, cnt_out))
        acc.set_param("wide8", -1)
        del alt
    steps = 3

    def timed(a, o):
        ev = EventPairs(hip, steps)
        a.intersect_device(d_org, d_dir, out=o); torch.cuda.synchronize(dev)
        for _ in range(steps):
            ev.begin(sptr); a.intersect_device(d_org, d_dir, out=o); ev.end(sptr)
        torch.cuda.synchronize(dev)
        return float(np.mean(ev.ms()))
    ms = timed(acc, out)
    ok = all(torch.equal(a[:ns], b) for a, b in zip(out, cnt_out))
    hit = float((out[0] != -1).float().mean().item())
    achieved = b_ray * n / (ms * 1e-3) / 1e9
    traffic = traffic_source = traffic_ms = None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest_hbm.json")
    if os.path.exists(pmc):
        try:
            j = json.load(open(pmc))
            if j.get("rays_per_launch") == n and j.get("triangles") == args.hbm_tris and j.get("kernel_tag") == node_fmt:
                traffic = j.get("hbm_bytes_per_launch"); traffic_ms = j.get("kernel_avg_ms_rocprof")
                traffic_source = pmc_source("profiles/pmc_latest_hbm.json", j)
        except Exception:
            traffic = traffic_source = None
    info = acc.info()
    hot = info["nnodes_traversal"] * 64 + info["ntriangles"] * 48
    hot8 = (info["nnodes_traversal"] * 128 * 3 // 7 if node_bytes == 128 else info["nnodes_traversal"] * 64) + info["ntriangles"] * 48       # an 8-wide tree has ~3/7 of the 4-wide tree's nodes
    acc.close()
    # the ceiling of THIS leg's access pattern, measured now: dependent random records of the size the walk fetches, at the
    # scene's footprint (HBM-resident), at the walk's occupancy (three workgroups per CU for the 8-wide walk, four for the 4-wide)
    gc = None if args.no_ceiling else gather_ceiling(min(4096.0, hot8 / 1e6), 5 if node_bytes == 128 else 0, lds=53000 if node_bytes == 128 else 40000)
    # the twin on the OTHER builder's tree: same rays, same records
    twin = None
    try:
        if args.no_other_builder:
            raise RuntimeError("skipped (--no-other-builder)")
        P2, idx2, _ = scenes.soup_triangles(args.hbm_tris, 0.002)
        acc2 = la.HipAccel(local); acc2.add_mesh(P2, idx2)
        t0 = time.perf_counter(); info2 = acc2.commit(build=other_b); commit2 = time.perf_counter() - t0
        del P2, idx2
        out2 = acc2.intersect_device(d_org, d_dir); torch.cuda.synchronize(dev)
        _, cnt2 = acc2.intersect_device(d_org[:ns], d_dir[:ns], counters=True)
        ms2 = timed(acc2, out2)
        same2 = all(bool(torch.equal(a, b)) for a, b in zip(out2, out))
        nn2 = cnt2["nodes"] / ns; nt2 = cnt2["tris"] / ns
        br2 = B_IN + B_OUT + B_NODE_SURVEY * nn2 + B_TRI * nt2
        twin = {"builder": other_b, "commit_s": round(commit2, 3), "kernel_ms": round(ms2, 3),
                "value": round(n / (ms2 * 1e-3) / 1e6, 1), "value_unit": "Mrays/s", "nodes_per_ray": round(nn2, 3), "tris_per_ray": round(nt2, 3),
                "bytes_per_ray": round(br2, 1), "achieved": round(br2 * n / (ms2 * 1e-3) / 1e9, 1),
                "frac": round(br2 * n / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "records_bit_equal": same2,
                "nodes": info2["nnodes_traversal"], "depth": info2["max_depth"]}
        ok = ok and same2
        acc2.close(); del out2
    except Exception as e:                                  # noqa: BLE001 -- the twin is context, the leg stands without it
        twin = {"error": repr(e)}
    return {"workload": "S-soup-10M ray dump: %d random triangles (half-extent 0.002), %d incoherent rays, closest-hit" % (args.hbm_tris, n),
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": traffic, "traffic_source": traffic_source,
            "traffic_frac_of_peak": None if traffic is None else round(traffic / ((traffic_ms or ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "traffic_frac_note": "bytes AND time of the same profiled run (%s ms per launch under the counters; this run: %.3f ms)" % (traffic_ms, ms),
            "gather_ceiling": gc, "records_per_s": round((n_nodes + n_tris) * n / (ms * 1e-3), 0),
            "frac_of_gather_ceiling": None if not gc or "records_per_s" not in gc else round((n_nodes + n_tris) * n / (ms * 1e-3) / gc["records_per_s"], 4),
            "traffic_over_algorithmic": None if traffic is None else round(traffic / (b_ray * n), 3),
            "formula": "bytes_per_ray = %d (ray in) + %d (hit record out) + %d x nodes_per_ray + %d x tris_per_ray -- SURVEY 8d's constants "
                       "(B_in, B_out, B_node, B_tri), whatever the walk really moves (this kernel reads 48 B of fp64 ray and writes a 28-B record per ray, "
                       "and an 8-wide visit fetches a 128-B record); achieved = bytes_per_ray x rays / kernel_ms; frac = achieved / peak; "
                       "traffic = 2 x FETCH_SIZE + WRITE_SIZE of the committed counter pass; traffic_over_algorithmic = traffic / (bytes_per_ray x rays)"
                       % (B_IN, B_OUT, B_NODE_SURVEY, B_TRI),
            "record_bytes_per_ray": round(B_IN + B_OUT + node_bytes * n_nodes + B_TRI * n_tris, 1),
            "record_bytes_note": "what the walk's own records add up to per ray (%d-B node records): includes bytes served by L2 / the Infinity Cache -- a count, not a bandwidth" % node_bytes,
            "residency": "hot set %.0f MB as 4-wide nodes + tri32 >> 256 MiB Infinity Cache: HBM" % (hot / 1e6),
            "builder": ("device" if dev_built else "host") + (": lh_accel_commit's own choice at this size" if args.leg_build == "auto" else ", asked for")
                       + "; `other_builder` is the same dump on the other builder's tree",
            "commit_s": round(commit1, 3), "other_builder": twin,
            "kernel": "k_trace_persist_lane<walk=spec8, q16x8 nodes: 128-byte 8-wide records, one cache line each>" if node_bytes == 128
                      else "k_trace_persist_lane<walk=spec,%s nodes>" % node_fmt,
            "node_bytes": node_bytes,
            "value": round(n / (ms * 1e-3) / 1e6, 1), "value_unit": "Mrays/s", "kernel_ms": round(ms, 3),
            "bytes_per_ray": round(b_ray, 1), "nodes_per_ray": round(n_nodes, 3), "tris_per_ray": round(n_tris, 3),
            "hit_rate": round(hit, 4), "device_bytes": info["device_bytes"],
            "build_s": round(info["build_seconds"], 3), "ref_tree_build_s": round(info["ref_build_seconds"], 3),
            "validation": {"timed_equals_counted_launch": bool(ok), "equals_4wide_walk_on_sample": cross,
                           "ok": bool(ok) and cross is not False and 0.5 < hit < 0.999}}


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
        # The frame is NOT bandwidth-bound (coherent rays: 0.15 KB of fabric traffic per ray).  With four workgroups per CU (round 4) the
        # fused any-hit kernel is bound by VALU ISSUE: profiles/r04_pmc_ao_dense.txt -- SQ_INSTS_VALU 2.96e10 wave instructions x 4
        # cycles / 1024 SIMDs = 1.16e8 of the launch's 1.21e8 cycles: the vector pipes are busy 96 % of the time, at 74 % lane use.
        # `achieved` / `peak` are therefore VALU lane operations per second: what the walk's own steps need -- 136 per node step
        # (lh_walk.h slab_w: the disassembly's count), 75 per triangle record through the fp32 filter (lh_filter.h) -- x the counted
        # steps of the frame, against 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz.  What `frac` leaves out is what the counters show the
        # pipes are busy WITH besides: idle lanes (26 %), the refill of finished lanes (ray generation + set-up, ~300 instructions a
        # regroup), the fp64 resolves.  The record rate is kept beside it (`records_per_s`): round 1's gather microbenchmark
        # (129 G random 64-B records/s) is no ceiling for these rays -- the 64 rays of a hemisphere share their first ten levels.
        recs = c["nodes"] + c["tris"]
        b_frame = 64.0 * c["nodes"] + 40.0 * c["tris"] + (48.0 + 28.0) * st["primary_rays"] + 4.0 * st["primary_hits"]
        lane_ops = VALU_NODE_STEP * c["nodes"] + VALU_TRI_STEP * c["tris"]
        sq = {"source": "profiles/r04_pmc_ao_dense.txt (tools/pmc_cmd.sh: separate SQ / TCC passes of this frame, the fused any-hit launch)",
              "valu_busy": 0.96, "valu_lane_use": 0.74, "wave_cycles_waiting": 0.49, "l2_hit_rate": 0.52,
              "valu_wave_instructions_per_ray": 66.8, "fabric_read_bytes_per_frame": 68.6e9}
        roof = {"bound": "valu issue", "achieved": round(lane_ops / min(times) / 1e12, 2), "peak": VALU_PEAK_TLANEOPS, "unit": "T lane-ops/s",
                "frac": round(lane_ops / min(times) / 1e12 / VALU_PEAK_TLANEOPS, 4), "traffic": sq["fabric_read_bytes_per_frame"],
                "formula": "achieved = (%d x node steps + %d x triangle records of the counted frame) / frame time; peak = 256 CUs x 64 lanes x 2.4 GHz"
                           % (VALU_NODE_STEP, VALU_TRI_STEP),
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
            "tile": tile if tile is not None else "%d full-width bands of %d rows dealt out to %d ranks in serpentine order (shard.bands_of_rank), one device batch per rank, one float per pixel gathered" % (
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


def config2_leg(la, acc_device, dev, steps, size=1024, gather=64):
    """BASELINE config 2 as stated: the reference's examples/ambient_occlusion.rib (tests/golden/rib/: 322 triangles, its own
    PixelSamples 3 3), 1024 x 1024, 64 AO samples, one GPU.  RIB reader -> accelerator -> one frame; timed: the frame with the
    image left in HBM (`frame_ms`) and through lh_render_ao_frame_host, the call lsh_hip makes (`frame_host_ms`: + the 12.6 MB
    image over PCIe).  tests/test_gpu_config2.py holds the parity side (camera-ray hits against the oracle, tiling, the driver)."""
    import torch
    from lucille_amd import render, rib
    t0 = time.perf_counter()
    sc = rib.RibScene(os.path.join(ROOT, "tests", "golden", "rib", "ambient_occlusion.rib"))
    parse_s = time.perf_counter() - t0
    acc = la.HipAccel(acc_device); sc.add_to(acc)
    t0 = time.perf_counter(); info = acc.commit(); commit_s = time.perf_counter() - t0
    ps = int(sc.info.pixel_samples[0])
    cam = la.Camera.make(size, size, sc.camera.flength, list(sc.camera.cam2world), sc.camera.rh)
    times = []; host_times = []; stats = []
    for it in range(steps + 1):
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        img, st = render.render_ao_frame(acc, cam, ps, gather, tile=size)
        torch.cuda.synchronize(dev)
        if it:
            times.append(time.perf_counter() - t0)
        stats.append(dict(st))
        t0 = time.perf_counter()
        himg, hst = acc.render_ao_frame_host(cam, ps, gather)
        if it:
            host_times.append(time.perf_counter() - t0)
    img2, st2 = render.render_ao_frame(acc, cam, ps, gather, tile=160)
    ok = all(s == stats[0] for s in stats) and st2 == stats[0] and bool(torch.equal(img, img2)) \
        and bool(np.array_equal(np.asarray(himg).reshape(size, size, 3), img.cpu().numpy()))
    rays = st["primary_rays"] + st["ao_rays"]
    acc.close(); sc.close()
    return {"workload": "BASELINE config 2: examples/ambient_occlusion.rib, %d triangles, %dx%d, PixelSamples %d %d, %d AO samples, one GPU"
                        % (info["ntriangles"], size, size, ps, ps, gather),
            "rib_parse_s": round(parse_s, 4), "commit_s": round(commit_s, 4), "rays_per_frame": int(rays),
            "primary_rays": int(st["primary_rays"]), "primary_hits": int(st["primary_hits"]), "ao_rays": int(st["ao_rays"]),
            "frame_ms": round(min(times) * 1e3, 3), "frame_host_ms": round(min(host_times) * 1e3, 3),
            "value": round(rays / min(times) / 1e6, 1), "unit": "Mrays/s",
            "validation": {"frames_repeat_and_retiled_bit_equal_and_host_call_equal": ok, "ok": ok}}


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
    # paths per pass (decided once: the first frame's buffers stay allocated): as many as 70 % of the free HBM holds (176 B of
    # path state each; a 2048^2 x 256 spp frame is 2^30 paths = 189 GB of the 288): every pass costs one kernel ramp + drain per
    # bounce, so fewer, larger wavefronts are faster (tools/experiments/pt_frames.py: 166.3 / 154.7 / 148.6 ms per frame as 4 / 2 / 1 passes;
    # the image does not change by a bit)
    torch.cuda.empty_cache()
    free_b = torch.cuda.mem_get_info(dev)[0]
    per_pass = max(64 << 20, min(1 << 30, int(free_b * 7 // 10 // 176)))
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
                        "bounce one shading pass (decide + compact + scatter, ray counts stay on the device) and one closest-hit launch: 15 "
                        "launches, no host round trip; kernel time = frame time, closest-hit kernels 67 % of it (profiles/r03_pt_kernel_stats.csv)"}
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
                      "there is nothing to run it against (SURVEY 8f-3; DESIGN.md 11 lists the departures from its text)",
            # white furnace with albedo 0.8 under a unit environment: every pixel's radiance lies in (0, 1]
            "validation": {"frames_repeat": repeat, "retiled_frame_bit_equal": retiled,
                           "radiance_in_0_1": bool(float(img.min().item()) >= 0.0 and float(img.max().item()) <= 1.0 + 1e-6),
                           "ok": repeat and retiled is not False and 0.0 < float(img.mean().item()) <= 1.0}}


def host_cores():
    """what this process may really use: os.cpu_count() is the box, the affinity mask and the cgroup CPU quota are the share"""
    n_os = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:                                            # noqa: BLE001
        aff = n_os
    quota = None
    try:                                                         # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:                                            # noqa: BLE001
        try:                                                     # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:                                        # noqa: BLE001
            quota = None
    eff = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return {"os_cpu_count": n_os, "sched_affinity": aff, "cgroup_cpu_quota": None if quota is None else round(quota, 2), "effective": eff}


def cpu_baseline(P, idx, org, dr):
    """The reference's CPU path on this box's host cores, bounded sample of the SAME
    workload (first rays of the dump).  kind "reference": the compiled reference
    itself (oracle/_ref, scalar double, single thread -- its own threading is a racy
    bucket queue that scales 1.36x on 8 cores); else kind "port": the
    bit-identical oracle.  Also reports the port on the cores this process may use (affinity mask and
    cgroup quota, not os.cpu_count()) with the speed-up over one thread.  The only place bench.py
    touches oracle/: the checker timed as the CPU baseline."""
    from oracle import pyoracle as po
    hc = host_cores()
    ncores = hc["effective"]
    out = {}
    if po.ref_available():
        ref = po.RefLib()
        ref.add_mesh(P, idx); ref.build()
        # three thirds of the sample, timed one after the other: the median, and the spread between them (r04: one un-repeated
        # sample read 0.128 and 0.156 Mrays/s on two boxes)
        m = org.shape[0] // 3; rates = []; dt = 0.0
        for k in range(3):
            t0 = time.perf_counter(); ref.intersect(org[k * m:(k + 1) * m], dr[k * m:(k + 1) * m]); d_ = time.perf_counter() - t0
            rates.append(m / d_ / 1e6); dt += d_
        rates.sort()
        out = {"value": round(rates[1], 4), "unit": "Mrays/s", "cores": 1, "kind": "reference",
               "repeats": [round(r_, 4) for r_ in rates], "spread": round((rates[2] - rates[0]) / rates[1], 3),
               "sample": "the first %d rays of the same S-soup ray dump in three parts of %d, ri_raytrace() per ray, %.1f s; value = the median part" % (3 * m, m, dt)}
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    sub = min(org.shape[0], 300_000)
    t0 = time.perf_counter(); o.intersect(org[:sub], dr[:sub], nthreads=1); dt1 = time.perf_counter() - t0
    one = sub / dt1 / 1e6
    if not out:
        out = {"value": round(one, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
               "sample": "first %d rays of the same S-soup ray dump, %.1f s" % (sub, dt1)}
    out["host"] = hc
    # the compiled reference (oracle/_ref: built by __graft_entry__.build() where /root/reference exists, shipped to the GPU box with
    # the snapshot) is what this leg is expected to time: a run that silently fell back to the port says so and turns the line red
    out["expected_kind"] = "port" if os.environ.get("LH_ALLOW_PORT_BASELINE") == "1" else "reference"
    out["kind_ok"] = out["kind"] == out["expected_kind"] or out["kind"] == "reference"
    curve = []
    for nt in sorted(set(t for t in (8, 32, ncores) if t <= ncores)):
        reps = max(1, min(8, nt // 8))
        big_o = np.concatenate([org] * reps); big_d = np.concatenate([dr] * reps)
        t0 = time.perf_counter(); o.intersect(big_o, big_d, nthreads=nt); dt = time.perf_counter() - t0
        curve.append({"threads": nt, "value": round(big_o.shape[0] / dt / 1e6, 3), "speedup_over_one_thread": round(big_o.shape[0] / dt / 1e6 / one, 1),
                      "rays": int(big_o.shape[0]), "seconds": round(dt, 1)})
    best = max(curve, key=lambda c: c["value"]) if curve else None
    if best is not None:
        out["port_all_cores"] = {"value": best["value"], "unit": "Mrays/s", "cores": best["threads"],
                                 "speedup_over_one_thread": best["speedup_over_one_thread"], "port_one_thread": round(one, 4),
                                 "thread_curve": curve,
                                 "sample": "%d rays, contiguous slices per thread, %.1f s" % (best["rays"], best["seconds"]),
                                 "note": "the port walks 240-byte pointer-linked nodes (the reference's layout): one dependent cache miss per step, so it scales with "
                                         "memory-level parallelism, not with cores -- `cores` is the thread count of the best point of the curve, `host` what the "
                                         "process is allowed to use"}
    return out


if __name__ == "__main__":
    main()
