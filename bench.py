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
its replica of the BVH.  The path has no per-ray exchange; its ONE exchange step is
the gather of the hit records to rank 0 (lh_dist_gather: RCCL point-to-point, one xGMI link per
peer), and at N > 1 that gather IS INSIDE the headline's timed region: a rank's slice is traced
in --chunks chunks, chunk c travels while chunk c + 1 is traced.  On the wire a record is 16
bytes (--record-bytes 16, the default: prim u32 + t, u, v rounded to fp32 from the fp64 bits --
6e-8 relative, north_star allows 1e-5; lh_dist_pack_records16); the fp64 records (28 bytes:
prim u32 + t, u, v f64) stay in the HBM of the rank that traced them -- in lucille a rank's
transport stage consumes the records of the rays it shot and only PIXELS travel to the display
owner ("every rank renders, rank 0 owns the display", render.c:468-514, parallel.c:101-119).
At 28 bytes a ray eight ranks are bound by rank 0's links (350 MB per peer against 5.4 ms of
tracing: profiles/r06_dump_cost_table.md); `record_gather_fp64` reports that reading beside
the headline, `records_stay_with_rank` the one with no record exchange at all (a digest per rank
travels: hits, sum of t), and --no-gather-records makes the latter the headline.
value = n x steps / max-over-ranks time.

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
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlegs.common import *  # noqa: E402,F401,F403 -- constants (HBM_PEAK_GBPS, B_IN ...), hip_events, EventPairs, upload_rays, record_views
from benchlegs.common import gather_ceiling, pmc_source, copy_rate, host_cores  # noqa: E402
from benchlegs.dump import validate_dump, gathered_vs_world1  # noqa: E402
from benchlegs.hostpath import host_path_leg  # noqa: E402
from benchlegs.hbm import hbm_leg  # noqa: E402
from benchlegs.ao import ao_frame_leg  # noqa: E402
from benchlegs.config2 import config2_leg  # noqa: E402
from benchlegs.pt import pt_frame_leg  # noqa: E402
from benchlegs.cpu import cpu_baseline  # noqa: E402


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
    ap.add_argument("--chunks", type=int, default=0, help="N>1: trace/gather pipeline depth per rank (with_record_gather); 0 = 16 // N (8 / 4 / 2 chunks at 2 / 4 / 8 ranks: "
                    "a chunk costs its launch's ramp and drain, ~0.6 ms for incoherent rays, profiles/r06_dump_cost_table.md)")
    ap.add_argument("--comm-wgs", type=int, default=32,
                    help="N>1: persistent workgroups the headline's launches leave out of their grid (of CUs x 4) so that the exchange's RCCL kernels find room "
                         "beside them: a persistent launch holds every wave slot of the chip until its cursors run dry, and a send / receive kernel that cannot "
                         "start until then would put the gather of chunk c BEHIND the tracing of chunk c + 1 instead of beside it")
    ap.add_argument("--record-bytes", type=int, choices=[16, 28], default=16,
                    help="N>1: bytes of a hit record on the wire -- 16: prim u32 + t, u, v fp32 (rounded from the fp64 records, which stay with the rank that traced them); 28: the fp64 records themselves")
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
    from lucille_amd import binding, scenes, shard

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

    if world > 1 and args.comm_wgs > 0:
        full_grid = torch.cuda.get_device_properties(dev).multi_processor_count * 4
        acc.set_param("grid", max(full_grid // 2, full_grid - args.comm_wgs))
    mode = la.MODE_CLOSEST if args.mode == "closest" else la.MODE_ANY
    rec_bytes = 28 if mode == la.MODE_CLOSEST else 1

    # chunks of this rank's slice (N = 1: one launch); record buffers; rank 0's gather destination
    nchunks = 1 if world == 1 else (max(1, args.chunks) if args.chunks > 0 else max(1, 16 // world))
    per = shard.chunk_capacity(n_total, world, nchunks)     # equal chunk capacity on every rank
    cb = [(c * per, min(n, (c + 1) * per)) for c in range(nchunks)]
    bufs = [torch.empty(per * rec_bytes, dtype=torch.uint8, device=dev) for _ in range(nchunks)]

    def outs_of(c):
        m = cb[c][1] - cb[c][0]
        return record_views(torch, bufs[c], per)[:4] if mode == la.MODE_CLOSEST else (bufs[c][:per],), max(m, 0)

    # what travels: the fp64 records themselves (28 B), or 16-byte wire records packed behind every chunk's launch (closest hit only)
    wire16 = world > 1 and mode == la.MODE_CLOSEST and args.record_bytes == 16
    wire_bytes = 16 if wire16 else rec_bytes
    wire = [torch.empty(per * 16, dtype=torch.uint8, device=dev) for _ in range(nchunks)] if wire16 else bufs
    gathered = None
    if world > 1 and rank == 0:
        gathered = [torch.empty((world, per * wire_bytes), dtype=torch.uint8, device=dev) for _ in range(nchunks)]
    # the exchange step's own stream: chunk c on the links while chunk c + 1 is traced.  HIGH priority = a hardware queue from another pool than
    # the trace stream's: the runtime maps the streams of one priority onto four in-order hardware queues, and two streams on one queue run one
    # after the other whatever the events say (seen on the host batch path: profiles/r06_hostpath.txt)
    gstream = torch.cuda.Stream(device=dev, priority=-1) if world > 1 else None

    hip = hip_events()
    stream = torch.cuda.current_stream(dev)
    sptr = C.c_void_p(stream.cuda_stream)
    evp = EventPairs(hip, (args.steps + args.warmup + 2) * nchunks)

    # N > 1: the chunks of a slice are launched one behind the other on ONE stream.  (Round 6 alternated them between two trace streams, the
    # first of higher priority, so that chunk c + 1's workgroups would take the CUs chunk c's last waves leave -- a persistent launch ends as
    # slowly as its last rays walk, ~0.6 ms per 1.5 M-ray chunk.  The two launches ran side by side instead and lost: a 12.5 M-ray slice in
    # 2 / 4 chunks 6.7 / 8.1 ms on one stream, 7.8 / 9.3 on two; a 50 M-ray slice in 2 chunks 23.2 -> 27.8 ms.  profiles/r06_dump_cost_table.md)
    tstreams = [stream, stream] if world > 1 else None

    def one_step(timed, gather=True, wire_=None, dst_=None):
        wire_ = wire if wire_ is None else wire_; dst_ = gathered if dst_ is None else dst_
        for c in range(nchunks):
            (o, m) = outs_of(c)
            ts = tstreams[c % 2] if world > 1 else stream
            tp = C.c_void_p(ts.cuda_stream)
            if m > 0:
                if timed:
                    evp.begin(tp)
                sl = slice(cb[c][0], cb[c][1])
                full = tuple(x[:m] for x in o)
                acc.intersect_device(d_org[sl], d_dir[sl], out=full, mode=mode, variant=args.variant, stream=ts.cuda_stream)
                if timed:
                    evp.end(tp)
            if world > 1 and gather:
                if wire_ is not bufs:              # the chunk's wire records, behind its launch on its trace stream (every slot: equal sizes on every rank)
                    binding.pack_records16(o[0], o[1], o[2], o[3], wire_[c], n=per, stream=ts)
                gstream.wait_stream(ts)
                shard.gather_bytes(wire_[c], dst_[c] if rank == 0 else None, stream=gstream)
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
        # the exchange with the fp64 records themselves on the wire (28 B a ray): the reading of rounds 1-5
        fp64_elapsed = None
        if wire16:
            g28 = [torch.empty((world, per * rec_bytes), dtype=torch.uint8, device=dev) for _ in range(nchunks)] if rank == 0 else None
            one_step(False, True, bufs, g28)
            barrier(); torch.cuda.synchronize(dev)
            tg = time.perf_counter()
            for k in range(gsteps):
                one_step(False, True, bufs, g28)
            torch.cuda.synchronize(dev); barrier()
            fp64_elapsed = shard.all_reduce_max((time.perf_counter() - tg) / gsteps)
            if rank == 0:
                fp64_gather_ok = all(bool(torch.equal(g28[c][0], bufs[c])) for c in range(nchunks))
            del g28
        # this rank's share of the exchange alone (nothing traced): what the links cost when they are not hidden
        barrier(); torch.cuda.synchronize(dev)
        tg = time.perf_counter()
        for c in range(nchunks):
            shard.gather_bytes(wire[c], gathered[c] if rank == 0 else None, stream=gstream)
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
            validation = validate_dump(torch, la, args, mode, lambda c: (whole_out, n), [(0, n)], cnt_out, ns, gathered, world, per, n_total, bufs, nchunks, wire_bytes)
        else:
            validation = validate_dump(torch, la, args, mode, outs_of, cb, cnt_out, ns, gathered, world, per, n_total, bufs, nchunks, wire_bytes)
        if world > 1:
            # every gathered slice against rank 0's OWN trace of that slice's rays (world-1 records), when the dump is small enough to repeat here
            validation.update(gathered_vs_world1(torch, la, scenes, shard, acc, args, mode, gathered, wire_bytes, per, nchunks, world, n_total, st_after_tris, dev))
            validation["ok"] = bool(validation["ok"]) and validation.get("gathered_equals_world1_records") is not False
            if wire16:
                validation["fp64_records_gathered_ok"] = bool(fp64_gather_ok); validation["ok"] = bool(validation["ok"]) and bool(fp64_gather_ok)

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
                                                                      "" if world == 1 else (", %d-B hit records gathered to rank 0 inside the timed region" % wire_bytes if head_gather
                                                                                             else ", hit records stay with the rank that traced them (digest to rank 0)")),
                       "rays": n_total, "rays_per_gpu": n, "triangles": args.tris, "mode": args.mode,
                       "variant": args.variant, "parallelism": "replicated BVH, ray slices x%d%s" % (world, "" if world == 1 else ((", %d-chunk trace/gather pipeline, %d workgroup slots left to the exchange's kernels" % (nchunks, args.comm_wgs)) if head_gather else ", one launch per rank, no per-ray exchange")),
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
                                         "is_headline": bool(head_gather), "record_bytes_on_the_wire": wire_bytes, "bytes_to_rank0_per_step": int(wire_bytes * (n_total - n)),
                                         "note": "every hit record gathered to rank 0 in %d chunks behind the tracing of the next chunk (lh_dist_gather: RCCL "
                                                 "point-to-point, one xGMI link per peer)%s" % (nchunks, "; on the wire: prim u32 + t, u, v fp32 rounded from the fp64 records, "
                                                 "which stay with the rank that traced them (lh_dist_pack_records16)" if wire16 else "")}
            if wire16 and fp64_elapsed is not None:
                res["record_gather_fp64"] = {"value": round(n_total / fp64_elapsed / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(fp64_elapsed * 1e3, 3),
                                             "record_bytes_on_the_wire": rec_bytes, "bytes_to_rank0_per_step": int(rec_bytes * (n_total - n)),
                                             "note": "the same pipeline with the 28-byte fp64 records themselves on the wire (rounds 1-5's exchange): link-bound at 8 ranks"}
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


if __name__ == "__main__":
    main()
