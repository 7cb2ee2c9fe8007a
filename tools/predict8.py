"""What eight GPUs would make of a frame / a dump, predicted BY CODE on one GPU.  Launched like the driver
launches bench.py -- one process per rank under torch.distributed.run -- with every rank on device 0 and lh_dist_*'s RCCL branch on
tests/mock_rccl, whose link model (MOCK_RCCL_LATENCY_US per call, MOCK_RCCL_GBPS per peer and direction) keeps a MODEL CLOCK of what
the exchange would cost on xGMI.  Three measured or modelled terms, no number added by hand:
  * a rank's batch: every rank renders / traces ITS share alone on the GPU while the other seven wait at a barrier (best of 4);
  * the exchange: the real call sequence (lh_dist_gather of the real slabs / records, verified against one batch) on the model clock;
  * the frame's two barriers: timed between the eight real processes (median of 300).
predicted = max over ranks(batch) + exchange + 2 barriers; speed-up = rank 0's one-batch time of the WHOLE job / predicted.
  tools/r06_predict8.sh runs it for 2, 4, 8 ranks:   python -m torch.distributed.run ... tools/predict8.py ao|dump"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import binding, render, scenes, shard
from benchlegs.common import upload_rays, record_views, hip_events, EventPairs

what = sys.argv[1] if len(sys.argv) > 1 else "ao"
rank, world, _ = shard.init_process_group()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
mock = C.CDLL(os.environ["LH_RCCL_LIBRARY"]); mock.mock_rccl_model_seconds.restype = C.c_double; mock.mock_rccl_model_calls.restype = C.c_ulonglong
mock.mock_rccl_model_bytes.restype = C.c_ulonglong
LAT_US = float(os.environ.get("MOCK_RCCL_LATENCY_US", "0")); GBPS = float(os.environ.get("MOCK_RCCL_GBPS", "0"))
assert shard.dist() is not None and shard.dist().transport == la.DIST_RCCL, "predict8 needs the RCCL branch (LH_DIST_TRANSPORT=rccl + LH_RCCL_LIBRARY)"


def solo(fn, reps=4):
    """every rank in turn, alone on the GPU: -> this rank's best wall time of fn() (seconds)"""
    best = None
    for r in range(world):
        shard.barrier()
        if r == rank:
            fn(); torch.cuda.synchronize(dev)
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize(dev); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(dev); ts.append(time.perf_counter() - t0)
            best = min(ts)
        shard.barrier()
    return best


def barrier_cost(n=300):
    ts = []
    shard.barrier()
    for _ in range(n):
        t0 = time.perf_counter(); shard.barrier(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


if what == "ao":
    size, tess, ns = int(os.environ.get("P8_SIZE", 4096)), int(os.environ.get("P8_TESS", 8)), 64
    g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
    acc = la.HipAccel(0); ntri = 0
    for r in range(world):                 # one rank at a time: the tessellated meshes are 1.5 GB of host memory per process while they are staged
        shard.barrier()
        if r == rank:
            for k in range(int(g["ngeoms"])):
                P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I); ntri += I.shape[0] // 3; del P, I
            acc.commit()
    shard.barrier()
    c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
    rows, y0s = render.bands_for(size, world)
    mine = [y0s[b] for b in shard.bands_of_rank(len(y0s), rank, world)]
    per = (len(y0s) + world - 1) // world
    slab = torch.zeros((per, rows * size * 3), dtype=torch.float32, device=dev)
    st_box = {}
    def my_batch():
        _, st = acc.render_ao_bands(cam, mine, rows, 1, ns, seed=1, out=slab[:len(mine)].view(len(mine), rows, size, 3)); st_box.update(st)
    # the whole frame as one batch on rank 0 (the others wait)
    one = None; t1 = 0.0
    shard.barrier()
    if rank == 0:
        full = torch.empty((size, size, 3), dtype=torch.float32, device=dev)
        def whole(): acc.render_ao_tile(cam, 0, 0, size, size, 1, ns, seed=1, out=full)
        whole(); torch.cuda.synchronize(dev); ts = []
        for _ in range(4):
            t0 = time.perf_counter(); whole(); torch.cuda.synchronize(dev); ts.append(time.perf_counter() - t0)
        t1 = min(ts); one = full          # a tile is top line first inside (render.render_ao_frame places it as it is)
    shard.barrier()
    tb = solo(my_batch)
    # the exchange: the real sharded frame (all ranks at once: its wall time means nothing here), its calls on the model clock
    img, _ = render.render_ao_frame_sharded(acc, cam, 1, ns, rank, world)
    shard.barrier(); mock.mock_rccl_model_reset()
    img, _ = render.render_ao_frame_sharded(acc, cam, 1, ns, rank, world); torch.cuda.synchronize(dev)
    ex = {"s": mock.mock_rccl_model_seconds(), "calls": int(mock.mock_rccl_model_calls()), "bytes": int(mock.mock_rccl_model_bytes())}
    same = bool(torch.equal(img, one)) if rank == 0 else None
    bc = barrier_cost()
    rows_all = shard.all_gather_object({"rank": rank, "batch_ms": tb * 1e3, "hits": int(st_box.get("primary_hits", 0)), "exchange": ex, "barrier_us": bc * 1e6})
    if rank == 0:
        busiest = max(r["batch_ms"] for r in rows_all); ex0 = rows_all[0]["exchange"]; bar = float(np.median([r["barrier_us"] for r in rows_all])) * 1e-3
        pred = busiest + ex0["s"] * 1e3 + 2 * bar
        print(json.dumps({"what": "ao", "world": world, "triangles": ntri, "size": size, "samples": ns, "band_rows": rows, "bands": len(y0s), "one_batch_ms": t1 * 1e3,
                          "batch_ms": [round(r["batch_ms"], 3) for r in rows_all], "hits_k": [r["hits"] // 1000 for r in rows_all], "sum_ms": sum(r["batch_ms"] for r in rows_all),
                          "busiest_ms": busiest, "exchange_model_ms": ex0["s"] * 1e3, "exchange_calls_rank0": ex0["calls"], "exchange_bytes_rank0": ex0["bytes"],
                          "barrier_ms": bar, "predicted_ms": pred, "speedup": t1 * 1e3 / pred, "speedup_without_barriers": t1 * 1e3 / (busiest + ex0["s"] * 1e3),
                          "frame_equals_one_batch": same, "latency_us_per_call": LAT_US, "GBps_per_link": GBPS}), flush=True)
    acc.close()
else:
    n_total, ntris = int(os.environ.get("P8_RAYS", 100_000_000)), 1_000_000
    nchunks = int(os.environ.get("P8_CHUNKS", 8))
    P, idx, st = scenes.soup_triangles(ntris, 0.005)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
    b0, b1 = shard.ray_slice(n_total, rank, world); n = b1 - b0
    d_org, d_dir, _ = upload_rays(scenes, torch, dev, scenes.skip(st, 5 * b0), n)
    per = shard.chunk_capacity(n_total, world, nchunks)
    cb = [(c * per, min(n, (c + 1) * per)) for c in range(nchunks)]
    bufs = [torch.empty(per * 28, dtype=torch.uint8, device=dev) for _ in range(nchunks)]
    wire = [torch.empty(per * 16, dtype=torch.uint8, device=dev) for _ in range(nchunks)]
    hip = hip_events(); stream = torch.cuda.current_stream(dev); sptr = C.c_void_p(stream.cuda_stream)
    ev_box = {}
    def my_slice(pack=True):
        evp = EventPairs(hip, nchunks)
        for c in range(nchunks):
            o = record_views(torch, bufs[c], per); m = cb[c][1] - cb[c][0]
            evp.begin(sptr)
            if m > 0:
                acc.intersect_device(d_org[cb[c][0]:cb[c][1]], d_dir[cb[c][0]:cb[c][1]], out=tuple(x[:m] for x in o))
                if pack:
                    binding.pack_records16(o[0], o[1], o[2], o[3], wire[c], n=per, stream=stream)
            evp.end(sptr)
        ev_box["evp"] = evp
    # the whole dump as ONE launch on rank 0
    t1 = 0.0
    shard.barrier()
    if rank == 0:
        fo, fd, _ = upload_rays(scenes, torch, dev, st, n_total)
        outs = (torch.empty(n_total, dtype=torch.int32, device=dev),) + tuple(torch.empty(n_total, dtype=torch.float64, device=dev) for _ in range(3))
        acc.intersect_device(fo, fd, out=outs); torch.cuda.synchronize(dev); ts = []
        for _ in range(4):
            t0 = time.perf_counter(); acc.intersect_device(fo, fd, out=outs); torch.cuda.synchronize(dev); ts.append(time.perf_counter() - t0)
        t1 = min(ts); del fo, fd, outs; torch.cuda.empty_cache()
    shard.barrier()
    out = {}
    for wb, pack in ((16, True), (28, False)):
        tb = solo(lambda: my_slice(pack))
        taus = ev_box["evp"].ms()                                  # this rank's chunks of its LAST solo repetition (ms)
        # the exchange: chunk by chunk through lh_dist_gather (the real records), on the model clock
        gathered = torch.empty((world, per * wb), dtype=torch.uint8, device=dev) if rank == 0 else None
        gm = []
        for c in range(nchunks):
            shard.barrier(); mock.mock_rccl_model_reset()
            shard.gather_bytes((wire if pack else bufs)[c], gathered, stream=stream); torch.cuda.synchronize(dev)
            gm.append(mock.mock_rccl_model_seconds() * 1e3)
        out[wb] = {"rank": rank, "slice_ms": tb * 1e3, "chunk_ms": taus, "gather_model_ms": gm}
    bc = barrier_cost()
    allr = shard.all_gather_object({"out": out, "barrier_us": bc * 1e6})
    if rank == 0:
        bar = float(np.median([r["barrier_us"] for r in allr])) * 1e-3
        res = {"what": "dump", "world": world, "rays": n_total, "chunks": nchunks, "one_launch_ms": t1 * 1e3, "barrier_ms": bar, "latency_us_per_call": LAT_US, "GBps_per_link": GBPS}
        for wb in (16, 28):
            rows = [r["out"][wb] for r in allr]
            ft = np.cumsum(np.array([r["chunk_ms"] for r in rows]), axis=1)            # [rank, chunk]: when a rank's chunk c is traced (and packed)
            ready = ft.max(axis=0)                                                       # rank 0 receives chunk c when EVERY peer has it
            g = rows[0]["gather_model_ms"]; fin = 0.0
            for c in range(nchunks):
                fin = max(ready[c], fin) + g[c]
            pred = fin + 2 * bar
            res["wire_%d" % wb] = {"slice_ms": [round(r["slice_ms"], 3) for r in rows], "busiest_slice_ms": max(r["slice_ms"] for r in rows),
                                   "gather_model_ms_per_chunk": round(float(np.mean(g)), 4), "gather_model_ms_total": round(float(np.sum(g)), 3),
                                   "pipeline_end_ms": round(fin, 3), "predicted_ms": round(pred, 3), "speedup": round(t1 * 1e3 / pred, 3),
                                   "bytes_per_peer": int(per * wb * nchunks)}
        print(json.dumps(res), flush=True)
    acc.close()
shard.barrier()
shard.dist().close()
torch.distributed.destroy_process_group()
