"""What eight GPUs would make of a frame / a dump, predicted BY CODE on one GPU.  Launched like the driver
launches bench.py -- one process per rank under torch.distributed.run -- with every rank on device 0 and lh_dist_*'s RCCL branch on
tests/mock_rccl, whose link model (MOCK_RCCL_LATENCY_US per call, MOCK_RCCL_GBPS per peer and direction) keeps a MODEL CLOCK of what
the exchange would cost on xGMI.  Three measured or modelled terms, no number added by hand:
  * a rank's batch: every rank's share rendered / traced ALONE on the GPU, one after the other, by ONE process (`solo-ao W` / `solo-dump W`,
    best of 4; eight processes that share a device disturb each other's "solo" launches even while seven of them only wait -- the
    first version of this tool timed them there and read 7.6 .. 19.8 ms for equal shares);
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
OUT = os.path.join(ROOT, "gpurun_out"); os.makedirs(OUT, exist_ok=True)
CHUNKS = (1, 2, 4, 8, 16)


def ao_scene(tess):
    g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz")); acc = la.HipAccel(0); ntri = 0
    for k in range(int(g["ngeoms"])):
        P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I); ntri += I.shape[0] // 3; del P, I
    acc.commit()
    return g, acc, ntri


def best_of(fn, reps=4):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)


if what == "solo-ao":            # ONE process: the whole frame as one batch, then every rank's batch of `world`, alone on the GPU
    world = int(sys.argv[2]); size, tess, ns = int(os.environ.get("P8_SIZE", 4096)), int(os.environ.get("P8_TESS", 8)), 64
    g, acc, ntri = ao_scene(tess)
    c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
    full = torch.empty((size, size, 3), dtype=torch.float32, device="cuda")
    t1 = best_of(lambda: acc.render_ao_tile(cam, 0, 0, size, size, 1, ns, seed=1, out=full))
    rows, y0s = render.bands_for(size, world); per = (len(y0s) + world - 1) // world
    slab = torch.zeros((per, rows, size, 3), dtype=torch.float32, device="cuda")
    batches = [[y0s[b] for b in shard.bands_of_rank(len(y0s), r, world)] for r in range(world)]
    for mine in batches:                 # one untimed pass over every rank's batch first (scratch buffers re-grown, clocks)
        acc.render_ao_bands(cam, mine, rows, 1, ns, seed=1, out=slab[:len(mine)])
    bms = []; hits = []
    for mine in batches:
        st_box = {}
        def one(): st_box.update(acc.render_ao_bands(cam, mine, rows, 1, ns, seed=1, out=slab[:len(mine)])[1])
        bms.append(best_of(one) * 1e3); hits.append(int(st_box["primary_hits"]))
    json.dump({"world": world, "triangles": ntri, "size": size, "samples": ns, "band_rows": rows, "bands": len(y0s), "one_batch_ms": t1 * 1e3, "batch_ms": bms, "hits": hits},
              open(os.path.join(OUT, "r06_solo_ao_%d.json" % world), "w"))
    print("solo-ao world %d: one batch %.2f ms; batches %s" % (world, t1 * 1e3, " ".join("%.2f" % b for b in bms)), flush=True)
    sys.exit(0)
if what == "solo-dump":          # ONE process: the whole dump as one launch, then every rank's slice in 1 / 2 / 4 / 8 chunks (+ the wire records), alone on the GPU
    world = int(sys.argv[2]); n_total = int(os.environ.get("P8_RAYS", 100_000_000))
    P, idx, st = scenes.soup_triangles(1_000_000, 0.005)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(); dev = torch.device("cuda", 0)
    fo, fd, _ = upload_rays(scenes, torch, dev, st, n_total)
    outs = (torch.empty(n_total, dtype=torch.int32, device=dev),) + tuple(torch.empty(n_total, dtype=torch.float64, device=dev) for _ in range(3))
    t1 = best_of(lambda: acc.intersect_device(fo, fd, out=outs))
    stream = torch.cuda.current_stream(dev)
    two = os.environ.get("P8_TWO_STREAMS") == "1"       # the chunks alternate between two trace streams (measured and dropped: bench.py one_step)
    tstreams = [torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev)] if two else [stream, stream]
    res = {"world": world, "rays": n_total, "one_launch_ms": t1 * 1e3, "chunks": {}}
    for nch in CHUNKS:
        per = shard.chunk_capacity(n_total, world, nch)
        bufs = [torch.empty(per * 28, dtype=torch.uint8, device=dev) for _ in range(nch)]; wire = [torch.empty(per * 16, dtype=torch.uint8, device=dev) for _ in range(nch)]
        rows = {16: [], 28: []}
        for r in range(world):
            b0, b1 = shard.ray_slice(n_total, r, world); n = b1 - b0
            cb = [(b0 + c * per, b0 + min(n, (c + 1) * per)) for c in range(nch)]
            for wb in (16, 28):
                box = {}
                def my_slice():
                    e0 = torch.cuda.Event(enable_timing=True); ends = [torch.cuda.Event(enable_timing=True) for _ in range(nch)]
                    e0.record(stream)
                    if two:
                        for ts in tstreams:
                            ts.wait_stream(stream)
                    for c in range(nch):
                        ts = tstreams[c % 2] if nch > 1 else stream
                        o = record_views(torch, bufs[c], per); m = cb[c][1] - cb[c][0]
                        if m > 0:
                            acc.intersect_device(fo[cb[c][0]:cb[c][1]], fd[cb[c][0]:cb[c][1]], out=tuple(x[:m] for x in o), stream=ts.cuda_stream)
                            if wb == 16:
                                binding.pack_records16(o[0], o[1], o[2], o[3], wire[c], n=per, stream=ts)
                        ends[c].record(ts)
                    if two:
                        for ts in tstreams:
                            stream.wait_stream(ts)
                    box["ev"] = (e0, ends)
                tb = best_of(my_slice, 3)
                e0, ends = box["ev"]
                rows[wb].append({"slice_ms": tb * 1e3, "done_ms": [e0.elapsed_time(e) for e in ends]})       # when chunk c of the LAST repetition was traced (and packed)
        res["chunks"][str(nch)] = {"per": per, "wire_16": rows[16], "wire_28": rows[28]}
        print("solo-dump world %d, %d chunk(s): slices (16-B wire) %s ms; rank 0's chunks done at %s ms" % (world, nch, " ".join("%.2f" % x["slice_ms"] for x in rows[16]),
              " ".join("%.2f" % d for d in rows[16][0]["done_ms"])), flush=True)
        del bufs, wire
    json.dump(res, open(os.path.join(OUT, "r06_solo_dump_%d.json" % world), "w"))
    sys.exit(0)

rank, world, _ = shard.init_process_group()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
mock = C.CDLL(os.environ["LH_RCCL_LIBRARY"]); mock.mock_rccl_model_seconds.restype = C.c_double; mock.mock_rccl_model_calls.restype = C.c_ulonglong
mock.mock_rccl_model_bytes.restype = C.c_ulonglong
LAT_US = float(os.environ.get("MOCK_RCCL_LATENCY_US", "0")); GBPS = float(os.environ.get("MOCK_RCCL_GBPS", "0"))
assert shard.dist() is not None and shard.dist().transport == la.DIST_RCCL, "predict8 needs the RCCL branch (LH_DIST_TRANSPORT=rccl + LH_RCCL_LIBRARY)"


def skew_term(n=300):
    """what the two barriers that bracket a timed frame cost between these `world` real processes (tools/skew_probe.py's measure): the
    ranks arrive at different times, as after a frame; spread of their exits + latency from the last one in to the last one out (ms, p50)"""
    import torch.distributed as tdist
    enter = np.zeros(n); leave = np.zeros(n)
    for it in range(n + 20):
        time.sleep(0.002 * ((rank * 7 + it) % 5) / 5.0)
        t0 = time.monotonic_ns(); shard.barrier(); t1 = time.monotonic_ns()
        if it >= 20:
            enter[it - 20] = t0; leave[it - 20] = t1
    allv = [None] * world; tdist.all_gather_object(allv, (enter, leave))
    E = np.stack([a for a, _ in allv]); L = np.stack([b for _, b in allv])
    return float(np.percentile((L.max(0) - L.min(0)) / 1e6 + (L.max(0) - E.max(0)) / 1e6, 50))


if what == "ao":
    size, tess, ns = int(os.environ.get("P8_SIZE", 4096)), int(os.environ.get("P8_TESS", 8)), 64
    solo = json.load(open(os.path.join(OUT, "r06_solo_ao_%d.json" % world)))
    acc = None
    for r in range(world):                 # one rank at a time: the tessellated meshes are 1.5 GB of host memory per process while they are staged
        shard.barrier()
        if r == rank:
            g, acc, ntri = ao_scene(tess)
    shard.barrier()
    c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
    one = None
    if rank == 0:
        one = torch.empty((size, size, 3), dtype=torch.float32, device=dev)
        acc.render_ao_tile(cam, 0, 0, size, size, 1, ns, seed=1, out=one); torch.cuda.synchronize(dev)       # a tile is top line first inside: the frame as it is
    # the exchange: the real sharded frame (all ranks at once on one GPU: its wall time means nothing here), its calls on the model clock
    img, _ = render.render_ao_frame_sharded(acc, cam, 1, ns, rank, world)
    shard.barrier(); mock.mock_rccl_model_reset()
    img, _ = render.render_ao_frame_sharded(acc, cam, 1, ns, rank, world); torch.cuda.synchronize(dev)
    ex = {"s": mock.mock_rccl_model_seconds(), "calls": int(mock.mock_rccl_model_calls()), "bytes": int(mock.mock_rccl_model_bytes())}
    same = bool(torch.equal(img, one)) if rank == 0 else None
    sk = skew_term()
    exs = shard.all_gather_object(ex)
    if rank == 0:
        busiest = max(solo["batch_ms"]); exm = max(e["s"] for e in exs) * 1e3
        pred = busiest + exm + sk
        print(json.dumps({"what": "ao", "world": world, "triangles": solo["triangles"], "size": size, "samples": ns, "band_rows": solo["band_rows"], "bands": solo["bands"],
                          "one_batch_ms": solo["one_batch_ms"], "batch_ms": [round(b, 3) for b in solo["batch_ms"]], "hits_k": [h // 1000 for h in solo["hits"]],
                          "sum_ms": sum(solo["batch_ms"]), "busiest_ms": busiest, "exchange_model_ms": exm, "exchange_calls_rank0": exs[0]["calls"],
                          "exchange_bytes_rank0": exs[0]["bytes"], "barriers_ms": sk, "predicted_ms": pred, "speedup": solo["one_batch_ms"] / pred,
                          "speedup_without_barriers": solo["one_batch_ms"] / (busiest + exm), "frame_equals_one_batch": same,
                          "latency_us_per_call": LAT_US, "GBps_per_link": GBPS}), flush=True)
    acc.close()
else:
    n_total = int(os.environ.get("P8_RAYS", 100_000_000))
    solo = json.load(open(os.path.join(OUT, "r06_solo_dump_%d.json" % world)))
    # the exchange: chunk by chunk through lh_dist_gather (buffers of the real sizes), on the model clock
    gm = {}
    stream = torch.cuda.current_stream(dev)
    for nch in CHUNKS:
        per = solo["chunks"][str(nch)]["per"]
        for wb in (16, 28):
            buf = torch.zeros(per * wb, dtype=torch.uint8, device=dev); buf[:8] = rank + 1
            gathered = torch.empty((world, per * wb), dtype=torch.uint8, device=dev) if rank == 0 else None
            shard.barrier(); mock.mock_rccl_model_reset()
            shard.gather_bytes(buf, gathered, stream=stream); torch.cuda.synchronize(dev)
            gm["%d_%d" % (nch, wb)] = mock.mock_rccl_model_seconds() * 1e3
            if rank == 0:
                assert [int(gathered[r][0]) for r in range(world)] == [r + 1 for r in range(world)]
            del buf, gathered
    sk = skew_term()
    gms = shard.all_gather_object(gm)
    if rank == 0:
        res = {"what": "dump", "world": world, "rays": n_total, "one_launch_ms": solo["one_launch_ms"], "barriers_ms": sk, "latency_us_per_call": LAT_US, "GBps_per_link": GBPS, "chunks": {}}
        for nch in CHUNKS:
            ch = solo["chunks"][str(nch)]; e = {}
            for wb in (16, 28):
                rows = ch["wire_%d" % wb]
                ft = np.array([r_["done_ms"] for r_ in rows])                              # [rank, chunk]: when a rank's chunk c is traced (and packed)
                ready = ft.max(axis=0)                                                     # rank 0 receives chunk c when EVERY peer has it
                g = max(x["%d_%d" % (nch, wb)] for x in gms); fin = 0.0
                for c in range(nch):
                    fin = max(ready[c], fin) + g
                pred = fin + sk
                e["wire_%d" % wb] = {"busiest_slice_ms": round(max(r_["slice_ms"] for r_ in rows), 3), "gather_model_ms_per_chunk": round(g, 4),
                                     "pipeline_end_ms": round(fin, 3), "predicted_ms": round(pred, 3), "speedup": round(solo["one_launch_ms"] / pred, 3),
                                     "bytes_per_peer": int(ch["per"] * wb * nch)}
            e["records_stay_with_rank"] = {"predicted_ms": round(max(r_["slice_ms"] for r_ in ch["wire_28"]) + sk, 3),
                                           "speedup": round(solo["one_launch_ms"] / (max(r_["slice_ms"] for r_ in ch["wire_28"]) + sk), 3)}
            res["chunks"][str(nch)] = e
        print(json.dumps(res), flush=True)
shard.barrier()
shard.dist().close()
torch.distributed.destroy_process_group()
