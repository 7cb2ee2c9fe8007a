"""N > 1 on a ONE-GPU box, the torch.distributed way (one process per rank, as bench.py / the driver launch it): two
ranks share cuda:0 (lh_dist_* over its shared-memory transport for the scene broadcast and the exchange step; gloo is the
control plane only), each with its own replica of the BVH -- built ONCE, on rank 0.  The sharded AO and path-traced
frames gathered on rank 0 must equal the unsharded frame bit for bit, and bench.py's own N = 2 code path (strong-scaling
ray dump with the hit-record gather inside the timed region, sharded AO / PT legs) must validate itself."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      LH_DEVICE_OVERRIDE="0")
    import torch
    import torch.distributed as dist
    import lucille_amd as la
    from lucille_amd import render, shard
    from tests.helpers import load_golden
    shard.init_process_group(backend="gloo")
    torch.cuda.set_device(0)
    g = load_golden("ao_ps")
    acc = la.HipAccel(0)
    def add_meshes(a):
        for k in range(int(g["ngeoms"])):
            a.add_mesh(g["pos%d" % k], g["idx%d" % k])
            if ("nrm%d" % k) in g.files:
                a.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
    # rank 0 builds, rank 1 receives the flattened scene (lh_dist_broadcast_scene; shared-memory transport: one GPU)
    info, _, _ = shard.commit_shared(acc, add_meshes, rank, world)
    assert shard.dist().transport == la.DIST_SHM and info["ntriangles"] == 1986
    c = g["camera"]
    cam = la.Camera.make(200, 150, c[16], c[:16], int(c[19]))        # 150 lines: the last band is clipped
    out = {}
    # one sample per pixel: the bands travel as one byte per pixel (the count of unoccluded rays), LH_DIST_AO_BYTES=4: as floats
    for name, env4 in (("bands_bytes", "1"), ("bands_floats", "4")):
        os.environ["LH_DIST_AO_BYTES"] = env4
        img, _ = render.render_ao_frame_sharded(acc, cam, 1, 16, rank, world, seed=4)
        if rank == 0:
            ref, _ = render.render_ao_frame(acc, cam, 1, 16, tile=200, seed=4)
            out[name] = bool(torch.equal(img, ref)) and int(torch.unique(ref).numel()) > 8
    os.environ.pop("LH_DIST_AO_BYTES")
    for name, kw in (("bands", {}), ("tiles", {"tile": 64})):
        img, st = render.render_ao_frame_sharded(acc, cam, 2, 16, rank, world, seed=3, **kw)
        stt = torch.tensor([st["primary_rays"], st["primary_hits"], st["ao_rays"], st["ao_occluded"]], dtype=torch.int64)
        dist.all_reduce(stt)
        if rank == 0:
            ref, st1 = render.render_ao_frame(acc, cam, 2, 16, tile=200, seed=3)
            out[name] = bool(torch.equal(img, ref)) and [int(x) for x in stt] == [st1[k] for k in ("primary_rays", "primary_hits", "ao_rays", "ao_occluded")]
    pimg, pst = render.render_pt_frame_sharded(acc, cam, 12, rank, world, tile=64, spp_chunk=6, kd=0.7, env=(1.0, 0.9, 0.8), max_vertices=5, seed=2)
    if rank == 0:
        ref = torch.zeros((150, 200, 3), dtype=torch.float32, device="cuda")
        for s0 in (0, 6):
            acc.render_pt_tile(cam, 0, 0, 200, 150, s0, 6, 12, kd=0.7, env=(1.0, 0.9, 0.8), max_vertices=5, seed=2, out=ref)
        torch.cuda.synchronize()
        out["pt"] = bool(torch.allclose(pimg, ref, rtol=0, atol=1e-6))
    # 152 lines: whole 4-line bands -> every rank's interleaved bands as ONE pass per sample chunk (lh_render_pt_bands)
    cam2 = la.Camera.make(200, 152, c[16], c[:16], int(c[19]))
    pimg, pst = render.render_pt_frame_sharded(acc, cam2, 12, rank, world, spp_chunk=6, kd=0.7, env=(1.0, 0.9, 0.8), max_vertices=5, seed=2)
    rays = torch.tensor([pst["rays"], pst["paths"]], dtype=torch.int64); dist.all_reduce(rays)
    if rank == 0:
        ref = torch.zeros((152, 200, 3), dtype=torch.float32, device="cuda"); rr = 0
        for s0 in (0, 6):
            _, st = acc.render_pt_tile(cam2, 0, 0, 200, 152, s0, 6, 12, kd=0.7, env=(1.0, 0.9, 0.8), max_vertices=5, seed=2, out=ref); rr += st["rays"]
        torch.cuda.synchronize()
        out["pt_bands"] = bool(torch.equal(pimg, ref)) and int(rays[0]) == rr and int(rays[1]) == 2 * 200 * 152 * 6
        q.put(out)
    dist.barrier()
    shard.dist().close()
    acc.close()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_the_unsharded_frames():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert out == {"bands_bytes": True, "bands_floats": True, "bands": True, "tiles": True, "pt": True, "pt_bands": True}, out


def test_bench_n2_code_path_on_one_gpu():
    """bench.py exactly as the driver launches it for N = 2, both ranks on device 0 (gloo): the JSON line must carry a
    passing validation of the strong-scaling ray dump (rank 0's own records + the gathered slices) and of the legs"""
    env = dict(os.environ)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--device-override", "0", "--rays", "3000001", "--tris", "200000", "--half-extent", "0.01",
           "--no-cpu", "--no-hbm", "--ao-size", "256", "--ao-tess", "2", "--ao-samples", "16", "--pt-size", "128", "--pt-spp", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["rays"] == 3000001
    assert j["validation"]["ok"] and j["validation"]["gathered_records_ok"] and j["validation"]["timed_equals_counted_launch"]
    assert j["ao_render"]["validation"]["ok"] and j["ao_render"]["rays_per_frame"] > 0
    assert j["pt_render"]["rays_per_frame"] > 0 and j["value"] > 0
    # at N > 1 the record gather is INSIDE the headline's timed region; both readings are emitted side by side
    assert j["exchange"]["headline_includes_record_gather"] and j["with_record_gather"]["is_headline"]
    assert abs(j["with_record_gather"]["value"] - j["value"]) <= 0.02 * j["value"] and j["records_stay_with_rank"]["value"] > 0
    assert "16-B hit records gathered to rank 0 inside the timed region" in j["config"]["workload"]
    assert j["validation"]["gathered_equals_world1_records"] is True and j["validation"]["gathered_records_checked"] == 3000001
    # every rank reported what it did (transport, launches, kernel and gather time); two ranks on one device = the shm transport
    assert [r_["rank"] for r_ in j["ranks"]] == [0, 1] and all(r_["transport"] == "shm" and "share a device" in r_["rccl_status"] for r_ in j["ranks"])
    assert all(r_["kernel_ms_per_step"] > 0 and r_["gather_only_ms"] >= 0 for r_ in j["ranks"])
    assert [r_["rank"] for r_ in j["ao_render"]["ranks"]] == [0, 1] and all(r_["bands"] > 0 and r_["batch_ms"] > 0 for r_ in j["ao_render"]["ranks"])
    assert "[bench rank 1]" in r.stderr


def test_bench_n2_over_the_rccl_branch_with_a_mock_library():
    """the same launch with LH_DIST_TRANSPORT=rccl and tests/mock_rccl as the library: bench.py's own N = 2 path -- the id from
    rank 0 through gloo, ncclCommInitRank on every rank, the scene through ncclBroadcast, the record and band-slab gathers through
    grouped ncclSend / ncclRecv -- with a real peer, on the one-GPU box; the line says transport rccl and validates"""
    from tests.test_gpu_dist_mock import build_mock
    env = dict(os.environ, LH_DIST_TRANSPORT="rccl", LH_RCCL_LIBRARY=build_mock(), MOCK_RCCL_TIMEOUT="120")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--device-override", "0", "--rays", "2000001", "--tris", "100000", "--half-extent", "0.01", "--record-bytes", "28",
           "--no-cpu", "--no-hbm", "--no-pt", "--ao-size", "192", "--ao-tess", "2", "--ao-samples", "16"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l][-1])
    assert j["n_gpus"] == 2 and j["validation"]["ok"] and j["validation"]["gathered_records_ok"]
    # the fp64 records themselves on the wire (28 B): every gathered slice == this rank's own trace of that slice's rays, bit for bit
    assert j["with_record_gather"]["record_bytes_on_the_wire"] == 28 and j["validation"]["gathered_equals_world1_records"] is True
    assert j["validation"]["gathered_records_checked"] == 2000001
    assert j["config"]["scene_load"]["transport"] == "rccl" and all(r_["transport"] == "rccl" and r_["rccl_status"] == "ok" for r_ in j["ranks"])
    assert j["exchange"]["transport_ok"] and j["ao_render"]["validation"]["ok"] and all(r_["transport"] == "rccl" for r_ in j["ao_render"]["ranks"])


def _bench_n8(extra_env, extra_args=()):
    env = dict(os.environ); env.update(extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--device-override", "0", "--rays", "1600003", "--tris", "60000", "--half-extent", "0.012",
           "--no-cpu", "--no-hbm", "--ao-size", "200", "--ao-tess", "2", "--ao-samples", "16", "--pt-size", "96", "--pt-spp", "4"] + list(extra_args)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1500)
    return r, (json.loads([l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l][-1]) if r.returncode == 0 else None)


def _check_n8(j, transport):
    assert j["n_gpus"] == 8 and j["scaling"] == "strong" and j["config"]["rays"] == 1600003
    assert j["validation"]["ok"] and j["validation"]["gathered_records_ok"] and j["validation"]["timed_equals_counted_launch"]
    # 16-byte wire records (the default): every rank's gathered slice == rank 0's own trace of that slice (prim equal, t / u / v the fp64
    # values rounded to fp32), and the fp64 records gathered beside them (`record_gather_fp64`) arrive bit for bit
    assert j["with_record_gather"]["record_bytes_on_the_wire"] == 16 and j["validation"]["gathered_equals_world1_records"] is True
    assert j["validation"]["gathered_records_checked"] == 1600003 and j["validation"]["fp64_records_gathered_ok"] is True
    assert j["record_gather_fp64"]["record_bytes_on_the_wire"] == 28
    assert [r_["rank"] for r_ in j["ranks"]] == list(range(8)) and all(r_["transport"] == transport for r_ in j["ranks"])
    assert sum(r_["rays"] for r_ in j["ranks"]) == 1600003
    ao = j["ao_render"]
    # 200 lines in 16-line bands: 13 bands on 8 ranks -- per = 2 slabs, three ranks hold an empty second slab, the last band is clipped
    assert ao["validation"]["ok"] and ao["validation"]["sharded_frame_equals_one_batch"] is True
    assert [r_["rank"] for r_ in ao["ranks"]] == list(range(8)) and sorted(r_["bands"] for r_ in ao["ranks"]) == [1, 1, 1, 2, 2, 2, 2, 2]
    assert all(r_["transport"] == transport for r_ in ao["ranks"]) and j["pt_render"]["rays_per_frame"] > 0


def test_bench_n8_code_path_on_one_gpu():
    """bench.py exactly as the driver launches it for N = 8 -- eight processes, all on device 0 (shared-memory
    transport): eight `ranks` entries, the gathered records and the gathered AO frame equal to what one rank produces alone"""
    r, j = _bench_n8({})
    assert r.returncode == 0, r.stderr[-3000:]
    _check_n8(j, "shm")
    assert all("share a device" in r_["rccl_status"] for r_ in j["ranks"])


def test_bench_n8_over_the_rccl_branch_with_a_mock_library():
    """... and over lh_dist_*'s RCCL branch (tests/mock_rccl): ncclCommInitRank with eight ranks, the scene through ncclBroadcast,
    the seven-receive group of every gather on rank 0, the status words agreed by eight ranks"""
    from tests.test_gpu_dist_mock import build_mock
    r, j = _bench_n8({"LH_DIST_TRANSPORT": "rccl", "LH_RCCL_LIBRARY": build_mock(), "MOCK_RCCL_TIMEOUT": "240"})
    assert r.returncode == 0, r.stderr[-3000:]
    _check_n8(j, "rccl")
    assert j["config"]["scene_load"]["transport"] == "rccl" and j["exchange"]["transport_ok"] and all(r_["rccl_status"] == "ok" for r_ in j["ranks"])
