"""One process per GPU through the C ABI (lh_dist_*: lucille's compiled-out MPI layer, src/base/parallel.c:62-232, over RCCL):
scene broadcast instead of N builds, the gather of band slabs, and `lsh_hip --rank / --world`.  The test box has ONE GPU:
two ranks share it through the shared-memory transport (RCCL refuses duplicate devices), and the RCCL transport itself is
exercised with a world of one (every nccl* call of the N-rank path runs: init, broadcast, grouped send / recv)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import lucille_amd as la
from lucille_amd import rib, scenes
from oracle import pyoracle as po
from tests.helpers import assert_hits_equal, load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RIB = os.path.join(ROOT, "tests", "golden", "rib")


def _lsh_async(args, cwd):
    return subprocess.Popen([rib.lsh_hip_path()] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_lsh_hip_ranks_write_the_single_gpu_frame(tmp_path, world):
    """`lsh_hip --rank R --world N` -- N processes, rank 0 builds and broadcasts the scene, every rank
    renders its bands (serpentine order) as one batch, rank 0 gathers and writes the .hdr: byte-equal to the single-process file.
    world = 8: 138 lines in 16-line bands are 9 bands -- per = ceil(9 / 8) = 2 slabs per rank, seven ranks with
    one band and an empty second slab, the last band clipped: the slab padding and the seven-receive group of the driver's launch"""
    rib_path = os.path.join(RIB, "ambient_occlusion.rib")
    common = ["--resolution", "200x138", "--gather", "16", "--pixelsamples", "2", "--seed", "5"]
    one = subprocess.run([rib.lsh_hip_path()] + common + ["--output", "one.hdr", rib_path], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr
    rdv = str(tmp_path / "rendezvous")
    procs = [_lsh_async(common + ["--rank", str(r), "--world", str(world), "--rendezvous", rdv, "--device", "0", "--output", "dist.hdr", rib_path],
                        str(tmp_path)) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se
    assert "built on the host, broadcast through shared memory" in outs[0][0]
    assert all("received from rank 0" in so for so, _ in outs[1:])
    assert all("Output written" not in so for so, _ in outs[1:]) and "Output written" in outs[0][0]      # rank 0 owns the display
    assert open(tmp_path / "one.hdr", "rb").read() == open(tmp_path / "dist.hdr", "rb").read()


_RANK_SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
from oracle import pyoracle as po
rank, world, rdv, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
d = la.HipDist(rank, world, 0, rendezvous=rdv)
P, idx, org, dr = po.soup(40000, 60000, 0.01, 77)
acc = la.HipAccel(0)
if rank == 0:
    acc.add_mesh(P, idx); acc.commit(build=sys.argv[5])
d.broadcast_scene(acc)
info = acc.info()
got = acc.intersect_host(org, dr)
occ = acc.intersect_host(org, dr, la.MODE_ANY)
x = torch.full((5, 3), float(rank), device="cuda")
g = d.gather(x)
b = torch.arange(7, device="cuda", dtype=torch.float64) * (1.0 if rank == 0 else 0.0)
d.broadcast(b)
d.barrier()
np.savez(out, prim=got[0], t=got[1], u=got[2], v=got[3], occ=occ, ntri=info["ntriangles"], build=info["build_seconds"],
         gathered=(g.cpu().numpy() if g is not None else np.zeros(0)), bcast=b.cpu().numpy(), transport=d.transport,
         lookup=np.array(acc.prim_lookup(12345)))
d.close(); acc.close()
"""


@pytest.mark.parametrize("build", ["host", "device"])
def test_scene_broadcast_gives_every_rank_the_same_records(tmp_path, build):
    """lh_dist_broadcast_scene: rank 1 never sees the meshes and never builds -- it receives rank 0's flattened arrays
    (traversal nodes, triangle records, lucille's own tree, primitive lookup) and answers ray batches with the oracle's bits;
    the same when rank 0 built both trees on its device (they exist in HBM only)"""
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % {"root": ROOT})
    rdv = str(tmp_path / "rdv")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", rdv, str(tmp_path / ("out%d.npz" % r)), build],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    P, idx, org, dr = po.soup(40000, 60000, 0.01, 77)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=8)
    for r in range(2):
        z = np.load(tmp_path / ("out%d.npz" % r))
        assert int(z["transport"]) == la.DIST_SHM and int(z["ntri"]) == 40000
        assert_hits_equal((z["prim"], z["t"], z["u"], z["v"]), exp, "rank %d" % r)
        assert np.array_equal(z["occ"].astype(bool), exp[0] != po.MISS)
        assert np.array_equal(z["bcast"], np.arange(7, dtype=np.float64))
        assert tuple(z["lookup"]) == (0, 3 * 12345)
    g = np.load(tmp_path / "out0.npz")["gathered"]
    assert g.shape == (2, 5, 3) and (g[0] == 0).all() and (g[1] == 1).all()


def test_rccl_transport_with_a_world_of_one():
    """the RCCL code path itself (ncclGetUniqueId, ncclCommInitRank, ncclBroadcast, ncclGroupStart / Send / Recv / GroupEnd)
    runs on the one-GPU box with world = 1; the frame through lh_dist_render_ao_frame_host equals lh_render_ao_frame_host"""
    import torch
    uid = la.HipDist.unique_id()
    d = la.HipDist(0, 1, 0, unique_id=uid, transport=la.DIST_RCCL)
    assert d.transport == la.DIST_RCCL
    g = load_golden("ao_c1")
    acc = la.HipAccel(0)
    for k in range(int(g["ngeoms"])):
        P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 2); acc.add_mesh(P, I)
    acc.commit()
    d.broadcast_scene(acc)                       # root only: header + arrays through ncclBroadcast
    c = g["camera"]; cam = la.Camera.make(160, 120, c[16], c[:16], int(c[19]))
    img, st = d.render_ao_frame(acc, cam, 2, 16, seed=3)
    ref, st_ref = acc.render_ao_frame_host(cam, 2, 16, seed=3)
    assert st == st_ref and np.array_equal(img, ref)
    x = torch.arange(12, device="cuda", dtype=torch.float32).reshape(3, 4)
    assert torch.equal(d.gather(x)[0], x)
    d.barrier(); d.close(); acc.close()


def test_wire_records_are_the_fp64_records_rounded_to_nearest():
    """lh_dist_pack_records16 (the exchange's 16-byte record: prim u32 + t, u, v fp32): against numpy's rounding of the fp64 records of a
    real launch -- hits, misses (prim 0xFFFFFFFF, t = 1e38), an odd count, and the refusals (a misaligned buffer, a NULL array)"""
    import torch
    from lucille_amd import binding
    P, idx, org, dr = po.soup(20000, 100003, 0.01, 77)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
    out = acc.intersect_device(torch.from_numpy(org).cuda(), torch.from_numpy(dr).cuda()); torch.cuda.synchronize()
    n = org.shape[0]
    rec = torch.full((16 * n + 16,), 0x55, dtype=torch.uint8, device="cuda")
    binding.pack_records16(out[0], out[1], out[2], out[3], rec, n=n); torch.cuda.synchronize()
    r = rec[:16 * n].cpu().numpy().view(np.uint32).reshape(n, 4)
    prim = out[0].cpu().numpy().view(np.uint32)
    assert np.array_equal(r[:, 0], prim) and 0 < int((prim == po.MISS).sum()) < n
    for k in (1, 2, 3):
        assert np.array_equal(r[:, k].view(np.float32), out[k].cpu().numpy().astype(np.float32))
    assert np.all(rec[16 * n:].cpu().numpy() == 0x55)                      # nothing behind the n-th record
    hit = prim != po.MISS
    t64 = out[1].cpu().numpy()[hit]
    assert np.max(np.abs(r[hit, 1].view(np.float32).astype(np.float64) - t64) / t64) < 6.1e-8       # north_star: 1e-5
    with pytest.raises(la.LucilleHipError):
        binding.pack_records16(out[0], out[1], out[2], out[3], rec[4:], n=n)
    acc.close()
