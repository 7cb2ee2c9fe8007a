"""The RCCL branch of lh_dist_* (lucille_amd/csrc/lh_dist.hip) WITH REAL PEERS on a one-GPU box.

RCCL refuses two ranks on one device, so on the test box the nccl* code path never met a peer.
tests/mock_rccl/mock_rccl.cpp implements the nine entry points lh_dist.hip dlsym()s over POSIX shared memory; with
LH_RCCL_LIBRARY pointing at it and LH_DIST_TRANSPORT=rccl the product takes its RCCL branch at world 2 and 4: the status words
agreed before a payload moves, ncclBroadcast of the scene image, the grouped ncclSend / ncclRecv gather (placement per rank),
the frame assembled on rank 0, and what happens when rank 0's commit fails or a send fails (nobody hangs)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import lucille_amd as la
from lucille_amd import scenes
from oracle import pyoracle as po
from tests.helpers import assert_hits_equal, load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK_DIR = os.path.join(ROOT, "tests", "mock_rccl")


def build_mock():
    so = os.path.join(MOCK_DIR, "libmock_rccl.so"); src = os.path.join(MOCK_DIR, "mock_rccl.cpp")
    if (not os.path.exists(so)) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", so,
                               "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"])
    return so


_RANK = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
from oracle import pyoracle as po
rank, world, rdv, out, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
d = la.HipDist(rank, world, 0, rendezvous=rdv)
acc = la.HipAccel(0)
if mode == "commit_fails":
    # rank 0's commit "fails" (it never commits): it still enters the broadcast, whose first word releases the peers
    try:
        d.broadcast_scene(acc)
        print("NO ERROR"); sys.exit(1)
    except la.LucilleHipError as e:
        open(out, "w").write(str(e)); d.close(); sys.exit(0)
g = np.load(os.path.join(%(root)r, "tests", "golden", "ao_c1.npz"))
if rank == 0:
    for k in range(int(g["ngeoms"])):
        P, I = scenes.tessellate(g["pos%%d" %% k], g["idx%%d" %% k], 2); acc.add_mesh(P, I)
    acc.commit()
d.broadcast_scene(acc)
P, idx, org, dr = po.soup(10, 20000, 0.3, 5)                       # rays only: aimed through the unit cube the scene does not fill -- any rays do
org = org * 4.0 - 2.0
got = acc.intersect_host(org, dr)
x = (torch.arange(15, device="cuda", dtype=torch.float32).reshape(5, 3) + 100.0 * rank)
if mode == "send_fails":
    try:
        gat = d.gather(x)
        print("NO ERROR"); sys.exit(1)
    except la.LucilleHipError as e:
        open(out, "w").write(str(e)); sys.exit(0)                 # no close(): the communicator's peers are gone
gat = d.gather(x)
b = torch.arange(7, device="cuda", dtype=torch.float64) * (1.0 if rank == 0 else 0.0)
d.broadcast(b)
d.barrier()
c = g["camera"]; cam = la.Camera.make(96, 70, c[16], c[:16], int(c[19]))
img, st = d.render_ao_frame(acc, cam, 2, 9, seed=3, band_rows=4)
# one sample per pixel, 9 and 16 AO rays: ONE BYTE per pixel travels (k_take_count8); 289 rays: the float again
img1 = {}
for ns in (9, 16, 289):
    im, _ = d.render_ao_frame(acc, cam, 1, ns, seed=5, band_rows=4)
    img1["img1_%%d" %% ns] = im if im is not None else np.zeros(0)
np.savez(out, **img1, prim=got[0], t=got[1], u=got[2], v=got[3], gathered=(gat.cpu().numpy() if gat is not None else np.zeros(0)), bcast=b.cpu().numpy(),
         transport=d.transport, img=(img if img is not None else np.zeros(0)), stats=np.array([st[k] for k in ("primary_rays", "primary_hits", "ao_rays", "ao_occluded")]))
d.close(); acc.close()
"""


def run_ranks(tmp_path, world, mode, extra_env=None, timeout=300):
    script = tmp_path / "rank.py"
    script.write_text(_RANK % {"root": ROOT})
    env = dict(os.environ, LH_RCCL_LIBRARY=build_mock(), LH_DIST_TRANSPORT="rccl", MOCK_RCCL_LOG=str(tmp_path / "calls"))
    env.update(extra_env or {})
    rdv = str(tmp_path / "rdv")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), rdv, str(tmp_path / ("out%d" % r)) + (".npz" if mode == "ok" else ".txt"), mode],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = [p.communicate(timeout=timeout) for p in procs]
    return procs, outs


@pytest.mark.parametrize("world", [2, 4])
def test_rccl_branch_with_peers(tmp_path, world):
    procs, outs = run_ranks(tmp_path, world, "ok")
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    # one process, no communicator: the expected records and frame
    g = load_golden("ao_c1")
    acc = la.HipAccel(0)
    for k in range(int(g["ngeoms"])):
        P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 2); acc.add_mesh(P, I)
    acc.commit()
    _, _, org, dr = po.soup(10, 20000, 0.3, 5); org = org * 4.0 - 2.0
    exp = acc.intersect_host(org, dr)
    c = g["camera"]; cam = la.Camera.make(96, 70, c[16], c[:16], int(c[19]))
    ref, st_ref = acc.render_ao_frame_host(cam, 2, 9, seed=3)
    ref1 = {ns: acc.render_ao_frame_host(cam, 1, ns, seed=5)[0] for ns in (9, 16, 289)}
    acc.close()
    assert (exp[0] != po.MISS).sum() > 500
    for r in range(world):
        z = np.load(tmp_path / ("out%d.npz" % r))
        assert int(z["transport"]) == la.DIST_RCCL
        assert_hits_equal((z["prim"], z["t"], z["u"], z["v"]), exp, "rank %d walks the scene it received" % r)
        assert np.array_equal(z["bcast"], np.arange(7, dtype=np.float64))
    z0 = np.load(tmp_path / "out0.npz")
    gat = z0["gathered"]
    assert gat.shape == (world, 5, 3)
    for r in range(world):                                      # every peer's slab landed in ITS slot of rank 0's buffer
        assert np.array_equal(gat[r], np.arange(15, dtype=np.float32).reshape(5, 3) + 100.0 * r)
    assert np.array_equal(z0["img"], ref)                       # bands of `world` ranks, gathered and placed == the one-process frame
    for ns in (9, 16, 289):                                     # ... and so are the one-sample frames, whose pixels travel as counts of unoccluded rays
        assert np.array_equal(z0["img1_%d" % ns], ref1[ns]), ns
        assert len(np.unique(ref1[ns])) > min(ns, 16) // 2
    # what rank 0 received for them: width x lines-of-a-rank bytes for 9 and 16 rays, floats for 289 (17 x 17 > 255)
    nbands = (70 + 3) // 4; per = (nbands + world - 1) // world * 4 * 96
    sizes = [int(l.split()[2]) for l in open(str(tmp_path / "calls") + ".rank1").read().splitlines() if l.startswith("Send 0 ")]
    assert sizes.count(per) == 2 and sizes.count(4 * per) == 2, (per, sizes)
    assert z0["stats"].tolist() == [st_ref[k] for k in ("primary_rays", "primary_hits", "ao_rays", "ao_occluded")]
    # the call pattern of the exchange step: rank 0 posts world - 1 receives in ONE group, every peer one send in one group
    log0 = open(str(tmp_path / "calls") + ".rank0").read().splitlines()
    assert sum(1 for l in log0 if l == "GroupEnd %d" % (world - 1)) >= 4 and not any(l.startswith("Send") for l in log0)
    for r in range(1, world):
        lr = open(str(tmp_path / "calls") + ".rank%d" % r).read().splitlines()
        assert not any(l.startswith("Recv") for l in lr) and sum(1 for l in lr if l == "GroupEnd 1") == sum(1 for l in lr if l.startswith("Send 0 "))
        assert sum(1 for l in lr if l.startswith("Broadcast")) == sum(1 for l in log0 if l.startswith("Broadcast"))
    assert not any("TIMEOUT" in l or "FAILED" in l for l in log0)


@pytest.mark.parametrize("transport", ["rccl", "shm"])
def test_a_failed_commit_on_rank_0_releases_the_peers(tmp_path, transport):
    """rank 0 used to leave before the broadcast and ranks 1 .. N-1 sat in ncclBroadcast for ever (RCCL has no
    timeout).  Now the first thing that travels is a status word: every rank returns an error, promptly"""
    procs, outs = run_ranks(tmp_path, 3, "commit_fails", {"LH_DIST_TRANSPORT": transport}, timeout=120)
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so, se[-2000:])
    assert "not committed" in open(tmp_path / "out0.txt").read()
    for r in (1, 2):
        assert "rank 0 has no committed scene" in open(tmp_path / ("out%d.txt" % r)).read()


def test_a_failing_send_is_an_error_on_both_ends(tmp_path):
    """the fourth ncclSend of rank 1 fails (injected; the first carries its host id at lh_dist_init -- are all ranks on one host?
    --, the next two the status words of the scene broadcast): rank 1's gather raises with the library's message, rank 0's
    receive gives up (the mock's timeout) and raises too"""
    procs, outs = run_ranks(tmp_path, 2, "send_fails", {"MOCK_RCCL_FAIL": "send:1:3", "MOCK_RCCL_TIMEOUT": "4"}, timeout=120)
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so, se[-2000:])
    assert "ncclGroupEnd failed" in open(tmp_path / "out1.txt").read()          # a grouped send is issued -- and fails -- at ncclGroupEnd
    assert "lh_dist_gather" in open(tmp_path / "out0.txt").read()
