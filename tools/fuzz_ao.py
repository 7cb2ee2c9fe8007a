"""Randomised sweep of the AO tile pipeline: random soups (some with zero-area triangles, some with vertex normals) in the example
camera's view; the fused stage (rays generated inside the any-hit kernel) against the materialised one -- frames bit-equal, counts
equal -- and the materialised AO rays' occlusion against the oracle's closest-hit answer for the same rays; both builders.
python tools/fuzz_ao.py [seed] [rounds]          (FUZZ_BUDGET_S=<seconds>: no new round after that long)"""
import os, sys, time
os.environ.setdefault("LH_POISON_OUTPUTS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from oracle import pyoracle as po
from tests.helpers import load_golden

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1; rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed)
g = load_golden("ao_c1")
allp = np.concatenate([g["pos%d" % k][:, :3] for k in range(int(g["ngeoms"]))]); lo, hi = allp.min(0), allp.max(0)
c = g["camera"]; nrays = 0; done = 0; T0 = time.time(); BUDGET = float(os.environ.get("FUZZ_BUDGET_S", "0"))
for r in range(rounds):
    if BUDGET > 0 and time.time() - T0 > BUDGET: break
    ntri = int(rng.choice([1, 6, 50, 800, 20000, 200000])); he = float(10.0 ** rng.uniform(-2.5, -0.3))
    ctr = rng.uniform(0, 1, (ntri, 1, 3)); T = (ctr + rng.normal(size=(ntri, 3, 3)) * he) * (hi - lo) + lo
    if r % 3 == 1: T[::4, 2] = T[::4, 1]
    P = T.reshape(-1, 3).copy(); idx = np.arange(3 * ntri, dtype=np.uint32)
    N = None
    if r % 2 == 1:
        n = np.cross(T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]); nn = np.linalg.norm(n, axis=1, keepdims=True); n = np.where(nn > 0, n / np.maximum(nn, 1e-300), [0.0, 0.0, 1.0])
        N = np.repeat(n, 3, 0) + rng.normal(size=(3 * ntri, 3)) * 0.2; N /= np.linalg.norm(N, axis=1, keepdims=True)
    W = int(rng.choice([33, 64, 96])); H = int(rng.choice([17, 48, 64])); pxs = int(rng.choice([1, 2])); ns = int(rng.choice([4, 16, 64])); sd = int(rng.integers(0, 1 << 30))
    cam = la.Camera.make(W, H, c[16], c[:16], int(c[19]))
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    for build in ("host", "device"):
        acc = la.HipAccel(0); acc.add_mesh(P, idx)
        if N is not None: acc.set_normals(0, N, int(r % 4 == 1))
        acc.commit(build=build); acc.wait_exact()
        acc.set_param("ao_fused", 1)
        img_f, st_f = acc.render_ao_tile(cam, 0, 0, W, H, pxs, ns, seed=sd)
        acc.set_param("ao_fused", 0)
        img_m, st_m = acc.render_ao_tile(cam, 0, 0, W, H, pxs, ns, seed=sd)
        aorg = acc.scratch(8, np.float64, 3); adir = acc.scratch(9, np.float64, 3); occ = acc.scratch(10, np.uint8, 1)
        torch.cuda.synchronize()
        if st_f != st_m or not torch.equal(img_f, img_m):
            print("MISMATCH round %d build %s: fused %s materialised %s, pixels %d" % (r, build, st_f, st_m, int((img_f != img_m).any(-1).sum()))); sys.exit(1)
        if aorg.shape[0]:
            ok = np.abs(adir[:, 1]) > 1e-14                      # the reference's |dir.y| <= 1e-14 branch is outside the contract
            exp = o.intersect(aorg[ok], adir[ok], nthreads=16)
            if not np.array_equal(occ[ok].astype(bool).ravel(), exp[0] != po.MISS):
                print("MISMATCH round %d build %s: occlusion of %d AO rays" % (r, build, int((occ[ok].astype(bool).ravel() != (exp[0] != po.MISS)).sum()))); sys.exit(1)
            nrays += int(ok.sum())
        acc.close()
    done += 1
print("%d AO rays over %d frames x 2 builders: fused == materialised bit for bit, occlusion equal to the oracle" % (nrays, done))
