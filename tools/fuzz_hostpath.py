"""Randomised sweep of the host entry points: lh_accel_intersect_host with batch sizes from 1 to a few million (the pipelined staging's
chunk boundaries, the small-batch kernel, the persistent kernel with and without its queue) in both modes, and lh_accel_intersect1 (one
ray at a time: the host walk, and with LH_HOST_WALK=0 the coalesced device path) -- against the oracle, bit for bit.
python tools/fuzz_hostpath.py [seed] [rounds]"""
import os, sys
os.environ.setdefault("LH_POISON_OUTPUTS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lucille_amd as la
from oracle import pyoracle as po

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1; rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(seed); total = 0
for r in range(rounds):
    ntri = int(rng.choice([1, 20, 3000, 80000])); he = float(10.0 ** rng.uniform(-2.5, -0.5))
    P, idx, _, _ = po.soup(ntri, 1, he, int(rng.integers(1, 1 << 30)))
    n = int(rng.choice([1, 2, 63, 64, 65, 1000, 65535, 65536, 65537, 300001, 1048576, 1048577, 2500003]))
    org = rng.uniform(-0.2, 1.2, (n, 3)); dr = rng.normal(size=(n, 3)); dr[np.abs(dr[:, 1]) < 1e-9, 1] = 0.5
    o = po.Oracle(); o.add_mesh(P, idx); o.build(); exp = o.intersect(org, dr, nthreads=16)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build=str(rng.choice(["host", "device"]))); acc.wait_exact()
    got = acc.intersect_host(org, dr); occ = acc.intersect_host(org, dr, mode=la.MODE_ANY)
    for k in range(4):
        if not np.array_equal(np.asarray(got[k]), exp[k]): print("MISMATCH round %d n %d field %d" % (r, n, k)); sys.exit(1)
    if not np.array_equal(np.asarray(occ).astype(np.uint8), (exp[0] != po.MISS).astype(np.uint8)): print("MISMATCH round %d n %d any-hit" % (r, n)); sys.exit(1)
    m = min(n, 300)
    for j in rng.integers(0, n, m):
        hit, p, t, u, v = acc.intersect1(org[j], dr[j])
        e = (exp[0][j] != po.MISS, exp[0][j], exp[1][j], exp[2][j], exp[3][j])
        if bool(hit) != bool(e[0]) or (hit and (p, t, u, v) != (int(e[1]), float(e[2]), float(e[3]), float(e[4]))):
            print("MISMATCH round %d one ray %d: %s vs %s" % (r, j, (hit, p, t, u, v), e)); sys.exit(1)
    acc.close(); total += n
print("%d rays over %d rounds (batches of 1 .. 2.5 M, both modes) and %d single rays: equal to the oracle" % (total, rounds, 300 * rounds))
