"""round 5: which rays of an any-hit dump are never written?  A fan of 300 k triangles about one vertex (tools/fuzz_parity.py kind 7, big),
400 k rays; the output buffer is filled with 7 before the launch."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lucille_amd as la, torch
from oracle import pyoracle as po
if len(sys.argv) > 1 and sys.argv[1].endswith(".npz"):
    z = np.load(sys.argv[1]); P, idx, org, dr = z["P"], z["idx"], z["org"], z["dr"]; n = org.shape[0]; ntri = P.shape[0] // 3
else:
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
    ntri = 300000; he = 0.05
    c = rng.uniform(0, 1, (ntri, 1, 3)); T = c + rng.normal(size=(ntri, 3, 3)) * he; T[:, 0] = T[0, 0]
    P = T.reshape(-1, 3).copy(); idx = np.arange(3 * ntri, dtype=np.uint32)
    n = 400000; pick = rng.integers(0, ntri, n)
    w = rng.random((n, 3)); w /= w.sum(1, keepdims=True); tgt = (T[pick] * w[:, :, None]).sum(1)
    tgt[:n // 4] = T[pick[:n // 4], rng.integers(0, 3, n // 4)]
    org = tgt + rng.normal(size=(n, 3)); dr = (tgt - org) * rng.uniform(0.001, 1000.0, (n, 1))
    ok = np.abs(dr[:, 1]) > 1e-14 * np.abs(dr).max(1); org, dr = np.ascontiguousarray(org[ok]), np.ascontiguousarray(dr[ok]); n = org.shape[0]
o = po.Oracle(); o.add_mesh(P, idx); o.build(); exp = o.intersect(org, dr, nthreads=16); hit = exp[0] != po.MISS
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build="host"); acc.wait_exact()
d_o = torch.from_numpy(org).cuda(); d_d = torch.from_numpy(dr).cuda()
for rep in range(1):
    out = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
    acc.intersect_device(d_o, d_d, out=(out,), mode=la.MODE_ANY); torch.cuda.synchronize()
    g = out.cpu().numpy()
    print("any-hit launch %d: never written %d, flagged values left %s, wrong among the written %d" % (rep, int((g == 7).sum()), np.unique(g[(g != 0) & (g != 1) & (g != 7)]), int(((g != 7) & (g.astype(bool) != hit)).sum())), flush=True)
    if (g == 7).any():
        w7 = np.nonzero(g == 7)[0]; print("   first unwritten", w7[:10], "oracle hit share among them", float(hit[w7].mean()))
pr = torch.full((n,), 0x77777777, dtype=torch.int32, device="cuda"); tt = torch.zeros(n, dtype=torch.float64, device="cuda"); uu = torch.zeros_like(tt); vv = torch.zeros_like(tt)
acc.intersect_device(d_o, d_d, out=(pr, tt, uu, vv)); torch.cuda.synchronize()
print("closest launch: never written %d, prim mismatches %d" % (int((pr == 0x77777777).sum()), int((pr.cpu().numpy().view(np.uint32) != exp[0]).sum())))
for mode, name in ((la.MODE_ANY, "any"), (la.MODE_CLOSEST, "closest")):
    outs, c = acc.intersect_device(d_o, d_d, mode=mode, counters=True)
    print("counted %s launch: %s" % (name, c), flush=True)
