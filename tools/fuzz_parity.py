"""Randomised parity sweep on the GPU box: many small random scenes (soups of varying triangle size, scaled / translated boxes, meshes
with shared vertices, sliver and zero-area triangles) x rays aimed at random points, at vertices, along edges and with unnormalised
directions, in coordinate planes, along an axis, starting on a vertex; closest-hit records and any-hit flags of the device (both builders) against the oracle, bit for bit.
python tools/fuzz_parity.py [rounds] [seed]   -> one line per scene, a total, exit code 1 on the first mismatch.
FUZZ_BUDGET_S=<seconds> in the environment: no new round is started after that long (tests/test_gpu_fuzz.py).
FUZZ_CPU=1: without a device -- the one-ray host walk on the host-built trees (tests/helpers.Model) against the oracle.
FUZZ_KIND=<0..9>: every round is of that kind;  FUZZ_SAVE=<file.npz> with a fifth argument (the round to replay): the scene and rays of that round, no tracing."""
import os, sys, time
os.environ.setdefault("LH_POISON_OUTPUTS", "1")          # an answer slot nobody writes must show up as a mismatch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
CPU = os.environ.get("FUZZ_CPU") == "1"      # no device: the product's one-ray host walk (lh_hostwalk.c through tests/helpers.Model) on the host-built trees
if CPU:
    from tests.helpers import Model
else:
    import lucille_amd as la
from oracle import pyoracle as po

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
ONLY = int(sys.argv[4]) if len(sys.argv) > 4 else -1       # replay: generate every round (the same random stream), trace only this one
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"        # big: 0.3 .. 1.5 M triangles, 400 k rays as ONE device batch (fix-up queue, cooperative walk, device builders' large paths)
KIND = int(os.environ.get("FUZZ_KIND", "-1"))      # every round of this kind (the random stream is then another one than without it)
total = 0; done = 0; T0 = time.time(); BUDGET = float(os.environ.get("FUZZ_BUDGET_S", "0"))
for r in range(rounds):
    if BUDGET > 0 and time.time() - T0 > BUDGET: break
    kind = r % 10 if KIND < 0 else KIND
    ntri = int(rng.choice([300000, 700000, 1500000])) if BIG else int(rng.choice([1, 2, 7, 60, 900, 12000, 150000]))
    scale = float(10.0 ** rng.uniform(-6, 6)); shift = rng.uniform(-1, 1, 3) * scale * float(rng.choice([0.0, 1.0, 100.0]))
    he = float(10.0 ** rng.uniform(-3, -0.5))
    c = rng.uniform(0, 1, (ntri, 1, 3)); T = c + rng.normal(size=(ntri, 3, 3)) * he
    if kind == 1:                                   # a strip mesh: shared vertices, shared edges (exact-t ties)
        g = int(max(2, np.sqrt(ntri))); xs, ys = np.meshgrid(np.linspace(0, 1, g + 1), np.linspace(0, 1, g + 1))
        V = np.stack([xs.ravel(), ys.ravel(), 0.3 + 0.2 * np.sin(5 * xs.ravel()) * np.cos(3 * ys.ravel())], 1)
        q = np.array([[i * (g + 1) + j, i * (g + 1) + j + 1, (i + 1) * (g + 1) + j, i * (g + 1) + j + 1, (i + 1) * (g + 1) + j + 1, (i + 1) * (g + 1) + j] for i in range(g) for j in range(g)]).reshape(-1, 3)
        T = V[q]
    if kind == 2: T[:, :, 2] = np.round(T[:, :, 2] * 4) / 4                     # axis-aligned sheets
    if kind == 3: T[::3, 2] = T[::3, 1]; T[1::7, 2] = T[1::7, 0] + 2.0 * (T[1::7, 1] - T[1::7, 0])      # zero-area triangles among the others: two equal vertices; three points on a line
    if kind == 5: T = np.concatenate([T, T[: max(1, ntri // 2)], T[: max(1, ntri // 3)][:, ::-1]])        # the same triangle two and three times (one of them wound the other way): ties everywhere
    if kind == 6: T[:, 1] = T[:, 0] + (T[:, 1] - T[:, 0]) * 200.0                # needles, 200 times as long as wide
    if kind == 7: T[:, 0] = T[0, 0]                                              # a fan: every triangle shares one vertex
    if kind == 8: T[0] = np.array([[-40.0, -40.0, 0.5], [80.0, -40.0, 0.5], [-40.0, 80.0, 0.5]]); T[1:] = 0.5 + (T[1:] - 0.5) * 1e-3      # one huge triangle over a speck of tiny ones
    if kind == 9:                                                                # ONE to three zero-area triangles among ordinary ones (three different points on a line, any size): they stay in
        for q in rng.integers(0, T.shape[0], int(rng.integers(1, 4))):           # the tree, and only rays that reach the box of their leaf in lucille's own tree take the reference walk (round 6)
            T[q, 2] = T[q, 0] + float(rng.uniform(0.2, 3.0)) * (T[q, 1] - T[q, 0])
    if kind == 4: T[:, 2] = T[:, 0] + (T[:, 1] - T[:, 0]) * 1.0000001 + rng.normal(size=(T.shape[0], 3)) * 1e-9     # slivers
    P = (T.reshape(-1, 3) * scale + shift).astype(np.float64); idx = np.arange(P.shape[0], dtype=np.uint32)
    n = 400000 if BIG else 60000
    tri = P.reshape(-1, 3, 3); pick = rng.integers(0, tri.shape[0], n)
    w = rng.random((n, 3)); w /= w.sum(1, keepdims=True); tgt = (tri[pick] * w[:, :, None]).sum(1)
    tgt[:n // 4] = tri[pick[:n // 4], rng.integers(0, 3, n // 4)]               # exactly a vertex
    tgt[n // 4:n // 2] = 0.5 * (tri[pick[n // 4:n // 2], 0] + tri[pick[n // 4:n // 2], 1])          # on an edge
    org = tgt + rng.normal(size=(n, 3)) * scale * float(rng.choice([0.1, 1.0, 30.0]))
    dr = (tgt - org) * rng.uniform(0.001, 1000.0, (n, 1))
    dr[-n // 8:] = rng.normal(size=(n // 8, 3))                                  # anywhere
    k8 = n // 8
    dr[k8:2 * k8, 0] = 0.0                                                        # in a coordinate plane (dir.y stays: the reference's |dir.y| <= 1e-14 branch is outside the contract)
    dr[2 * k8:3 * k8, 2] = 0.0; dr[2 * k8:2 * k8 + k8 // 2, 0] = 0.0               # ... along the y axis
    org[3 * k8:3 * k8 + k8 // 2] = tri[pick[3 * k8:3 * k8 + k8 // 2], 0]           # starting exactly on a vertex
    dr[3 * k8:3 * k8 + k8 // 2] = rng.normal(size=(k8 // 2, 3))
    ok = np.abs(dr[:, 1]) > 1e-14 * np.abs(dr).max(1)                            # the reference's |dir.y| <= 1e-14 branch is outside the contract
    org, dr = np.ascontiguousarray(org[ok]), np.ascontiguousarray(dr[ok])
    if ONLY >= 0 and r != ONLY: continue
    if os.environ.get("FUZZ_SAVE"): np.savez(os.environ["FUZZ_SAVE"], P=P, idx=idx, org=org, dr=dr); print("saved round", r, "kind", kind, "scale", scale, "shift", shift, "he", he); sys.exit(0)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=16); occ = exp[0] != po.MISS            # the reference's any-hit answer: is there a closest hit
    for build in (("hostwalk",) if CPU else ("host", "device")):
        if CPU:
            m = Model(P, idx); m.ref_build()
            try: got = m.hostwalk(org, dr)
            finally: Model.ref_off()
            gocc = None; acc = None
        else:
            acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build=build)
        if CPU: pass
        elif BIG:
            import torch
            d_o = torch.from_numpy(org).cuda(); d_d = torch.from_numpy(dr).cuda()
            got = tuple(x.cpu().numpy() for x in acc.intersect_device(d_o, d_d)); gocc = acc.intersect_device(d_o, d_d, mode=la.MODE_ANY)[0].cpu().numpy()
        else:
            got = acc.intersect_host(org, dr); gocc = acc.intersect_host(org, dr, mode=la.MODE_ANY)
        for k, name in enumerate(("prim", "t", "u", "v")):
            g = np.asarray(got[k]); g = g.view(np.uint32) if g.dtype == np.int32 else g
            bad = np.nonzero(g != np.asarray(exp[k]))[0]
            if bad.size:
                print("MISMATCH round %d kind %d build %s: %s at %d rays, first %s" % (r, kind, build, name, bad.size, bad[:5])); sys.exit(1)
        if gocc is not None and not np.array_equal(np.asarray(gocc).astype(np.uint8), np.asarray(occ).astype(np.uint8)):          # raw: 0 or 1, nothing else
            print("MISMATCH round %d kind %d build %s: any-hit flags" % (r, kind, build)); sys.exit(1)
        if acc is not None: acc.close()
    total += org.shape[0]; done += 1
    print("round %2d kind %d: %6d triangles, scale %.1e, %d rays, hits %.2f: equal %s" % (r, kind, tri.shape[0], scale, org.shape[0], float((exp[0] != po.MISS).mean()), "(host walk)" if CPU else "on both builders"), flush=True)
print(("%d rays over %d scenes: the one-ray host walk's closest-hit records equal to the oracle" if CPU else "%d rays over %d scenes x 2 builders: closest-hit records and any-hit flags equal to the oracle") % (total, done))
