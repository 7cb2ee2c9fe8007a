"""tools/fuzz_parity.py's scene and ray generator on the CPU: the product's one-ray host walk (lh_hostwalk.c, through tests/cpu_model) and the
host builders (lh_bvh.c, lh_refbvh.c) against the oracle.  Under the sanitizers: build tests/cpu_model/liblh_model.so with
-fsanitize=address,undefined and run with LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)".
python tools/fuzz_hostwalk.py <seed> <rounds>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from tests.helpers import Model
rng = np.random.default_rng(int(sys.argv[1])); rounds = int(sys.argv[2]); total = 0
for r in range(rounds):
    kind = r % 5
    ntri = int(rng.choice([1, 2, 7, 60, 900, 12000]))
    scale = float(10.0 ** rng.uniform(-6, 6)); shift = rng.uniform(-1, 1, 3) * scale * float(rng.choice([0.0, 1.0, 100.0]))
    he = float(10.0 ** rng.uniform(-3, -0.5))
    c = rng.uniform(0, 1, (ntri, 1, 3)); T = c + rng.normal(size=(ntri, 3, 3)) * he
    if kind == 1:
        g = int(max(2, np.sqrt(ntri))); xs, ys = np.meshgrid(np.linspace(0, 1, g + 1), np.linspace(0, 1, g + 1))
        V = np.stack([xs.ravel(), ys.ravel(), 0.3 + 0.2 * np.sin(5 * xs.ravel()) * np.cos(3 * ys.ravel())], 1)
        q = np.array([[i * (g + 1) + j, i * (g + 1) + j + 1, (i + 1) * (g + 1) + j, i * (g + 1) + j + 1, (i + 1) * (g + 1) + j + 1, (i + 1) * (g + 1) + j] for i in range(g) for j in range(g)]).reshape(-1, 3)
        T = V[q]
    if kind == 2: T[:, :, 2] = np.round(T[:, :, 2] * 4) / 4
    if kind == 3: T[::3, 2] = T[::3, 1]; T[1::7, 2] = T[1::7, 0] + 2.0 * (T[1::7, 1] - T[1::7, 0])
    if kind == 4: T[:, 2] = T[:, 0] + (T[:, 1] - T[:, 0]) * 1.0000001 + rng.normal(size=(T.shape[0], 3)) * 1e-9
    P = (T.reshape(-1, 3) * scale + shift).astype(np.float64); idx = np.arange(P.shape[0], dtype=np.uint32)
    n = 20000
    tri = P.reshape(-1, 3, 3); pick = rng.integers(0, tri.shape[0], n)
    w = rng.random((n, 3)); w /= w.sum(1, keepdims=True); tgt = (tri[pick] * w[:, :, None]).sum(1)
    tgt[:n // 4] = tri[pick[:n // 4], rng.integers(0, 3, n // 4)]
    tgt[n // 4:n // 2] = 0.5 * (tri[pick[n // 4:n // 2], 0] + tri[pick[n // 4:n // 2], 1])
    org = tgt + rng.normal(size=(n, 3)) * scale * float(rng.choice([0.1, 1.0, 30.0]))
    dr = (tgt - org) * rng.uniform(0.001, 1000.0, (n, 1))
    dr[-n // 8:] = rng.normal(size=(n // 8, 3))
    k8 = n // 8
    dr[k8:2 * k8, 0] = 0.0; dr[2 * k8:3 * k8, 2] = 0.0; dr[2 * k8:2 * k8 + k8 // 2, 0] = 0.0
    org[3 * k8:3 * k8 + k8 // 2] = tri[pick[3 * k8:3 * k8 + k8 // 2], 0]; dr[3 * k8:3 * k8 + k8 // 2] = rng.normal(size=(k8 // 2, 3))
    ok = np.abs(dr[:, 1]) > 1e-14 * np.abs(dr).max(1)
    org, dr = np.ascontiguousarray(org[ok]), np.ascontiguousarray(dr[ok])
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org, dr, nthreads=8)
    m = Model(P, idx); m.ref_build()
    got = m.hostwalk(org, dr)
    for k in range(4):
        bad = np.nonzero(np.asarray(got[k]) != np.asarray(exp[k]))[0]
        if bad.size: print("MISMATCH round", r, "kind", kind, "field", k, bad.size, bad[:5]); sys.exit(1)
    Model.ref_off(); total += org.shape[0]
print(total, "rays over", rounds, "scenes: the host walk's records equal the oracle's")
