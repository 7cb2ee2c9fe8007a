"""Scene helpers for benches/tests (inputs only; no ray arithmetic here)."""
import numpy as np


def tessellate(positions, indices, levels):
    """Midpoint subdivision: every triangle -> 4, `levels` times (BASELINE config 5's
    "tessellated RIB": deterministic, no vertex sharing, so triangle i -> 4i..4i+3).
    positions [n,3] float64, indices [3m] -> (positions', indices')"""
    P = np.asarray(positions, np.float64)
    tri = P[np.asarray(indices, np.int64).reshape(-1, 3)]          # [m,3,3]
    for _ in range(levels):
        a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
        ab, bc, ca = 0.5 * (a + b), 0.5 * (b + c), 0.5 * (c + a)
        tri = np.stack([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1),
                        np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)], 1).reshape(-1, 3, 3)
    Pn = np.ascontiguousarray(tri.reshape(-1, 3))
    return Pn, np.arange(Pn.shape[0], dtype=np.uint32)


def load_fixture_scene(npz):
    """geoms [(positions, indices, normals|None, two_side)] + camera(20 doubles) from a golden fixture"""
    g = np.load(npz) if isinstance(npz, str) else npz
    geoms = []
    for k in range(int(g["ngeoms"])):
        n = g["nrm%d" % k] if ("nrm%d" % k) in g.files else None
        geoms.append((g["pos%d" % k], g["idx%d" % k], n, int(g["two_side%d" % k]) if ("two_side%d" % k) in g.files else 0))
    return geoms, g["camera"]
