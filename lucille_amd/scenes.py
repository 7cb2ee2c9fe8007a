"""Scene helpers for benches/tests (inputs only; no ray arithmetic here)."""
import ctypes as C

import numpy as np

SOUP_SEED = 88172645463325252


def soup_triangles(ntriangles, half_extent, state=SOUP_SEED):
    """S-soup triangles (lh_synth_soup_triangles): -> (positions [3n,3] float64, indices [3n] uint32,
    the stream state after the last triangle -- the ray dump continues from it)"""
    from . import binding
    st = C.c_uint64(int(state))
    P = np.empty((3 * ntriangles, 3), np.float64); idx = np.empty(3 * ntriangles, np.uint32)
    binding.lib().lh_synth_soup_triangles(C.byref(st), int(ntriangles), float(half_extent), P.ctypes.data, idx.ctypes.data)
    return P, idx, int(st.value)


def soup_rays(n, state, org=None, dr=None):
    """the next n rays of the stream (lh_synth_soup_rays) -> (org [n,3], dir [n,3], new state);
    org / dr: optional preallocated float64 arrays of at least n rows"""
    from . import binding
    st = C.c_uint64(int(state))
    if org is None:
        org = np.empty((n, 3), np.float64)
    if dr is None:
        dr = np.empty((n, 3), np.float64)
    assert org.flags.c_contiguous and dr.flags.c_contiguous and org.shape[0] >= n and dr.shape[0] >= n
    binding.lib().lh_synth_soup_rays(C.byref(st), int(n), org.ctypes.data, dr.ctypes.data)
    return org[:n], dr[:n], int(st.value)


def skip(state, ndraws):
    """the stream state after `ndraws` more uniforms (lh_synth_skip; a ray is 5 draws, a triangle 12)"""
    from . import binding
    st = C.c_uint64(int(state))
    binding.lib().lh_synth_skip(C.byref(st), int(ndraws))
    return int(st.value)


def soup(ntriangles, nrays, half_extent, state=SOUP_SEED):
    P, idx, st = soup_triangles(ntriangles, half_extent, state)
    org, dr, _ = soup_rays(nrays, st)
    return P, idx, org, dr


def tessellate(positions, indices, levels):
    """Midpoint subdivision: every triangle -> 4, `levels` times (BASELINE config 5's
    "tessellated RIB": deterministic, no vertex sharing, so triangle i -> 4i..4i+3).
    positions [n,3] float64, indices [3m] -> (positions', indices')"""
    from . import binding
    P = np.asarray(positions, np.float64)[:, :3]
    tri = np.ascontiguousarray(P[np.asarray(indices, np.int64).reshape(-1, 3)])          # [m,3,3]
    out = np.empty((tri.shape[0] * 4 ** int(levels), 3, 3), np.float64)
    binding.lib().lh_synth_tessellate(tri.ctypes.data, tri.shape[0], int(levels), out.ctypes.data)
    Pn = out.reshape(-1, 3)
    return Pn, np.arange(Pn.shape[0], dtype=np.uint32)


def load_fixture_scene(npz):
    """geoms [(positions, indices, normals|None, two_side)] + camera(20 doubles) from a golden fixture"""
    g = np.load(npz) if isinstance(npz, str) else npz
    geoms = []
    for k in range(int(g["ngeoms"])):
        n = g["nrm%d" % k] if ("nrm%d" % k) in g.files else None
        geoms.append((g["pos%d" % k], g["idx%d" % k], n, int(g["two_side%d" % k]) if ("two_side%d" % k) in g.files else 0))
    return geoms, g["camera"]
