"""Host logic, fuzzed: the kernel algorithm's host model (shared arithmetic, lh_filter.h) against
the oracle over randomly scaled / translated / degenerate scenes and awkward rays.  The filter must
stay conservative (never lose a hit) whatever the magnitudes: bit-exact records, ties included
(the model runs with the reference-order tree)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st, HealthCheck

from oracle import pyoracle as po
from tests.helpers import Model, assert_hits_equal


def make_scene(rng, ntri, scale, offset, kind):
    if kind == "soup":
        c = rng.uniform(0, 1, (ntri, 1, 3)); P = (c + rng.uniform(-0.05, 0.05, (ntri, 3, 3))).reshape(-1, 3)
    elif kind == "slivers":      # long thin triangles: tiny determinants
        c = rng.uniform(0, 1, (ntri, 1, 3)); d = rng.normal(size=(ntri, 1, 3))
        P = (c + d * rng.uniform(-0.3, 0.3, (ntri, 3, 1)) + rng.normal(size=(ntri, 3, 3)) * 1e-5).reshape(-1, 3)
    elif kind == "axis":         # axis-aligned planes with shared vertices (zero-thickness boxes)
        n = max(2, int(np.sqrt(ntri / 2)))
        xs = np.linspace(0, 1, n + 1); P = np.array([[x, y, 0.37] for y in xs for x in xs])
        idx = []
        for j in range(n):
            for i in range(n):
                a = j * (n + 1) + i; idx += [a, a + 1, a + n + 2, a, a + n + 2, a + n + 1]
        return P * scale + offset, np.array(idx, np.uint32)
    else:                        # degenerate mix: zero-area, duplicated, collinear
        P = rng.uniform(0, 1, (ntri * 3, 3))
        k = min(len(P[::7]), len(P[1::7])); P[::7][:k] = P[1::7][:k]
        if ntri >= 2:
            P[3:6] = P[0:3]
    return P * scale + offset, np.arange(P.shape[0], dtype=np.uint32)


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(seed=st.integers(0, 2 ** 31), ntri=st.integers(1, 400), log_scale=st.floats(-3, 3), off=st.floats(-1e3, 1e3),
       kind=st.sampled_from(["soup", "slivers", "axis", "degenerate"]), far=st.floats(0.0, 50.0))
def test_model_equals_oracle_under_fuzz(seed, ntri, log_scale, off, kind, far):
    rng = np.random.default_rng(seed)
    scale = 10.0 ** log_scale
    offset = np.array([off, -0.5 * off, 0.25 * off])
    P, idx = make_scene(rng, ntri, scale, offset, kind)
    n = 1500
    tgt = rng.uniform(-0.1, 1.1, (n, 3)) * scale + offset
    org = tgt + rng.normal(size=(n, 3)) * scale * (0.5 + far)
    # a share of rays aimed EXACTLY at vertices / edge midpoints / centroids, and of origins lying on a
    # triangle or a vertex (t == 0): hits on the box faces of the reference's own tree, which its fp64 box
    # test may or may not let through -- the kernel must answer what the reference answers
    T = P[idx].reshape(-1, 3, 3)
    # ... except at zero-area triangles: there the reference's det is rounding noise that can pass its
    # |det| > 1e-14 test, t = u = v = -0.0 comes out for a ray through the doubled vertex, and whether
    # it keeps that "hit" depends on its visiting order (outside the contract)
    area = np.linalg.norm(np.cross(T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]), axis=1)
    proper = np.nonzero(area > 1e-9 * scale * scale)[0]
    if len(proper) == 0:
        proper = np.arange(T.shape[0])
    pick = proper[rng.integers(0, len(proper), n)]
    tgt[::5] = T[pick[::5], rng.integers(0, 3, len(pick[::5]))]
    tgt[1::7] = 0.5 * (T[pick[1::7], 0] + T[pick[1::7], 1])
    tgt[2::9] = T[pick[2::9]].mean(axis=1)
    org[3::23] = T[pick[3::23], 2]
    org[4::29] = T[pick[4::29]].mean(axis=1)
    dr = tgt - org
    # a share of axis-parallel / unnormalised / tiny-component directions
    dr[::11, 0] = 0.0; dr[1::13, 1] = 0.0; dr[2::17] *= 1e-3; dr[3::19, 2] = 1e-20
    keep = np.linalg.norm(dr, axis=1) > 0
    org, dr = org[keep], dr[keep]
    # the reference leaves invdir[1] unset for |dir.y| <= 1e-14 (bvh.c:483-487): excluded by contract
    ok = np.abs(dr[:, 1]) > 1e-14
    org, dr = org[ok], dr[ok]
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    m = Model(P, idx, nthreads=1)
    m.ref_build(nthreads=1, use_for_ties=True)
    try:
        exp = o.intersect(org, dr)
        for q in (4, 2, 0):
            got, _ = m.trace(org, dr, qnodes=q, nthreads=1)
            assert_hits_equal(got, exp, "%s scale %g off %g fmt %d" % (kind, scale, off, q))
        occ, _ = m.trace(org, dr, anyhit=True, nthreads=1)
        assert np.array_equal(occ.astype(bool), exp[0] != po.MISS)
    finally:
        Model.ref_off()
