"""The beam-raster path, SURVEY 8f-4: ri_beam_set + ri_raster_plane_setup + ri_bvh_intersect_beam
(/root/reference/src/render/bvh.c:544-609 -> :2547-2643, :2315-2426, :2751-2820; beam.c:469-730; raster.c:166-435).

The reference left this path unfinished and never calls it; what it computes (plane->t of the raster window) is pinned
here quirk for quirk: goldens = the COMPILED reference run in a child process (tests/golden/make_golden.py
--beam-raster), oracle = oracle/lucille_oracle_beam.c, product = lh_beam.hip through the C ABI.  fp64, bit-exact.
Where the reference is undefined (it writes plane->t unchecked for beams that leave the window, its asserts abort) the
product and the oracle report flags instead; those cases are compared product-vs-oracle only."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import RASTER_GOLDEN_CASES, load_golden, raster_case


def oracle_planes(c):
    o = po.Oracle(); o.add_mesh(c["P"], c["idx"]); o.build()
    n = c["org"].shape[0]
    rc = np.empty(n, np.int32); t = np.zeros((n, c["height"], c["width"])); fl = np.zeros((n, 4), np.uint64)
    for i in range(n):
        rc[i], ti, fl[i] = o.beam_raster(c["org"][i], c["dirs"][i], c["width"], c["height"], c["frame"], c["corners"][i],
                                        c["eye"], c["fov"])
        if rc[i] == 0:
            t[i] = ti
    return rc, t, fl


def test_oracle_matches_reference_golden():
    g = load_golden("beam_raster")
    assert int(g["ncases"]) == len(RASTER_GOLDEN_CASES)
    written = 0
    for k, kw in enumerate(RASTER_GOLDEN_CASES):
        c = raster_case(**kw)
        rc, t, fl = oracle_planes(c)
        assert np.array_equal(rc, g["rc%d" % k])
        assert np.array_equal(t, g["t%d" % k])                 # every double of every plane
        assert not fl[:, :3].any()                             # nothing undefined happened in the pinned cases
        written += int((t != 0).sum())
    assert written > 5000


@pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref not built")
def test_oracle_matches_live_reference():
    for kw in (dict(seed=101, ntri=120, width=40, height=40, nbeams=10, eye=(0.05, 0.02, -0.1)),
               dict(seed=102, ntri=2500, width=64, height=64, nbeams=6, tri_size=0.2),
               dict(seed=103, ntri=1, width=32, height=32, nbeams=5, tri_size=2.5)):
        c = raster_case(**kw)
        beams = [(c["org"][i], c["dirs"][i], c["width"], c["height"], c["frame"], c["corners"][i], c["eye"], c["fov"])
                 for i in range(c["org"].shape[0])]
        ref = po.ref_beam_raster_child([(c["P"], c["idx"])], beams)
        rc, t, fl = oracle_planes(c)
        for i, r in enumerate(ref):
            assert r is not None, "the reference died on a beam inside the window"
            assert r[0] == rc[i] and np.array_equal(r[1], t[i])
        assert not fl[:, :3].any()


def test_oracle_quirks_and_refusals():
    c = raster_case(seed=5, ntri=200, width=32, height=32, nbeams=1)
    o = po.Oracle(); o.add_mesh(c["P"], c["idx"]); o.build()
    # a beam over the window's centre: its corner directions differ in sign, ri_beam_set refuses (beam.c:352-376)
    from tests.helpers import raster_beam_dirs
    d = raster_beam_dirs(c["corner"], c["frame"], 16, 8, 8)
    rc, t, fl = o.beam_raster(c["eye"], d, 32, 32, c["frame"], c["corner"], c["eye"], c["fov"])
    assert rc == -1 and not t.any()
    # a beam in one quadrant: the beam's whole footprint is written -- the parts OUTSIDE each triangle (bvh.c:2387-2390)
    d = raster_beam_dirs(c["corner"], c["frame"], 14, 1, 1)
    rc, t, fl = o.beam_raster(c["eye"], d, 32, 32, c["frame"], c["corner"], c["eye"], c["fov"])
    assert rc == 0 and (t[1:15, 1:15] != 0).all() and int((t != 0).sum()) < 2 * 14 * 14
    # empty scene: nothing happens
    e = po.Oracle(); e.add_mesh(np.zeros((0, 3)), np.zeros(0, np.uint32)); e.build()
    rc, t, fl = e.beam_raster(c["eye"], d, 32, 32, c["frame"], c["corner"], c["eye"], c["fov"])
    assert rc == 0 and not t.any() and fl[3] == 0
    # a beam that hangs over the window's edge: undefined in the reference (unchecked writes), counted here
    h = raster_case(seed=5, ntri=3, width=32, height=32, nbeams=4, inside=False)
    o2 = po.Oracle(); o2.add_mesh(h["P"], h["idx"]); o2.build()
    cut = 0
    for i in range(4):
        rc, t, fl = o2.beam_raster(h["org"][i], h["dirs"][i], 32, 32, h["frame"], h["corners"][i], h["eye"], h["fov"])
        cut += int(fl[0])
    assert cut > 0


# ---------------------------------------------------------------------------------------------------------------------
# the product (lh_beam.hip, k_beam_raster) through the C ABI
# ---------------------------------------------------------------------------------------------------------------------
def hip_planes(c, **kw):
    import lucille_amd as la
    acc = la.HipAccel(0)
    acc.add_mesh(c["P"], c["idx"])
    acc.commit(**kw)
    out = acc.beam_raster(c["org"], c["dirs"], c["corners"], c["width"], c["height"], c["frame"], c["eye"], c["fov"])
    acc.close()
    return out


@pytest.mark.gpu
def test_hip_matches_reference_golden():
    g = load_golden("beam_raster")
    for k, kw in enumerate(RASTER_GOLDEN_CASES):
        c = raster_case(**kw)
        t, st, fl = hip_planes(c)
        assert np.array_equal(st, g["rc%d" % k])
        assert np.array_equal(t, g["t%d" % k])
        assert not fl[:, :3].any()


@pytest.mark.gpu
@pytest.mark.parametrize("build", ["host", "device"])
def test_hip_matches_oracle(build):
    """seeded cases beyond the goldens, incl. beams over the window's edge (flags equal the oracle's count, planes equal the
    oracle's cut planes), windows wider than a wave, a non-square window, one triangle, a device-built scene"""
    for kw in (dict(seed=201, ntri=500, width=64, height=64, nbeams=16, eye=(0.3, 0.1, 0.2)),
               dict(seed=202, ntri=4000, width=96, height=96, nbeams=8, tri_size=0.15),
               dict(seed=203, ntri=60, width=160, height=80, nbeams=8, fov=70.0),
               dict(seed=204, ntri=1, width=32, height=32, nbeams=6, tri_size=2.0),
               dict(seed=205, ntri=300, width=48, height=48, nbeams=10, inside=False)):
        c = raster_case(**kw)
        rc, t_exp, fl_exp = oracle_planes(c)
        t, st, fl = hip_planes(c, build=build)
        assert np.array_equal(np.where(st == 1, 0, st), rc)      # status 1 = returned before the plane was cleared: rc 0, plane untouched
        assert np.array_equal(t, t_exp)
        assert np.array_equal(fl, fl_exp)


@pytest.mark.gpu
def test_hip_status_codes_and_untouched_planes():
    import lucille_amd as la
    from tests.helpers import raster_beam_dirs
    c = raster_case(seed=5, ntri=200, width=32, height=32, nbeams=1)
    good = raster_beam_dirs(c["corner"], c["frame"], 12, 2, 3)
    straddle = raster_beam_dirs(c["corner"], c["frame"], 16, 8, 8)
    dirs = np.stack([good, straddle, good, good])
    org = np.repeat(c["eye"][None], 4, 0); org[2, 0] += 100.0       # beam 2 starts far to the side: its frustum misses the scene box
    corners = np.repeat(c["corner"][None], 4, 0)
    acc = la.HipAccel(0); acc.add_mesh(c["P"], c["idx"]); acc.commit()
    init = np.full((4, 32, 32), 7.0)
    t, st, fl = acc.beam_raster(org, dirs, corners, 32, 32, c["frame"], c["eye"], c["fov"], t_init=init)
    assert st.tolist() == [0, -1, 1, 0]
    assert (t[1] == 7.0).all() and (t[2] == 7.0).all()             # refused / nothing done: the plane is not touched
    assert np.array_equal(t[0], t[3]) and (t[0] != 7.0).all() and (t[0][3:15, 2:14] != 0).all()
    acc.close()
    e = la.HipAccel(0); e.add_mesh(np.zeros((0, 3)), np.zeros(0, np.uint32)); e.commit()
    t, st, fl = e.beam_raster(org[:1], dirs[:1], corners[:1], 32, 32, c["frame"], c["eye"], c["fov"], t_init=init[:1])
    assert st.tolist() == [1] and (t == 7.0).all()                  # empty accel: returns before anything (bvh.c:560-563)
    e.close()
