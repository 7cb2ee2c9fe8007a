"""oracle/lucille_oracle_pt.c on its own (CPU): the invariants a path tracer must hold, before it is used as the checker of
the device's path-traced tile (tests/test_gpu_pt_oracle.py).  Transport-level parity is UNPINNED (the reference's
pathtrace.c is dead code); each vertex's closest hit is the pinned ri_bvh_intersect restatement."""
import numpy as np

from oracle import pyoracle as po
from tests.helpers import load_golden
from tests.test_oracle_ao import oracle_from_fixture

DIFF = lambda kd: [kd] * 3 + [0.0] * 6 + [1.0]


def scene():
    g = load_golden("ao_ps")
    return oracle_from_fixture(g), po.Camera.from_ref(g["camera"], 48, 48)


def test_uniform_stream_is_a_uniform_stream():
    L = po.lib(); L.lo_pt_rnd.restype = po.C.c_double; L.lo_pt_rnd.argtypes = [po.C.c_uint64]
    x = np.array([L.lo_pt_rnd(k * 64 + 4) for k in range(20000)])
    assert 0.0 <= x.min() and x.max() < 1.0 and abs(x.mean() - 0.5) < 0.01 and abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 0.03


def test_furnace_and_absorption():
    o, cam = scene()
    img, st, per = o.render_pt(cam, 0, 0, 48, 48, 0, 4, 4, max_vertices=400, override=DIFF(1.0), env_rgb=(0.25, 0.5, 1.0))
    ok = np.isclose(img, np.array([0.25, 0.5, 1.0], np.float32), atol=1e-6).all(-1)
    assert ok.mean() > 0.99 and img.max() <= 1.0 + 1e-6                 # reflectance 1: every escaping path carries 1
    assert st["rays"] == int(per.astype(np.int64).sum()) and st["max_depth_reached"] == per.max()
    # unbiased weights: splitting an albedo of 1 over the three lobes (ior 1: refraction goes straight on) leaves the white
    # furnace white -- each survivor is divided by P(type)
    mix = [0.5] * 3 + [0.25] * 3 + [0.25] * 3 + [1.0]
    f, _, _ = o.render_pt(cam, 0, 0, 48, 48, 0, 8, 8, max_vertices=400, override=mix, env_rgb=(1, 1, 1))
    assert np.isclose(f, 1.0, atol=1e-5).all(-1).mean() > 0.99
    # albedo 0.5: roulette ends half the paths at every vertex, the survivors keep throughput 1 -> darker, never above 1
    half, _, _ = o.render_pt(cam, 0, 0, 48, 48, 0, 16, 16, max_vertices=400, override=DIFF(0.5), env_rgb=(1, 1, 1))
    assert 0.4 < half.mean() < 0.9 and half.max() <= 1.0 + 1e-6
    # the reference's own factors (kd / pi per diffuse vertex, no pdf): darker
    ref, _, _ = o.render_pt(cam, 0, 0, 48, 48, 0, 16, 16, max_vertices=8, override=DIFF(0.8), ref_weights=1)
    unb, _, _ = o.render_pt(cam, 0, 0, 48, 48, 0, 16, 16, max_vertices=8, override=DIFF(0.8))
    assert ref.mean() < unb.mean()


def test_frame_is_independent_of_tiling_and_sample_passes():
    o, cam = scene()
    whole, st, _ = o.render_pt(cam, 0, 0, 48, 48, 0, 8, 8, override=DIFF(0.8), seed=9)
    img = np.zeros_like(whole); rays = 0
    for (x0, y0, w, h) in ((0, 0, 20, 48), (20, 0, 28, 17), (20, 17, 28, 31)):
        t, s, _ = o.render_pt(cam, x0, y0, w, h, 0, 8, 8, override=DIFF(0.8), seed=9)
        img[48 - (y0 + h):48 - y0, x0:x0 + w] = t; rays += s["rays"]
    assert np.array_equal(img, whole) and rays == st["rays"]
    two = np.zeros_like(whole)
    o.render_pt(cam, 0, 0, 48, 48, 0, 4, 8, override=DIFF(0.8), seed=9, out=two)
    o.render_pt(cam, 0, 0, 48, 48, 4, 4, 8, override=DIFF(0.8), seed=9, out=two)
    assert np.allclose(two, whole, atol=1e-6)
    other, _, _ = o.render_pt(cam, 0, 0, 48, 48, 0, 8, 8, override=DIFF(0.8), seed=10)
    assert not np.array_equal(other, whole)


def test_transmission_with_ior_one_is_invisible():
    o, cam = scene()
    glass = [0.0] * 6 + [1.0] * 3 + [1.0]
    img, st, per = o.render_pt(cam, 0, 0, 48, 48, 0, 4, 4, max_vertices=64, override=glass, env_rgb=(0.25, 0.5, 1.0))
    ok = np.isclose(img, np.array([0.25, 0.5, 1.0], np.float32), atol=1e-6).all(-1)
    assert ok.mean() > 0.99 and st["rays"] > st["paths"]
