"""The path-traced tile against oracle/lucille_oracle_pt.c: the plain-C, one-path-at-a-time restatement
of lh_pt.h with the same counter-based keys.  BASELINE config 4's scene (examples/plane_sphere, tests/golden/ao_ps.npz) at
96 x 96 x 16 spp: the number of rays traced and the longest path are equal, the frame agrees to 1e-6 (the resolve kernel may
contract `sum * 1/spp + old` into an fma; libm's sin/cos vs the device's may move a direction by an ulp)."""
import numpy as np
import pytest

import lucille_amd as la
from tests.test_gpu_ao import load_case

pytestmark = pytest.mark.gpu

SPP = 16


@pytest.fixture(scope="module")
def ps():
    return load_case("ao_ps")


def mat10(m):
    return [m.kd[0], m.kd[1], m.kd[2], m.ks[0], m.ks[1], m.ks[2], m.kt[0], m.kt[1], m.kt[2], m.ior]


def close(a, b):
    return np.abs(a - b) <= 1e-6 * np.maximum(1.0, np.abs(b))


def test_diffuse_paths_equal_the_oracle(ps):
    acc, cam, o, ocam = ps["acc"], ps["cam"], ps["oracle"], ps["ocam"]
    W, H = cam.width, cam.height
    assert (W, H) == (96, 96)
    for kd, mv, seed in ((0.8, 8, 1), (1.0, 40, 7)):
        img, st = acc.render_pt_tile(cam, 0, 0, W, H, 0, SPP, SPP, max_vertices=mv, kd=kd, env=(0.9, 0.8, 0.7), seed=seed)
        exp, est, per = o.render_pt(ocam, 0, 0, W, H, 0, SPP, SPP, max_vertices=mv, override=[kd] * 3 + [0.0] * 6 + [1.0],
                                    env_rgb=(0.9, 0.8, 0.7), seed=seed)
        assert st == est, (st, est)                                     # paths, rays, longest path: equal
        assert per.max() == st["max_depth_reached"] and per.max() > 3
        got = img.cpu().numpy()
        assert close(got, exp).all(), float(np.abs(got - exp).max())


def test_tiles_and_sample_passes_equal_the_oracle(ps):
    """a ragged tile of the frame, samples 4 .. 11 of 16: the keys are absolute (frame pixel, sample)"""
    acc, cam, o, ocam = ps["acc"], ps["cam"], ps["oracle"], ps["ocam"]
    x0, y0, w, h = 37, 20, 41, 33
    img, st = acc.render_pt_tile(cam, x0, y0, w, h, 4, 8, 16, max_vertices=8, kd=0.6, env=(1, 1, 1), seed=3)
    exp, est, _ = o.render_pt(ocam, x0, y0, w, h, 4, 8, 16, max_vertices=8, override=[0.6] * 3 + [0.0] * 6 + [1.0], seed=3)
    assert st == est
    assert close(img.cpu().numpy(), exp).all()


def test_glass_mirror_and_light_probe_equal_the_oracle(ps):
    """per-mesh materials (the sphere: glass with a diffuse and a mirror part; the plane: coloured diffuse), an angular-map
    light probe, both weightings"""
    acc, cam, o, ocam, g = ps["acc"], ps["cam"], ps["oracle"], ps["ocam"], ps["g"]
    W, H = cam.width, cam.height
    nm = int(g["ngeoms"])
    rng = np.random.default_rng(11)
    envmap = rng.uniform(0.1, 2.0, (32, 48, 4)).astype(np.float32)
    mats = [la.Material.make(kd=(0.2, 0.2, 0.2), ks=(0.1, 0.1, 0.1), kt=(0.7, 0.7, 0.7), ior=1.5) if k == nm - 1
            else la.Material.make(kd=(0.7, 0.6, 0.5)) for k in range(nm)]
    try:
        acc.set_environment((1.0, 0.5, 2.0), envmap)
        for k, m in enumerate(mats):
            acc.set_material(k, m)
        for flags in (0, la.PT_REFERENCE_WEIGHTS):
            img, st = acc.render_pt_tile2(cam, 0, 0, W, H, 0, SPP, SPP, max_vertices=12, flags=flags, seed=5)
            exp, est, per = o.render_pt(ocam, 0, 0, W, H, 0, SPP, SPP, max_vertices=12, materials=[mat10(m) for m in mats],
                                        env_rgb=(1.0, 0.5, 2.0), env_map=envmap, ref_weights=flags, seed=5)
            assert st == est, (st, est)
            got = img.cpu().numpy()
            ok = close(got, exp)
            # the probe lookup is continuous in the direction: an ulp in sin / cos moves a texel weight by ~1e-16
            assert ok.all(), (float(np.abs(got - exp).max()), int((~ok).sum()))
        assert st["rays"] > 1.3 * st["paths"] and per.max() >= 6      # refraction chains through the sphere really happen
    finally:
        acc.set_environment((1.0, 1.0, 1.0), None); acc.set_material(la.ALL_MESHES, la.Material.make())
