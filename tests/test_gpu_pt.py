"""Wavefront path tracer (BASELINE config 4's scene examples/plane_sphere).  The
reference's pathtrace.c is dead code, so there is no reference image: parity is at the ray level
(every bounce goes through the closest-hit kernel the parity tests pin) and the image is checked
by invariants: furnace test, one-bounce == ambient occlusion, convergence, determinism."""
import numpy as np
import pytest

import lucille_amd as la
from oracle import pyoracle as po
from tests.helpers import load_golden
from tests.test_gpu_ao import load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ps():
    return load_case("ao_ps")


def render(acc, cam, spp, tile=None, **kw):
    import torch
    W, H = cam.width, cam.height
    tile = tile or W
    img = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    rays = 0
    for y0 in range(0, H, tile):
        for x0 in range(0, W, tile):
            w, h = min(tile, W - x0), min(tile, H - y0)
            rgb, st = acc.render_pt_tile(cam, x0, y0, w, h, 0, spp, spp, **kw)
            img[H - (y0 + h):H - y0, x0:x0 + w] = rgb
            rays += st["rays"]
    return img.cpu().numpy(), rays


def test_furnace(ps):
    """reflectance 1, white environment, generous vertex limit: every path eventually leaves the
    (open) scene carrying throughput 1 -> every pixel is exactly the environment radiance"""
    img, rays = render(ps["acc"], ps["cam"], 16, kd=1.0, env=(0.25, 0.5, 1.0), max_vertices=400)
    assert rays >= 96 * 96 * 16
    ok = np.isclose(img, np.array([0.25, 0.5, 1.0], np.float32)[None, None, :], rtol=0, atol=1e-6).all(-1)
    # a few silhouette samples bounce INTO the closed sphere (interpolated shading normal vs the
    # facet) and die at the vertex limit
    assert ok.mean() > 0.995, ok.mean()
    assert img.max() <= 1.0 + 1e-6


def test_one_bounce_equals_ambient_occlusion(ps):
    """max 3 path vertices (camera, hit, one bounce) with reflectance 1 is the AO estimator:
    the frame's mean must match the reference's own AO frame of the same scene"""
    img, _ = render(ps["acc"], ps["cam"], 64, kd=1.0, env=(1, 1, 1), max_vertices=3)
    ref = ps["g"]["image"]
    hitmask = ref[..., 0] > 0
    # background pixels: AO writes 0 on a miss, the path tracer sees the environment
    assert abs(img[hitmask].mean() - ref[hitmask].mean()) < 0.01


def test_energy_and_convergence(ps):
    acc, cam = ps["acc"], ps["cam"]
    a16, _ = render(acc, cam, 16, kd=0.8, seed=1); b16, _ = render(acc, cam, 16, kd=0.8, seed=2)
    a256, _ = render(acc, cam, 256, kd=0.8, seed=1); b256, _ = render(acc, cam, 256, kd=0.8, seed=2)
    assert a256.max() <= 1.0 + 1e-5 and a256.min() >= 0.0
    r16 = np.sqrt(((a16 - b16) ** 2).mean()); r256 = np.sqrt(((a256 - b256) ** 2).mean())
    assert r256 < 0.4 * r16, (r16, r256)          # Monte-Carlo 1/sqrt(n): expect ~0.25
    dark, _ = render(acc, cam, 64, kd=0.3, seed=1)
    assert dark.mean() < a256.mean()               # less reflectance, less light


def test_deterministic_and_tiling_independent(ps):
    acc, cam = ps["acc"], ps["cam"]
    a, ra = render(acc, cam, 8, kd=0.7, seed=5)
    b, rb = render(acc, cam, 8, kd=0.7, seed=5)
    c, rc = render(acc, cam, 8, tile=32, kd=0.7, seed=5)
    assert np.array_equal(a, b) and ra == rb
    assert np.array_equal(a, c) and ra == rc       # the RNG is keyed by absolute pixel / sample / bounce
    # accumulating samples in two passes == one pass (spp_begin / spp_total contract)
    import torch
    out = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    acc.render_pt_tile(cam, 0, 0, cam.width, cam.height, 0, 4, 8, kd=0.7, seed=5, out=out)
    acc.render_pt_tile(cam, 0, 0, cam.width, cam.height, 4, 4, 8, kd=0.7, seed=5, out=out)
    assert np.allclose(out.cpu().numpy(), a, atol=1e-6)


# ---- the reference's three reflection types, roulette on kd + ks + kt, IBL on a miss (pathtrace.c:189-314,407-537) ----

def render2(acc, cam, spp, max_vertices=8, flags=0, seed=1):
    rgb, st = acc.render_pt_tile2(cam, 0, 0, cam.width, cam.height, 0, spp, spp, max_vertices=max_vertices, flags=flags, seed=seed)
    return rgb.cpu().numpy(), st


def ibl_fetch_numpy(envmap, d):
    """ri_texture_ibl_fetch + ri_texture_fetch (texture.c:86-180,238-276) restated with numpy for [n,3] directions"""
    H, W = envmap.shape[:2]
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    r = np.where((d[:, 2] >= -1.0) & (d[:, 2] < 1.0), np.arccos(np.clip(d[:, 2], -1, 1)) / 3.1415926535, 0.0)
    n2 = d[:, 0] ** 2 + d[:, 1] ** 2
    r = np.where(n2 > 1e-6, r / np.sqrt(np.maximum(n2, 1e-300)), r)
    u = 0.5 * d[:, 0] * r + 0.5; v = 0.5 - 0.5 * d[:, 1] * r
    u = np.clip(u - np.floor(u), 0, 1); v = np.clip(v - np.floor(v), 0, 1)
    px = u * (W - 1); py = v * (H - 1)
    x = px.astype(int); y = py.astype(int); fx = px - x; fy = py - y
    x1 = np.minimum(x + 1, W - 1); y1 = np.minimum(y + 1, H - 1)
    w0 = ((1 - fx) * (1 - fy))[:, None]; w1 = ((1 - fx) * fy)[:, None]; w2 = (fx * (1 - fy))[:, None]; w3 = (fx * fy)[:, None]
    return (w0 * envmap[y, x, :3] + w1 * envmap[y1, x, :3] + w2 * envmap[y, x1, :3] + w3 * envmap[y1, x1, :3])


def test_ibl_lookup_on_a_miss(ps):
    """camera rays that leave the scene return the light probe's radiance in their direction"""
    import torch
    acc, cam = ps["acc"], ps["cam"]
    rng = np.random.default_rng(3)
    envmap = rng.uniform(0.1, 2.0, (32, 48, 4)).astype(np.float32)
    acc.set_environment((1.0, 0.5, 2.0), envmap)
    acc.set_material(la.ALL_MESHES, la.Material.make(kd=(0.5, 0.5, 0.5)))
    img, st = render2(acc, cam, 1, max_vertices=2)          # 2 vertices: camera + first hit; hits end with 0
    org, dr = acc.primary_rays(cam, 0, 0, cam.width, cam.height, 1)
    prim = acc.intersect_device(org, dr)[0].cpu().numpy()
    miss = (prim == -1).reshape(cam.height, cam.width)[::-1]
    assert 0.1 < miss.mean() < 0.9
    # away from silhouettes (the path tracer jitters inside the pixel): 3x3 neighbourhoods that agree
    def core(mask):
        c = mask.copy()
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                c &= np.roll(np.roll(mask, dy, 0), dx, 1)
        c[0] = c[-1] = False; c[:, 0] = c[:, -1] = False
        return c
    hit_core, miss_core = core(~miss), core(miss)
    assert (img[hit_core] == 0).all() and (img[miss_core] > 0).all()
    # the path tracer jitters inside the pixel: compare with the probe at the pixel-centre direction, loosely, and
    # exactly at the level of "some probe value scaled by rgb"
    d = dr.cpu().numpy().reshape(cam.height, cam.width, 3)[::-1][miss_core]
    exp = ibl_fetch_numpy(envmap.astype(np.float64), d) * np.array([1.0, 0.5, 2.0])
    got = img[miss_core]
    assert np.abs(got - exp).mean() < 0.05 * exp.mean()      # sub-pixel jitter moves the lookup by < 1 texel
    acc.set_environment((1.0, 1.0, 1.0), None)
    acc.set_material(la.ALL_MESHES, la.Material.make())


def test_transmission_with_ior_one_is_invisible(ps):
    """kt = 1, ior = 1: every surface lets the ray through unbent (ri_refract with eta 1), so every path leaves the
    scene with throughput 1: the frame is the environment everywhere -- and the interior flag toggles per crossing"""
    acc, cam = ps["acc"], ps["cam"]
    acc.set_environment((0.25, 0.5, 1.0), None)
    acc.set_material(la.ALL_MESHES, la.Material.make(kd=(0, 0, 0), kt=(1, 1, 1), ior=1.0))
    img, st = render2(acc, cam, 4, max_vertices=64)
    ok = np.isclose(img, np.array([0.25, 0.5, 1.0], np.float32)[None, None, :], rtol=0, atol=1e-6).all(-1)
    # (a few silhouette samples where the interpolated shading normal and the facet disagree about the side the ray
    # continues on re-hit their own triangle until the vertex limit, as in test_furnace)
    assert ok.mean() > 0.99 and img.max() <= 1.0 + 1e-6
    assert st["rays"] > st["paths"]                           # paths really crossed surfaces
    acc.set_environment((1.0, 1.0, 1.0), None); acc.set_material(la.ALL_MESHES, la.Material.make())


def test_mirror_and_glass_conserve_energy_and_roulette_is_unbiased(ps):
    acc, cam = ps["acc"], ps["cam"]
    acc.set_environment((1.0, 1.0, 1.0), None)
    # perfect mirror: no roulette loss (ks = 1); every escaping path carries 1
    acc.set_material(la.ALL_MESHES, la.Material.make(kd=(0, 0, 0), ks=(1, 1, 1)))
    img, _ = render2(acc, cam, 8, max_vertices=200)
    assert img.max() <= 1.0 + 1e-6 and img.mean() > 0.97
    # glass sphere over a diffuse plane: mesh materials differ; energy stays below the white furnace
    g = ps["g"]
    nm = int(g["ngeoms"])
    for k in range(nm):
        acc.set_material(k, la.Material.make(kd=(0.2, 0.2, 0.2), ks=(0.1, 0.1, 0.1), kt=(0.7, 0.7, 0.7), ior=1.5) if k == nm - 1
                         else la.Material.make(kd=(0.7, 0.6, 0.5)))
    a, sa = render2(acc, cam, 64, seed=1); b, sb = render2(acc, cam, 64, seed=2)
    # (roulette runs on the channel AVERAGE: a coloured reflectance weighs single channels by kd_c / mean(kd), so a
    # 64-sample pixel mean may sit a little above 1; the frame may not)
    assert a.max() <= 1.25 and 0.2 < a.mean() < 1.0
    assert abs(a.mean() - b.mean()) < 0.01 and not np.array_equal(a, b)
    a2, _ = render2(acc, cam, 64, seed=1)
    assert np.array_equal(a, a2)                              # deterministic in (seed, sample, bounce)
    # the reference's own factors (brdf value, no pdf): darker by construction (kd / pi per diffuse bounce)
    c, _ = render2(acc, cam, 64, flags=la.PT_REFERENCE_WEIGHTS, seed=1)
    assert c.mean() < a.mean()
    # unbiased roulette: raising ks + kt at equal albedo sum must not change a white furnace
    acc.set_material(la.ALL_MESHES, la.Material.make(kd=(0.5, 0.5, 0.5), ks=(0.25, 0.25, 0.25), kt=(0.25, 0.25, 0.25), ior=1.0))
    f, _ = render2(acc, cam, 16, max_vertices=400)
    ok = np.isclose(f, 1.0, atol=1e-5).all(-1)
    assert ok.mean() > 0.99
    with pytest.raises(la.LucilleHipError, match="exceed 1"):
        acc.set_material(0, la.Material.make(kd=(1, 1, 1), ks=(0.5, 0.5, 0.5)))
    acc.set_material(la.ALL_MESHES, la.Material.make())


def test_interleaved_bands_as_one_pass_equal_the_frame(ps):
    """lh_render_pt_bands: the bands three ranks would own (4 lines each, band_id % 3 == rank), each rank's as ONE pass, put
    back together == the frame rendered as one tile, bit for bit (keys are (frame pixel, sample, bounce)); ray counts add up"""
    import torch
    acc, cam = ps["acc"], ps["cam"]
    W, H = cam.width, cam.height
    assert H % 4 == 0
    acc.set_environment((0.9, 0.8, 0.7), None)
    ref, st = acc.render_pt_tile(cam, 0, 0, W, H, 0, 8, 8, kd=0.75, env=(0.9, 0.8, 0.7), max_vertices=6, seed=4)
    nb = H // 4
    img = torch.zeros((nb, 4, W, 3), dtype=torch.float32, device="cuda"); rays = 0
    for r in range(3):
        mine = list(range(r, nb, 3))
        out, s = acc.render_pt_bands(cam, 4 * r, 4, 12, len(mine), 0, 8, 8, max_vertices=6, override=la.Material.make(kd=(0.75,) * 3), seed=4)
        img[r::3] = out; rays += s["rays"]
    torch.cuda.synchronize()
    acc.set_environment((1.0, 1.0, 1.0), None)
    assert torch.equal(img.flip(0).reshape(H, W, 3), ref) and rays == st["rays"]
    with pytest.raises(la.LucilleHipError, match="inside the frame"):
        acc.render_pt_bands(cam, 4, 4, 12, nb // 3 + 1, 0, 8, 8)


def test_sharded_frame_arguments_mean_the_same_on_every_path(ps):
    """render.render_pt_frame_sharded at world 1: material / flags are honoured whatever path the frame takes (they used to be
    dropped on the tile path), kd == a diffuse material of that reflectance, env is this frame's environment on both paths and
    the accelerator's own environment is put back afterwards, an explicit black environment is black (it used to turn white)"""
    import torch
    from lucille_amd import render as R
    acc, cam = ps["acc"], ps["cam"]
    acc.set_environment((0.25, 0.5, 1.0), None)                                     # the accelerator's own: must survive
    mat = la.Material.make(kd=(0.6,) * 3)
    env = (1.0, 0.9, 0.8)
    a, sa = R.render_pt_frame_sharded(acc, cam, 8, 0, 1, tile=48, spp_chunk=4, kd=0.6, env=env, max_vertices=5, seed=2)        # tiles
    b, sb = R.render_pt_frame_sharded(acc, cam, 8, 0, 1, tile=48, spp_chunk=4, material=mat, env=env, max_vertices=5, seed=2)  # -> bands
    c, sc = R.render_pt_frame_sharded(acc, cam, 8, 0, 1, spp_chunk=4, band_rows=4, kd=0.6, env=env, max_vertices=5, seed=2)     # bands asked for
    assert torch.equal(a, b) and torch.equal(a, c) and sa == sb == sc
    dark = la.Material.make(kd=(0.2,) * 3)
    d, _ = R.render_pt_frame_sharded(acc, cam, 8, 0, 1, tile=48, spp_chunk=4, material=dark, env=env, max_vertices=5, seed=2)
    assert float(d.mean()) < 0.8 * float(a.mean())                                  # the material was used, not a default kd
    # the accelerator's own environment is back: a frame without env uses it, on both paths
    e, _ = R.render_pt_frame_sharded(acc, cam, 4, 0, 1, tile=48, spp_chunk=4, kd=0.6, max_vertices=5, seed=2)
    f, _ = R.render_pt_frame_sharded(acc, cam, 4, 0, 1, spp_chunk=4, band_rows=4, kd=0.6, max_vertices=5, seed=2)
    g, _ = acc.render_pt_tile(cam, 0, 0, cam.width, cam.height, 0, 4, 4, kd=0.6, env=(0.25, 0.5, 1.0), max_vertices=5, seed=2)
    assert torch.equal(e, f) and torch.equal(e, g)
    # black is black
    k, _ = R.render_pt_frame_sharded(acc, cam, 2, 0, 1, spp_chunk=2, band_rows=4, kd=0.6, env=(0.0, 0.0, 0.0), max_vertices=5, seed=2)
    assert float(k.abs().max()) == 0.0
    # a frame height the band rows do not divide: one-line bands, same frame
    cam2 = la.Camera.make(cam.width, cam.height - 2, cam.flength, np.array(list(cam.cam2world)), cam.rh)
    t1, _ = R.render_pt_frame_sharded(acc, cam2, 4, 0, 1, tile=64, spp_chunk=4, kd=0.6, env=env, max_vertices=5, seed=2)
    t2, _ = R.render_pt_frame_sharded(acc, cam2, 4, 0, 1, spp_chunk=4, band_rows=4, material=mat, env=env, max_vertices=5, seed=2)
    assert torch.equal(t1, t2)
    acc.set_environment((1.0, 1.0, 1.0), None)


def test_empty_and_one_triangle_scenes():
    """no geometry: every camera ray leaves the scene and returns the environment (the bounce chain's ray counts live on the
    device: the launches after the first find zero rays); one triangle with a 2-vertex limit: hits end black"""
    import torch
    cam = la.Camera.make(64, 48, 2.0, np.eye(4).ravel(), 1)
    acc = la.HipAccel(0); acc.commit()
    img, st = acc.render_pt_tile(cam, 0, 0, 64, 48, 0, 4, 4, kd=0.8, env=(0.5, 0.25, 1.0), seed=1)
    assert st == {"paths": 64 * 48 * 4, "rays": 64 * 48 * 4, "max_depth_reached": 1}
    assert torch.equal(img, torch.tensor([0.5, 0.25, 1.0], device="cuda").expand(48, 64, 3))
    out, st = acc.render_pt_bands(cam, 0, 4, 8, 6, 0, 4, 4, seed=1)           # the accelerator's default environment: white
    assert out.shape == (6, 4, 64, 3) and float(out.min()) == 1.0 and st["rays"] == 6 * 4 * 64 * 4
    acc.close()
    P = np.array([[0, 0, -5], [1, 0, -5], [0, 1, -5]], float)
    acc = la.HipAccel(0); acc.add_mesh(P, np.arange(3, dtype=np.uint32)); acc.commit()
    img, st = acc.render_pt_tile(cam, 0, 0, 64, 48, 0, 4, 4, kd=0.8, env=(1, 1, 1), seed=1, max_vertices=2)
    assert st["rays"] == st["paths"] and 0.9 < float(img.mean()) < 1.0 and float(img.min()) == 0.0
    acc.close()


def test_pixel_sums_are_fixed_point(ps):
    """a path's radiance goes into its pixel's three 64-bit fixed-point sums (32 fraction bits) where the path ends: integer sums do
    not depend on the order of their terms -- a pixel does not depend on slot order, tiling or how the samples are cut into
    passes beyond the one float addition per pass -- a sample is clamped to +-2^18, a NaN is dropped, and one pass holds at most
    4096 samples of a pixel"""
    import torch
    cam = la.Camera.make(32, 16, 2.0, np.eye(4).ravel(), 1)
    acc = la.HipAccel(0); acc.commit()                       # no geometry: every sample is the environment, exactly
    env = (0.3, 1.0e-7, 3.0)                                 # not dyadic: 0.3f = 10066330 x 2^-25 is a 32-fraction-bit number all the same
    img, _ = acc.render_pt_tile(cam, 0, 0, 32, 16, 0, 64, 64, kd=0.8, env=env, seed=1)
    # 64 equal samples summed exactly, converted to fp32 once, times 1 / 64 (exact): the environment itself -- the 1e-7 channel to
    # the 2^-32 of its quantum (floor)
    got = img.cpu().numpy()
    assert (got[..., 0] == np.float32(0.3)).all() and (got[..., 2] == np.float32(3.0)).all()
    q = np.floor(np.float64(np.float32(1.0e-7)) * 2.0 ** 32) / 2.0 ** 32
    assert np.allclose(got[..., 1], q, rtol=1e-6, atol=0)
    # the clamp and the NaN rule
    big, _ = acc.render_pt_tile(cam, 0, 0, 32, 16, 0, 4, 4, kd=0.8, env=(1.0e9, -1.0e9, float("nan")), seed=1)
    big = big.cpu().numpy()
    assert (big[..., 0] == 262144.0).all() and (big[..., 1] == -262144.0).all() and (big[..., 2] == 0.0).all()
    with pytest.raises(Exception, match="4096 samples"):
        acc.render_pt_tile(cam, 0, 0, 4, 4, 0, 4097, 4097, kd=0.8, env=(1, 1, 1), seed=1)
    out, _ = acc.render_pt_tile(cam, 0, 0, 4, 4, 0, 4096, 4096, kd=0.8, env=(0.5, 1.0e9, -1.0e9), seed=1)      # the largest pass at the clamp: no overflow
    assert torch.equal(out, torch.tensor([0.5, 262144.0, -262144.0], device="cuda").expand(4, 4, 3))
    acc.close()
