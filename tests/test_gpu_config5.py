"""BASELINE config 5 AS STATED, on the GPU: the AO example scene tessellated to >= 10 M triangles
(midpoint subdivision x 8: 21.1 M triangles, 4.8 GB of trees + triangles in HBM), 4096 x 4096 pixels,
64 AO samples, replicated BVH.  The oracle builds the same 21.1 M-triangle scene (its single-threaded
reference build is most of this module's ~2 minutes) and checks

  * camera rays + hit records of a window of the 4096^2 frame, bit for bit;
  * the hit epilogue and the AO ray dump of that window (fed with caller uniforms, so the rays are
    materialised), bit for bit, and their occlusion against the oracle's closest hits;
  * any-hit == a closest hit exists, over the whole frame's primary rays and the window's AO rays;
  * the full 4096^2 frame: primary hit count == the oracle's over all 16.8 M camera rays, AO ray count,
    tiled == untiled bit-equal, occlusion statistics.
"""
import numpy as np
import pytest

import lucille_amd as la
from lucille_amd import render, scenes
from oracle import pyoracle as po
from tests.helpers import assert_hits_equal, load_golden

pytestmark = pytest.mark.gpu

TESS = 8
SIZE = 4096
NS = 64


@pytest.fixture(scope="module")
def c5():
    g = load_golden("ao_c1")
    acc = la.HipAccel(0); o = po.Oracle(); ntri = 0
    for k in range(int(g["ngeoms"])):
        P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], TESS)
        acc.add_mesh(P, I); o.add_mesh(P, I); ntri += I.shape[0] // 3
        del P, I
    info = acc.commit()
    assert info["ntriangles"] == ntri == 322 * 4 ** TESS >= 10_000_000
    o.build()
    c = g["camera"]
    cam = la.Camera.make(SIZE, SIZE, c[16], c[:16], int(c[19]))
    yield {"acc": acc, "oracle": o, "cam": cam, "info": info}
    acc.close()


def test_window_ray_dumps_bit_exact(c5):
    """primary rays, hit records, epilogue records, AO rays and AO occlusion of a 160 x 96 window in the
    middle of the 4096^2 frame (silhouettes + floor + background) against the oracle"""
    import torch
    acc, o, cam = c5["acc"], c5["oracle"], c5["cam"]
    w, h = 160, 96
    N = 64
    # a window that straddles a silhouette: scan the frame's middle rows at coarse steps for mixed hit / miss
    x0 = y0 = None
    for yy in range(1024, 3072, 256):
        for xx in range(512, 3584, 160):
            po_, pd_ = acc.primary_rays(cam, xx, yy, w, h, 1)
            pr = acc.intersect_device(po_, pd_)[0]
            frac = float((pr != -1).float().mean().item())
            if 0.3 < frac < 0.8:
                x0, y0 = xx, yy
                break
        if x0 is not None:
            break
    assert x0 is not None, "no window with a silhouette found"
    rng = np.random.default_rng(11)
    uni = torch.from_numpy(rng.random(2 * N * w * h)).cuda()
    rgb, st = acc.render_ao_tile(cam, x0, y0, w, h, 1, NS, uniforms=uni)
    torch.cuda.synchronize()
    org = acc.scratch(0, np.float64, 3); dr = acc.scratch(1, np.float64, 3)
    got = (acc.scratch(2, np.uint32, 1), acc.scratch(3, np.float64, 1), acc.scratch(4, np.float64, 1), acc.scratch(5, np.float64, 1))
    # camera rays: the oracle's own camera for the same pixels
    ocam = po.Camera.from_ref(load_golden("ao_c1")["camera"], SIZE, SIZE)
    L = po.lib(); jit = (po.C.c_double * 2)()
    L.lo_subpixel_jitter.argtypes = [po.C.c_int] * 4 + [po.C.c_double * 2]
    L.lo_subpixel_jitter(0, 0, 1, 1, jit)
    eo = np.empty((w * h, 3)); ed = np.empty((w * h, 3)); k = 0
    for ly in range(h):
        for lx in range(w):
            L.lo_camera_ray(po.C.byref(ocam), float(x0 + lx + jit[0]), float(y0 + ly + jit[1]),
                            eo[k].ctypes.data_as(po._dp), ed[k].ctypes.data_as(po._dp))
            k += 1
    assert np.array_equal(org, eo) and np.array_equal(dr, ed)
    exp = o.intersect(org, dr, nthreads=32)
    assert_hits_equal(got, exp, "config-5 window, primary")
    nhit = int((exp[0] != po.MISS).sum())
    assert st["primary_hits"] == nhit and 0.2 * w * h < nhit < w * h
    # AO ray dump of the window: origins on the surface (+1e-6 Ns), cosine-stratified directions
    aorg = acc.scratch(8, np.float64, 3); adir = acc.scratch(9, np.float64, 3); occ = acc.scratch(10, np.uint8, 1)
    assert aorg.shape[0] == nhit * N == st["ao_rays"]
    rec = acc.scratch(7, np.float64, 12)
    assert np.array_equal(aorg.reshape(nhit, N, 3)[:, 0], rec[:, :3])          # every AO ray starts at the epilogue's origin
    eao = o.intersect(aorg, adir, nthreads=32)
    assert np.array_equal(occ.astype(bool), eao[0] != po.MISS)
    assert st["ao_occluded"] == int((eao[0] != po.MISS).sum())
    # the same AO rays through closest-hit: records bit for bit
    out = acc.intersect_device(torch.from_numpy(aorg).cuda(), torch.from_numpy(adir).cuda()); torch.cuda.synchronize()
    assert_hits_equal(tuple(x.cpu().numpy() for x in out), eao, "config-5 window, AO rays")


def test_full_frame_counts_tiling_and_any_hit(c5):
    import torch
    acc, o, cam = c5["acc"], c5["oracle"], c5["cam"]
    # the whole frame's camera rays: closest == oracle on hit/miss for ALL 16.8 M rays, records for a strided sample
    org, dr = acc.primary_rays(cam, 0, 0, SIZE, SIZE, 1)
    out = acc.intersect_device(org, dr)
    occ = acc.intersect_device(org, dr, mode=la.MODE_ANY)[0]
    torch.cuda.synchronize()
    hit = out[0] != -1
    assert torch.equal(hit, occ.bool())                                       # any-hit == a closest hit exists
    ho = org.cpu().numpy(); hd = dr.cpu().numpy()
    eprim = o.intersect(ho, hd, nthreads=64)
    assert np.array_equal(out[0].cpu().numpy().view(np.uint32), eprim[0])      # ids of all 16.8 M rays
    assert np.array_equal(out[1].cpu().numpy(), eprim[1])                      # and t
    nhit = int((eprim[0] != po.MISS).sum())
    # the frame: one tile vs 1024^2 tiles
    img1, st1 = render.render_ao_frame(acc, cam, 1, NS, tile=SIZE)
    img2, st2 = render.render_ao_frame(acc, cam, 1, NS, tile=1024)
    torch.cuda.synchronize()
    assert st1["primary_rays"] == SIZE * SIZE and st1["primary_hits"] == nhit and st1["ao_rays"] == nhit * 64
    assert st1 == st2 and torch.equal(img1, img2)
    frac = st1["ao_occluded"] / st1["ao_rays"]
    assert 0.02 < frac < 0.6
    im = img1.cpu().numpy()
    miss = ~hit.cpu().numpy().reshape(SIZE, SIZE)[::-1]                         # bucket_write's y flip
    assert (im[miss] == 0).all() and im[~miss].mean() > 0.5
    # same radiance as the untessellated scene's frame up to Monte-Carlo noise: tessellation moves no surface
    assert abs(float(im.mean()) - float(load_golden("ao_c1")["image"].mean())) < 5e-3
