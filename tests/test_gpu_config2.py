"""BASELINE config 2 as it is stated: examples/ambient_occlusion.rib (tests/golden/rib/, the reference's own file),
1024 x 1024, 64 AO samples, the RIB's own PixelSamples 3 3 -- through the `lsh_hip` driver and through the tile pipeline.

  * which camera samples hit geometry: the oracle's walk of the device-generated camera rays (those rays are the reference's
    bit for bit: test_gpu_ao.py::test_primary_rays_bit_exact) against the frame's own counters, hit record by hit record;
  * the frame does not depend on how it is cut into tiles (the sample stream is keyed by absolute sample position);
  * the driver writes that frame.
"""
import os
import re
import subprocess

import numpy as np
import pytest

import lucille_amd as la
from lucille_amd import render, rib
from oracle import pyoracle as po
from tests.helpers import assert_hits_equal

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
RIB_FILE = os.path.join(HERE, "golden", "rib", "ambient_occlusion.rib")
SIZE, GATHER = 1024, 64


@pytest.fixture(scope="module")
def scene():
    sc = rib.RibScene(RIB_FILE)
    assert tuple(sc.info.pixel_samples) == (3, 3) and sc.info.nmeshes == 4
    acc = la.HipAccel(0); sc.add_to(acc); info = acc.commit()
    assert info["ntriangles"] == 322
    cam = la.Camera.make(SIZE, SIZE, sc.camera.flength, list(sc.camera.cam2world), sc.camera.rh)
    o = po.Oracle()
    for m in sc.meshes():
        o.add_mesh(m["positions"][:, :3], m["indices"])
    o.build()
    yield {"sc": sc, "acc": acc, "cam": cam, "oracle": o}
    acc.close()


def test_config2_primary_hits_equal_the_oracle(scene):
    import torch
    acc, cam, o = scene["acc"], scene["cam"], scene["oracle"]
    img, st = render.render_ao_frame(acc, cam, 3, GATHER, tile=SIZE, seed=1)
    torch.cuda.synchronize()
    assert st["primary_rays"] == SIZE * SIZE * 9 and st["ao_rays"] == st["primary_hits"] * GATHER
    hits = 0
    for y0 in range(0, SIZE, 128):                                    # 1.2 M camera rays per band: seconds on the host cores
        org, dr = acc.primary_rays(cam, 0, y0, SIZE, 128, 3)
        out = acc.intersect_device(org, dr); torch.cuda.synchronize()
        exp = o.intersect(org.cpu().numpy(), dr.cpu().numpy(), nthreads=min(16, os.cpu_count() or 1))
        assert_hits_equal(tuple(x.cpu().numpy() for x in out[:4]), exp, "config 2 camera rays, band %d" % y0)
        hits += int((exp[0] != po.MISS).sum())
    assert hits == st["primary_hits"], (hits, st)
    assert 0.2 * st["primary_rays"] < hits < 0.9 * st["primary_rays"]             # the frame shows the scene and its background
    # a pixel whose nine samples all miss is black, one with a hit is not (unoccluded directions exist above every surface)
    lum = img.sum(dim=2)
    assert 0.1 < float((lum == 0).float().mean()) < 0.8 and 0.0 < float(img.max()) <= 1.0


def test_config2_frame_does_not_depend_on_the_tiling(scene):
    import torch
    acc, cam = scene["acc"], scene["cam"]
    whole, s0 = render.render_ao_frame(acc, cam, 3, GATHER, tile=SIZE, seed=1)
    tiled, s1 = render.render_ao_frame(acc, cam, 3, GATHER, tile=160, seed=1)       # 7 x 7 tiles, ragged right / bottom edge
    torch.cuda.synchronize()
    assert s0 == s1 and torch.equal(whole, tiled)
    acc.set_param("ao_fused", 0)                                                     # materialised AO rays: the same frame
    try:
        unfused, s2 = render.render_ao_frame(acc, cam, 3, GATHER, tile=256, seed=1)
        torch.cuda.synchronize()
    finally:
        acc.set_param("ao_fused", 1)
    assert s2 == s0 and torch.equal(whole, unfused)


def test_config2_through_lsh_hip(scene, tmp_path):
    """the driver, left to the RIB's own PixelSamples and gather defaults except for the sizes config 2 names"""
    import torch
    acc, cam = scene["acc"], scene["cam"]
    r = subprocess.run([rib.lsh_hip_path(), "--resolution", "%dx%d" % (SIZE, SIZE), "--gather", str(GATHER), "--seed", "1", "--verbose", RIB_FILE],
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "322 triangles" in r.stdout
    m = re.search(r"\((\d+) primary \+ (\d+) AO rays", r.stdout)
    img, st = render.render_ao_frame(acc, cam, 3, GATHER, tile=SIZE, seed=1)
    torch.cuda.synchronize()
    assert m and int(m.group(1)) == st["primary_rays"] == SIZE * SIZE * 9 and int(m.group(2)) == st["ao_rays"]     # 9: the RIB's PixelSamples 3 3
    p = str(tmp_path / "ambient_occlusion.hdr")
    ref = str(tmp_path / "ref.hdr")
    rib.hdr_write(ref, img.cpu().numpy())
    assert open(p, "rb").read() == open(ref, "rb").read()                           # the driver's file == the pipeline's frame
