"""The drop-in, end to end: the REFERENCE's own renderer (its Ri API, camera, AO transport,
MT19937, display path -- compiled from /root/reference into oracle/_ref/liblucille_ref_hip.so)
with its accelerator swapped for RI_ACCEL_HIP through the glue in integration/ri_accel_hip.c.
Every ray the reference issues goes through ri_raytrace -> accel->intersect ->
lh_accel_intersect1 -> the HIP kernel; the frame and every hit record must equal what the
reference's CPU BVH produces.  Needs no /root/reference at run time (the .so travels)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import load_golden

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.exists(os.path.join(po.HERE, "_ref", "liblucille_ref_hip.so")),
                                 reason="oracle/_ref/liblucille_ref_hip.so not built")]


def test_reference_renderer_on_hip_accel(tmp_path):
    from oracle import ref_rib
    g = load_golden("ao_c1")
    c2w = np.asarray(g["camera"][:16]).reshape(4, 4)
    # world_to_camera for RiConcatTransform: camera_to_world = inverse(w2c) * orientation(rh: z flip)
    orient = np.diag([1.0, 1.0, -1.0, 1.0])
    w2c = np.linalg.inv(c2w @ np.linalg.inv(orient))
    scene = {"ngeoms": int(g["ngeoms"]), "w2c": w2c, "fov": 45.0}
    for k in range(int(g["ngeoms"])):
        scene["pos%d" % k] = g["pos%d" % k]; scene["idx%d" % k] = g["idx%d" % k]
    sp = str(tmp_path / "scene.npz")
    np.savez(sp, **scene)
    kw = dict(width=40, height=40, gather_nsamples=4, pixel_samples=1, lib="liblucille_ref_hip.so")
    cpu = ref_rib.render_scene_subprocess(sp, str(tmp_path / "cpu.npz"), accel_method=1, **kw)
    hip = ref_rib.render_scene_subprocess(sp, str(tmp_path / "hip.npz"), accel_method=2, **kw)
    assert len(cpu["records"]) == len(hip["records"]) > 1600
    for f in ("org", "dir", "hit", "geom", "index", "t", "u", "v"):
        assert np.array_equal(cpu["records"][f], hip["records"][f]), f
    assert np.array_equal(cpu["image"], hip["image"])
    assert cpu["image"].max() > 0


def test_reference_renderer_with_render_threads_on_hip_accel(tmp_path):
    """the reference's default is one render thread per core (option.c:134): 4 of its pthreads call
    accel->intersect concurrently.  Per-thread MT19937 streams make the AO rays differ from the
    single-thread frame, so: same primary-ray hit records (as a set), frames agree statistically,
    CPU-BVH and HIP runs with 4 threads are both sane"""
    from oracle import ref_rib
    g = load_golden("ao_c1")
    c2w = np.asarray(g["camera"][:16]).reshape(4, 4)
    w2c = np.linalg.inv(c2w @ np.linalg.inv(np.diag([1.0, 1.0, -1.0, 1.0])))
    scene = {"ngeoms": int(g["ngeoms"]), "w2c": w2c, "fov": 45.0}
    for k in range(int(g["ngeoms"])):
        scene["pos%d" % k] = g["pos%d" % k]; scene["idx%d" % k] = g["idx%d" % k]
    sp = str(tmp_path / "scene.npz")
    np.savez(sp, **scene)
    kw = dict(width=64, height=64, gather_nsamples=16, pixel_samples=1, lib="liblucille_ref_hip.so", record=False)
    one = ref_rib.render_scene_subprocess(sp, str(tmp_path / "t1.npz"), accel_method=2, nthreads=1, **kw)
    four = ref_rib.render_scene_subprocess(sp, str(tmp_path / "t4.npz"), accel_method=2, nthreads=4, **kw)
    a, b = one["image"], four["image"]
    assert a.shape == b.shape and b.max() > 0
    assert np.array_equal(a.sum(axis=2) == 0, b.sum(axis=2) == 0)          # same pixels see geometry
    assert abs(float(a.mean()) - float(b.mean())) < 0.01                  # 16 AO samples per pixel, 4 096 pixels
