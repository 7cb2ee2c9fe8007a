"""bench.py: BASELINE config 2 -- the example RIB as stated (1024^2, 3 x 3 pixel samples, 64 AO samples)."""
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403 -- the constants and helpers every leg shares
from .common import ROOT, gather_ceiling, pmc_source, host_cores


def config2_leg(la, acc_device, dev, steps, size=1024, gather=64):
    """BASELINE config 2 as stated: the reference's examples/ambient_occlusion.rib (tests/golden/rib/: 322 triangles, its own
    PixelSamples 3 3), 1024 x 1024, 64 AO samples, one GPU.  RIB reader -> accelerator -> one frame; timed: the frame with the
    image left in HBM (`frame_ms`) and through lh_render_ao_frame_host, the call lsh_hip makes (`frame_host_ms`: + the 12.6 MB
    image over PCIe).  tests/test_gpu_config2.py holds the parity side (camera-ray hits against the oracle, tiling, the driver)."""
    import torch
    from lucille_amd import render, rib
    t0 = time.perf_counter()
    sc = rib.RibScene(os.path.join(ROOT, "tests", "golden", "rib", "ambient_occlusion.rib"))
    parse_s = time.perf_counter() - t0
    acc = la.HipAccel(acc_device); sc.add_to(acc)
    t0 = time.perf_counter(); info = acc.commit(); commit_s = time.perf_counter() - t0
    ps = int(sc.info.pixel_samples[0])
    cam = la.Camera.make(size, size, sc.camera.flength, list(sc.camera.cam2world), sc.camera.rh)
    times = []; host_times = []; stats = []
    for it in range(steps + 1):
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        img, st = render.render_ao_frame(acc, cam, ps, gather, tile=size)
        torch.cuda.synchronize(dev)
        if it:
            times.append(time.perf_counter() - t0)
        stats.append(dict(st))
        t0 = time.perf_counter()
        himg, hst = acc.render_ao_frame_host(cam, ps, gather)
        if it:
            host_times.append(time.perf_counter() - t0)
    img2, st2 = render.render_ao_frame(acc, cam, ps, gather, tile=160)
    ok = all(s == stats[0] for s in stats) and st2 == stats[0] and bool(torch.equal(img, img2)) \
        and bool(np.array_equal(np.asarray(himg).reshape(size, size, 3), img.cpu().numpy()))
    rays = st["primary_rays"] + st["ao_rays"]
    acc.close(); sc.close()
    return {"workload": "BASELINE config 2: examples/ambient_occlusion.rib, %d triangles, %dx%d, PixelSamples %d %d, %d AO samples, one GPU"
                        % (info["ntriangles"], size, size, ps, ps, gather),
            "rib_parse_s": round(parse_s, 4), "commit_s": round(commit_s, 4), "rays_per_frame": int(rays),
            "primary_rays": int(st["primary_rays"]), "primary_hits": int(st["primary_hits"]), "ao_rays": int(st["ao_rays"]),
            "frame_ms": round(min(times) * 1e3, 3), "frame_host_ms": round(min(host_times) * 1e3, 3),
            "value": round(rays / min(times) / 1e6, 1), "unit": "Mrays/s",
            "validation": {"frames_repeat_and_retiled_bit_equal_and_host_call_equal": ok, "ok": ok}}
