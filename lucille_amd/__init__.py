"""lucille_amd -- MI355X (gfx950) accelerator for lucille's ray-query hot path.

The product is the C-ABI shared library ``lucille_amd/csrc/liblucille_hip.so``
(declared in ``include/lucille_hip.h``); this package is the thin Python
plumbing used by the tests, ``bench.py`` and the multi-GPU driver:

  lucille_amd.binding   ctypes view of the C ABI (+ torch device-pointer helpers)
  lucille_amd.render    frame loops over the tile entry points (AO, path-traced), sharded or not
  lucille_amd.shard     image-space / ray-slice sharding over torch.distributed
  lucille_amd.rib       RIB-subset reader / .hdr writer entry points, lsh_hip path
  lucille_amd.scenes    test / bench scene helpers (tessellation, fixtures)

The host mirror of lucille's own plugin API (ri_geom_*, ri_accel_*, ri_raytrace ...,
include/lucille_accel.h, lh_host.c) is C and is exercised by the C program in tests/c/.

There is no CPU fallback: loading fails loudly when the library is missing and
every query fails loudly when no HIP device is visible.
"""
from .binding import (Camera, HipAccel, LucilleHipError, MISS, MODE_ANY, MODE_CLOSEST,  # noqa: F401
                      VARIANT_DEFAULT, VARIANT_DIRECT,
                      VARIANT_SPEC, build_library, device_count, library_path,
                      HipMulti, HipDist, DIST_RCCL, DIST_SHM, Material, Environment, ALL_MESHES, PT_REFERENCE_WEIGHTS, ATTR_COLOR, ATTR_TANGENT, ATTR_BINORMAL,
                      ATTR_TEXCOORD, ATTR_TEXCOORD_UNSHARED, STATE_DOUBLES)
