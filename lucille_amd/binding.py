"""ctypes binding of include/lucille_hip.h (the drop-in C ABI).

Reference interface mirrored: accel_build_func / accel_free_func /
accel_intersect_func (lucille src/render/accel.h:24-34) -- see the header for
the per-function mapping.  No compute happens in Python.
"""
import atexit
import ctypes as C
import os
import subprocess
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")

MISS = 0xFFFFFFFF
T_INF = 1.0e38
MODE_CLOSEST, MODE_ANY = 0, 1
VARIANT_DEFAULT, VARIANT_DIRECT, VARIANT_SPEC = -1, 0, 4      # the tuned walk (= SPEC), the textbook reference walk


class LucilleHipError(RuntimeError):
    pass


def library_path():
    """LH_LIBRARY overrides the path (A/B runs of differently built kernels in tools/)"""
    return os.environ.get("LH_LIBRARY") or os.path.join(CSRC, "liblucille_hip.so")


def build_library(force=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC]
    if force:
        subprocess.check_call(args + ["clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args + ["liblucille_hip.so"], stdout=subprocess.DEVNULL)
    return library_path()


class AccelInfo(C.Structure):
    _fields_ = [("ntriangles", C.c_uint32), ("nnodes", C.c_uint32), ("nleaves", C.c_uint32),
                ("max_depth", C.c_uint32), ("device_bytes", C.c_uint64), ("build_seconds", C.c_double),
                ("upload_seconds", C.c_double), ("device", C.c_int), ("ref_build_seconds", C.c_double),
                ("nnodes_traversal", C.c_uint32), ("ntriangles_in_tree", C.c_uint32)]


class RasterPlane(C.Structure):
    """lh_raster_plane_t (include/lucille_hip.h)"""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("frame", C.c_double * 9), ("eye", C.c_double * 3), ("fov", C.c_double)]


class Camera(C.Structure):
    """lh_camera_t: the members of ri_camera_t the ray generator reads (camera.c:248-318)"""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("rh", C.c_int), ("ortho", C.c_int),
                ("flength", C.c_double), ("cam2world", C.c_double * 16)]

    @classmethod
    def make(cls, width, height, flength, cam2world, rh=1):
        c = cls(); c.width = int(width); c.height = int(height); c.rh = int(rh); c.flength = float(flength)
        m = np.asarray(cam2world, np.float64).reshape(16)
        for i in range(16):
            c.cam2world[i] = float(m[i])
        return c


class Material(C.Structure):
    """lh_material_t = ri_material_t's kd, ks, kt, ior (src/render/material.h:21-30)"""
    _fields_ = [("kd", C.c_float * 3), ("ks", C.c_float * 3), ("kt", C.c_float * 3), ("ior", C.c_float)]

    @classmethod
    def make(cls, kd=(1, 1, 1), ks=(0, 0, 0), kt=(0, 0, 0), ior=1.0):
        m = cls()
        for k in range(3):
            m.kd[k] = float(kd[k]); m.ks[k] = float(ks[k]); m.kt[k] = float(kt[k])
        m.ior = float(ior)
        return m


class Environment(C.Structure):
    _fields_ = [("rgb", C.c_float * 3), ("map_rgba", C.c_void_p), ("width", C.c_int), ("height", C.c_int)]


ALL_MESHES = 0xFFFFFFFF
PT_REFERENCE_WEIGHTS = 1
ATTR_COLOR, ATTR_TANGENT, ATTR_BINORMAL, ATTR_TEXCOORD, ATTR_TEXCOORD_UNSHARED = 0, 1, 2, 3, 4
STATE_DOUBLES = 24


class PtStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("max_depth_reached", C.c_uint64)]


class TileStats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("primary_hits", C.c_uint64), ("ao_rays", C.c_uint64),
                ("ao_occluded", C.c_uint64)]


# lh_beam_set_t (include/lucille_hip.h): a beam as lucille's ri_beam_set leaves it -- 232 bytes
BEAM_SET_DTYPE = np.dtype([("org", np.float64, (3,)), ("dir", np.float64, (4, 3)), ("normal", np.float64, (4, 3)),
                           ("dominant_axis", np.int32), ("dirsign", np.int32, (3,))])

# every symbol include/lucille_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "lh_device_count", "lh_last_error", "lh_accel_create", "lh_accel_add_mesh", "lh_accel_commit", "lh_accel_wait_exact", "lh_accel_ref_tree",
    "lh_accel_destroy", "lh_accel_info", "lh_accel_prim_lookup", "lh_accel_intersect1", "lh_accel_combine_statistics", "lh_accel_intersect_diag_host", "lh_accel_intersect_diag_device",
    "lh_accel_intersect_host", "lh_accel_intersect_device", "lh_accel_intersect_device_counted", "lh_accel_last_retraced", "lh_accel_dump_node_bytes",
    "lh_accel_set_grid", "lh_accel_set_param", "lh_accel_export", "lh_accel_set_normals", "lh_render_primary_rays",
    "lh_render_ao_tile", "lh_render_ao_tile_host", "lh_render_ao_bands", "lh_render_scratch", "lh_accel_beam_visibility_host", "lh_accel_beam_visibility_device", "lh_accel_beam_visibility_set_host", "lh_accel_beam_raster_host", "lh_accel_beam_raster_device", "lh_accel_beam_raster_set_host", "lh_render_pt_tile",
    "lh_accel_trace_statistics", "lh_accel_statistics", "lh_accel_slot_statistics",
    "lh_render_ao_frame_host", "lh_rib_load", "lh_rib_free", "lh_rib_last_error", "lh_rib_info", "lh_rib_messages",
    "lh_rib_mesh", "lh_accel_add_rib_scene", "lh_hdr_write",
    "lh_accel_set_material", "lh_accel_set_environment", "lh_render_pt_tile2", "lh_render_pt_bands", "lh_accel_set_attribute",
    "lh_accel_state_build_device", "lh_accel_state_build_host",
    "lh_multi_create", "lh_multi_destroy", "lh_multi_ndevices", "lh_multi_accel", "lh_multi_add_mesh", "lh_multi_set_normals",
    "lh_multi_add_rib_scene", "lh_multi_commit", "lh_multi_set_material", "lh_multi_set_environment", "lh_multi_intersect_host",
    "lh_multi_render_ao_frame_host", "lh_multi_render_pt_frame_host",
    "lh_synth_soup_triangles", "lh_synth_soup_rays", "lh_synth_tessellate", "lh_synth_skip",
    "lh_dist_unique_id", "lh_dist_init", "lh_dist_init_file", "lh_dist_destroy", "lh_dist_rank", "lh_dist_world", "lh_dist_transport",
    "lh_dist_barrier", "lh_dist_host_barrier", "lh_dist_broadcast", "lh_dist_gather", "lh_dist_pack_records16", "lh_dist_broadcast_scene", "lh_dist_render_ao_frame_host",
]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    try:                      # torch ships its own HIP runtime: let it initialise first when both live in one process
        import torch          # noqa: F401
    except Exception:         # noqa: BLE001 -- the library itself does not need torch
        pass
    if not os.path.exists(path):
        raise LucilleHipError(
            "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C %s` "
            "(there is no CPU fallback)" % (path, CSRC))
    L = C.CDLL(path)
    vp, sz, i32, u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32
    L.lh_device_count.restype = i32
    L.lh_last_error.restype = C.c_char_p
    L.lh_accel_create.argtypes = [C.POINTER(vp), i32]
    L.lh_accel_add_mesh.argtypes = [vp, u32, vp, sz, u32, vp]
    L.lh_accel_commit.argtypes = [vp, i32]
    L.lh_accel_wait_exact.argtypes = [vp]
    L.lh_accel_ref_tree.argtypes = [vp, C.POINTER(C.c_uint32), vp, vp]
    L.lh_accel_destroy.argtypes = [vp]
    L.lh_accel_destroy.restype = None
    L.lh_accel_info.argtypes = [vp, C.POINTER(AccelInfo)]
    L.lh_accel_prim_lookup.argtypes = [vp, u32, C.POINTER(u32), C.POINTER(u32)]
    L.lh_accel_intersect1.argtypes = [vp, vp, vp, C.POINTER(u32), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.lh_accel_intersect_host.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp, i32]
    L.lh_accel_intersect_device.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    L.lh_accel_last_retraced.argtypes = [vp]; L.lh_accel_last_retraced.restype = C.c_uint64
    L.lh_accel_dump_node_bytes.argtypes = [vp]
    L.lh_accel_intersect_device_counted.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp, i32, i32,
                                                    C.POINTER(C.c_uint64)]
    L.lh_accel_set_grid.argtypes = [vp, i32]
    L.lh_accel_set_param.argtypes = [vp, C.c_char_p, i32]
    L.lh_accel_trace_statistics.argtypes = [vp, i32]
    L.lh_accel_statistics.argtypes = [vp, C.POINTER(C.c_uint64), i32]
    L.lh_accel_slot_statistics.argtypes = [vp, C.POINTER(C.c_uint64), i32]
    L.lh_accel_export.argtypes = [vp, vp, vp]
    L.lh_accel_set_normals.argtypes = [vp, u32, vp, sz, i32]
    L.lh_render_primary_rays.argtypes = [vp, C.POINTER(Camera), i32, i32, i32, i32, i32, vp, vp, vp]
    L.lh_render_ao_tile.argtypes = [vp, C.POINTER(Camera), i32, i32, i32, i32, i32, i32, C.c_uint64, vp, vp,
                                    C.POINTER(TileStats), vp]
    L.lh_render_ao_bands.argtypes = [vp, C.POINTER(Camera), i32, C.POINTER(i32), i32, i32, i32, C.c_uint64, vp, C.POINTER(TileStats), vp]
    L.lh_render_scratch.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(sz)]
    L.lh_render_pt_tile.argtypes = [vp, C.POINTER(Camera), i32, i32, i32, i32, i32, i32, i32, i32, C.c_float,
                                    C.POINTER(C.c_float * 3), C.c_uint64, vp, C.POINTER(PtStats), vp]
    L.lh_accel_beam_visibility_host.argtypes = [vp, sz, vp, vp, vp]
    L.lh_accel_beam_visibility_device.argtypes = [vp, sz, vp, vp, vp, vp]
    L.lh_accel_beam_raster_host.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp]
    L.lh_accel_beam_raster_device.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp, vp]
    L.lh_accel_set_material.argtypes = [vp, u32, C.POINTER(Material)]
    L.lh_accel_set_environment.argtypes = [vp, C.POINTER(Environment)]
    L.lh_render_pt_bands.argtypes = [vp, C.POINTER(Camera), i32, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(Material), C.c_uint64, vp,
                                     C.POINTER(PtStats), vp]
    L.lh_render_pt_tile2.argtypes = [vp, C.POINTER(Camera), i32, i32, i32, i32, i32, i32, i32, i32, i32, C.c_uint64, vp,
                                     C.POINTER(PtStats), vp]
    L.lh_accel_set_attribute.argtypes = [vp, u32, i32, vp, sz, u32]
    L.lh_accel_state_build_device.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp, vp]
    L.lh_accel_state_build_host.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp]
    L.lh_multi_create.argtypes = [C.POINTER(vp), i32, C.POINTER(i32)]
    L.lh_multi_destroy.argtypes = [vp]; L.lh_multi_destroy.restype = None
    L.lh_multi_ndevices.argtypes = [vp]
    L.lh_multi_accel.argtypes = [vp, i32]; L.lh_multi_accel.restype = vp
    L.lh_multi_add_mesh.argtypes = [vp, u32, vp, sz, u32, vp]
    L.lh_multi_set_normals.argtypes = [vp, u32, vp, sz, i32]
    L.lh_multi_add_rib_scene.argtypes = [vp, vp]
    L.lh_multi_commit.argtypes = [vp, i32]
    L.lh_multi_set_material.argtypes = [vp, u32, C.POINTER(Material)]
    L.lh_multi_set_environment.argtypes = [vp, C.POINTER(Environment)]
    L.lh_multi_intersect_host.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp, i32]
    L.lh_multi_render_ao_frame_host.argtypes = [vp, C.POINTER(Camera), i32, i32, C.c_uint64, i32, vp, C.POINTER(TileStats), vp]
    L.lh_multi_render_pt_frame_host.argtypes = [vp, C.POINTER(Camera), i32, i32, i32, i32, C.c_uint64, i32, vp, C.POINTER(PtStats), vp]
    L.lh_synth_soup_triangles.argtypes = [C.POINTER(C.c_uint64), u32, C.c_double, vp, vp]
    L.lh_synth_soup_triangles.restype = None
    L.lh_synth_soup_rays.argtypes = [C.POINTER(C.c_uint64), sz, vp, vp]
    L.lh_synth_soup_rays.restype = None
    L.lh_synth_skip.argtypes = [C.POINTER(C.c_uint64), C.c_uint64]
    L.lh_synth_skip.restype = None
    L.lh_synth_tessellate.argtypes = [vp, sz, i32, vp]
    L.lh_synth_tessellate.restype = None
    _lib = L
    return L


def device_count():
    return int(lib().lh_device_count())


def _check(rc, what):
    if rc < 0:
        raise LucilleHipError("%s: %s" % (what, lib().lh_last_error().decode()))
    return rc


def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _dptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else C.c_void_p(t.data_ptr())


_live = weakref.WeakSet()


@atexit.register
def _close_all():
    # device memory must be released while the HIP runtime is still alive
    for a in list(_live):
        a.close()


class HipAccel:
    """One committed accelerator == the `void *accel` the reference's vtable carries."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        _check(self.L.lh_accel_create(C.byref(self.h), int(device)), "lh_accel_create")
        self.device = int(device)
        self.committed = False
        self._npos = []            # vertex count per added mesh (set_normals validates against it)
        self._env = None           # what set_environment was last given (render.render_pt_frame_sharded restores it)
        _live.add(self)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.lh_accel_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- build ------------------------------------------------------------
    def add_mesh(self, positions, indices):
        """positions: [npos,3] (packed xyz) or [npos,4] (lucille's ri_vector_t) float64"""
        P = _np(positions, np.float64)
        if P.ndim != 2 or P.shape[1] not in (3, 4):
            raise ValueError("positions must be [n,3] or [n,4]")
        I = _np(indices, np.uint32).reshape(-1)
        _check(self.L.lh_accel_add_mesh(self.h, P.shape[0], P.ctypes.data, P.shape[1] * 8, I.shape[0],
                                        I.ctypes.data), "lh_accel_add_mesh")
        self._npos.append(P.shape[0])

    def set_normals(self, mesh, normals, two_side=0):
        N = _np(normals, np.float64) if normals is not None else None
        if int(mesh) < 0 or int(mesh) >= len(self._npos):
            raise ValueError("set_normals: mesh %d was not added" % int(mesh))
        if N is not None and (N.ndim != 2 or N.shape[1] not in (3, 4) or N.shape[0] != self._npos[int(mesh)]):
            # the C ABI reads one normal per vertex of the mesh: a short array would be read past its end
            raise ValueError("normals must be [%d,3] or [%d,4] (one per vertex of mesh %d)"
                             % (self._npos[int(mesh)], self._npos[int(mesh)], int(mesh)))
        _check(self.L.lh_accel_set_normals(self.h, int(mesh), N.ctypes.data if N is not None else None,
                                           (N.shape[1] * 8) if N is not None else 24, int(two_side)),
               "lh_accel_set_normals")

    def commit(self, build_threads=0, on_device=False, build=None):
        """build: "device" (= on_device: both trees on the GPU, LH_BUILD_ON_DEVICE), "host" (LH_BUILD_ON_HOST; a thread count
        in build_threads also means the host), None / "auto": the library chooses by the size of the scene"""
        if build not in (None, "auto", "host", "device"):
            raise ValueError("build must be 'auto', 'host' or 'device'")
        code = -2 if (on_device or build == "device") else (int(build_threads) if int(build_threads) > 0 else (-3 if build == "host" else 0))
        _check(self.L.lh_accel_commit(self.h, code), "lh_accel_commit")
        self.committed = True
        return self.info()

    def wait_exact(self):
        _check(self.L.lh_accel_wait_exact(self.h), "lh_accel_wait_exact")

    REF_NODE = np.dtype([("box", "<f8", (2, 6)), ("child", "<i4", (2,)), ("axis", "<i4"), ("is_leaf", "<i4"), ("first", "<u4"),
                         ("count", "<u4"), ("parent", "<i4"), ("depth", "<i4")])

    def ref_tree(self):
        """lucille's own tree as the kernels read it: (nodes [structured, 128 bytes each, root 0], leaf_prims)"""
        nn = C.c_uint32()
        _check(self.L.lh_accel_ref_tree(self.h, C.byref(nn), None, None), "lh_accel_ref_tree")
        nodes = np.zeros(nn.value, dtype=self.REF_NODE)
        prims = np.zeros(self.info()["ntriangles"], dtype=np.uint32)
        _check(self.L.lh_accel_ref_tree(self.h, C.byref(nn), nodes.ctypes.data_as(C.c_void_p), prims.ctypes.data_as(C.c_void_p)), "lh_accel_ref_tree")
        return nodes, prims

    def info(self):
        s = AccelInfo()
        _check(self.L.lh_accel_info(self.h, C.byref(s)), "lh_accel_info")
        return {k: getattr(s, k) for k, _ in s._fields_}

    def prim_lookup(self, prim):
        m, i = C.c_uint32(), C.c_uint32()
        _check(self.L.lh_accel_prim_lookup(self.h, int(prim), C.byref(m), C.byref(i)), "lh_accel_prim_lookup")
        return int(m.value), int(i.value)

    def set_param(self, name, value):
        _check(self.L.lh_accel_set_param(self.h, name.encode(), int(value)), "lh_accel_set_param")

    def set_grid(self, blocks):
        _check(self.L.lh_accel_set_grid(self.h, int(blocks)), "lh_accel_set_grid")

    def export(self):
        inf = self.info()
        nodes = np.zeros((inf["nnodes"], 16), np.float32)
        tri32 = np.zeros((inf["ntriangles"], 12), np.float32)
        _check(self.L.lh_accel_export(self.h, nodes.ctypes.data, tri32.ctypes.data), "lh_accel_export")
        return nodes, tri32

    # ---- queries ----------------------------------------------------------
    def intersect_diag(self, org, dr):
        """per-ray traversal diagnostics -> ((prim, t, u, v), diag uint32 [n, 4]: 4-wide node visits, leaf visits, triangle
        records through the fp32 filter, fp64 tests) -- the sequential walk's numbers (ri_bvh_diag_t, bvh.h:103-110)"""
        o = _np(org, np.float64).reshape(-1, 3); d = _np(dr, np.float64).reshape(-1, 3); n = o.shape[0]
        prim = np.empty(n, np.uint32); t = np.empty(n); u = np.empty(n); v = np.empty(n); diag = np.zeros((n, 4), np.uint32)
        self.L.lh_accel_intersect_diag_host.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 7
        _check(self.L.lh_accel_intersect_diag_host(self.h, n, o.ctypes.data, d.ctypes.data, prim.ctypes.data, t.ctypes.data, u.ctypes.data,
                                                   v.ctypes.data, diag.ctypes.data), "lh_accel_intersect_diag_host")
        return (prim, t, u, v), diag

    def intersect_diag_device(self, org, dr, mode=MODE_CLOSEST):
        """CUDA float64 [n,3] rays -> uint32 [n,4] CUDA tensor of the sequential walk's per-ray counts (closest- or any-hit)"""
        import torch
        n = org.shape[0]
        diag = torch.zeros((n, 4), dtype=torch.int32, device=org.device)
        self.L.lh_accel_intersect_diag_device.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _check(self.L.lh_accel_intersect_diag_device(self.h, n, _dptr(org), _dptr(dr), int(mode), _dptr(diag),
                                                     C.c_void_p(torch.cuda.current_stream(org.device).cuda_stream)), "lh_accel_intersect_diag_device")
        return diag

    def combine_statistics(self, clear=False):
        """(launches, rays) of the coalesced single-ray path since the last clear"""
        o = (C.c_uint64 * 2)()
        self.L.lh_accel_combine_statistics.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
        _check(self.L.lh_accel_combine_statistics(self.h, o, 1 if clear else 0), "lh_accel_combine_statistics")
        return int(o[0]), int(o[1])

    def intersect1(self, org, dr):
        o = _np(org, np.float64).reshape(3); d = _np(dr, np.float64).reshape(3)
        p = C.c_uint32(); t = C.c_double(); u = C.c_double(); v = C.c_double()
        hit = _check(self.L.lh_accel_intersect1(self.h, o.ctypes.data, d.ctypes.data, C.byref(p), C.byref(t),
                                                C.byref(u), C.byref(v)), "lh_accel_intersect1")
        return hit, int(p.value), t.value, u.value, v.value

    def intersect_host(self, org, dr, mode=MODE_CLOSEST):
        o = _np(org, np.float64).reshape(-1, 3); d = _np(dr, np.float64).reshape(-1, 3)
        n = o.shape[0]
        if mode == MODE_CLOSEST:
            prim = np.empty(n, np.uint32); t = np.empty(n); u = np.empty(n); v = np.empty(n)
            _check(self.L.lh_accel_intersect_host(self.h, n, o.ctypes.data, d.ctypes.data, prim.ctypes.data,
                                                  t.ctypes.data, u.ctypes.data, v.ctypes.data, None, mode),
                   "lh_accel_intersect_host")
            return prim, t, u, v
        occ = np.empty(n, np.uint8)
        _check(self.L.lh_accel_intersect_host(self.h, n, o.ctypes.data, d.ctypes.data, None, None, None, None,
                                              occ.ctypes.data, mode), "lh_accel_intersect_host")
        return occ

    def intersect_device(self, org, dr, out=None, mode=MODE_CLOSEST, variant=VARIANT_DEFAULT, stream=None,
                         counters=False):
        """org, dr: CUDA(HIP) float64 tensors [n,3], contiguous.  Enqueues on `stream`
        (default: torch's current stream) and returns the output tensors."""
        import torch
        assert org.is_cuda and dr.is_cuda and org.dtype == torch.float64 and dr.dtype == torch.float64
        assert org.is_contiguous() and dr.is_contiguous()
        n = org.shape[0]
        dev = org.device
        if out is None:
            if mode == MODE_CLOSEST:
                out = (torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.float64, device=dev),
                       torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev))
            else:
                out = (torch.empty(n, dtype=torch.uint8, device=dev),)
        if mode == MODE_CLOSEST:
            prim, t, u, v = out; occ = None
        else:
            prim = t = u = v = None; occ = out[0]
        if counters:
            torch.cuda.synchronize(dev)
            c = (C.c_uint64 * 4)()
            _check(self.L.lh_accel_intersect_device_counted(self.h, n, _dptr(org), _dptr(dr), _dptr(prim), _dptr(t),
                                                            _dptr(u), _dptr(v), _dptr(occ), mode, variant, c),
                   "lh_accel_intersect_device_counted")
            return out, {"nodes": int(c[0]), "tris": int(c[1]), "exact": int(c[2]), "rays": int(c[3]),
                         "retraced": int(self.L.lh_accel_last_retraced(self.h))}
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        _check(self.L.lh_accel_intersect_device(self.h, n, _dptr(org), _dptr(dr), _dptr(prim), _dptr(t), _dptr(u),
                                                _dptr(v), _dptr(occ), mode, variant, C.c_void_p(stream)),
               "lh_accel_intersect_device")
        return out

    def trace_statistics(self, enable=True):
        """ri_bvh_clear_stat_traversal / RI_BVH_TRACE_STATISTICS: count node visits, triangle tests, fp64 re-tests, rays and hits of
        the host ray dumps and of the AO tile pipeline (the counting instantiations of the same kernels)"""
        _check(self.L.lh_accel_trace_statistics(self.h, 1 if enable else 0), "lh_accel_trace_statistics")

    def statistics(self, clear=False):
        c = (C.c_uint64 * 5)()
        _check(self.L.lh_accel_statistics(self.h, c, 1 if clear else 0), "lh_accel_statistics")
        return dict(zip(("nodes", "tris", "exact", "rays", "hits"), (int(x) for x in c)))

    def slot_statistics(self, clear=False):
        c = (C.c_uint64 * 3)()
        _check(self.L.lh_accel_slot_statistics(self.h, c, 1 if clear else 0), "lh_accel_slot_statistics")
        return dict(zip(("node_slots", "tri_slots", "regroups"), (int(x) for x in c)))

    def dump_node_bytes(self):
        """64: ray dumps walk the 4-wide nodes; 128: the 8-wide nodes (scene larger than the Infinity Cache, or wide8 = 1)"""
        return int(self.L.lh_accel_dump_node_bytes(self.h))

    def beam_raster(self, org, corner_dirs, corners, width, height, frame, eye, fov, t_init=None):
        """ri_beam_set + ri_raster_plane_setup + ri_bvh_intersect_beam for n beams over one width x height raster window
        (frame = du dv dw, eye = plane->org, fov in degrees; corners [n,3] = plane->corner per beam).
        -> (t [n,height,width] float64, status int32 [n] (0 traced, 1 nothing done, -1 invalid beam), flags uint64 [n,4]);
        planes that are not traced keep t_init (default zeros = a freshly set-up plane)"""
        o = np.ascontiguousarray(org, np.float64).reshape(-1, 3); n = o.shape[0]
        d = np.ascontiguousarray(corner_dirs, np.float64).reshape(n, 4, 3)
        c = np.ascontiguousarray(corners, np.float64).reshape(n, 3)
        pl = RasterPlane(int(width), int(height), (C.c_double * 9)(*np.asarray(frame, np.float64).reshape(-1)),
                         (C.c_double * 3)(*np.asarray(eye, np.float64).reshape(-1)), float(fov))
        t = np.zeros((n, height, width)) if t_init is None else np.ascontiguousarray(t_init, np.float64).reshape(n, height, width).copy()
        st = np.empty(n, np.int32); fl = np.zeros((n, 4), np.uint64)
        _check(self.L.lh_accel_beam_raster_host(self.h, n, o.ctypes.data, d.ctypes.data, c.ctypes.data, C.addressof(pl),
                                                t.ctypes.data, st.ctypes.data, fl.ctypes.data), "lh_accel_beam_raster_host")
        return t, st, fl

    def beam_visibility(self, org, corner_dirs):
        """ri_beam_set + ri_bvh_intersect_beam_visibility for n beams: org [n,3], corner_dirs [n,4,3]
        -> int32 [n] in {0 miss, 1 hit completely, 2 hit partially, -1 invalid beam}"""
        o = _np(org, np.float64).reshape(-1, 3); d = _np(corner_dirs, np.float64).reshape(-1, 4, 3)
        res = np.empty(o.shape[0], np.int32)
        _check(self.L.lh_accel_beam_visibility_host(self.h, o.shape[0], o.ctypes.data, d.ctypes.data, res.ctypes.data),
               "lh_accel_beam_visibility_host")
        return res

    def beam_visibility_set(self, beams):
        """ri_bvh_intersect_beam_visibility for n beams a caller's ri_beam_set has ALREADY set up: beams = a structured array of
        BEAM_SET_DTYPE (lh_beam_set_t: org, dir[4], normal[4], dominant_axis, dirsign[3]) -> int32 [n]"""
        b = np.ascontiguousarray(beams, BEAM_SET_DTYPE).reshape(-1)
        res = np.empty(b.shape[0], np.int32)
        self.L.lh_accel_beam_visibility_set_host.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        _check(self.L.lh_accel_beam_visibility_set_host(self.h, b.shape[0], b.ctypes.data, res.ctypes.data), "lh_accel_beam_visibility_set_host")
        return res

    # ---- tile rendering (device-resident pipeline) ---------------------------
    def primary_rays(self, cam, x0, y0, w, h, pixel_samples=1, stream=None):
        import torch
        n = w * h * pixel_samples * pixel_samples
        dev = torch.device("cuda", self.device)
        org = torch.empty((n, 3), dtype=torch.float64, device=dev); dr = torch.empty((n, 3), dtype=torch.float64, device=dev)
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        _check(self.L.lh_render_primary_rays(self.h, C.byref(cam), x0, y0, w, h, pixel_samples, _dptr(org), _dptr(dr),
                                             C.c_void_p(stream)), "lh_render_primary_rays")
        return org, dr

    def render_ao_tile(self, cam, x0, y0, w, h, pixel_samples, gather_nsamples, seed=1, uniforms=None, out=None,
                       stream=None):
        """-> (rgb float32 [h,w,3] CUDA tensor in image orientation, stats dict)"""
        import torch
        dev = torch.device("cuda", self.device)
        if out is None:
            out = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        st = TileStats()
        _check(self.L.lh_render_ao_tile(self.h, C.byref(cam), x0, y0, w, h, pixel_samples, gather_nsamples, int(seed),
                                        _dptr(uniforms), _dptr(out), C.byref(st), C.c_void_p(stream)),
               "lh_render_ao_tile")
        return out, {k: int(getattr(st, k)) for k, _ in st._fields_}

    def render_ao_frame_host(self, cam, pixel_samples, gather_nsamples, seed=1, tile=0):
        """lh_render_ao_frame_host: the whole frame into host memory -> (float32 [H, W, 3] numpy, top row first; stats)"""
        rgb = np.empty((cam.height, cam.width, 3), np.float32)
        st = TileStats()
        self.L.lh_render_ao_frame_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
        _check(self.L.lh_render_ao_frame_host(self.h, C.byref(cam), int(pixel_samples), int(gather_nsamples), int(seed), int(tile),
                                              rgb.ctypes.data, C.byref(st)), "lh_render_ao_frame_host")
        return rgb, {k: int(getattr(st, k)) for k, _ in st._fields_}

    def render_ao_bands(self, cam, band_y0, band_rows, pixel_samples, gather_nsamples, seed=1, out=None, stream=None):
        """full-width bands (first lines band_y0, band_rows lines each) as ONE device batch ->
        (float32 [nbands, band_rows, W, 3] CUDA tensor, every band in image orientation, stats)"""
        import torch
        dev = torch.device("cuda", self.device)
        nb = len(band_y0)
        if out is None:
            out = torch.zeros((nb, band_rows, cam.width, 3), dtype=torch.float32, device=dev)
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        st = TileStats(); arr = (C.c_int * max(nb, 1))(*[int(y) for y in band_y0])
        _check(self.L.lh_render_ao_bands(self.h, C.byref(cam), nb, arr, int(band_rows), pixel_samples, gather_nsamples, int(seed),
                                         _dptr(out), C.byref(st), C.c_void_p(stream)), "lh_render_ao_bands")
        return out, {k: int(getattr(st, k)) for k, _ in st._fields_}

    def render_pt_tile(self, cam, x0, y0, w, h, spp_begin, spp_count, spp_total, max_vertices=8, kd=0.8,
                       env=(1.0, 1.0, 1.0), seed=1, out=None, stream=None):
        """adds spp_count path-traced samples to `out` (float32 [h,w,3] CUDA, zeroed if None) -> (out, stats)"""
        import torch
        dev = torch.device("cuda", self.device)
        if out is None:
            out = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        st = PtStats(); e = (C.c_float * 3)(*[float(x) for x in env])
        _check(self.L.lh_render_pt_tile(self.h, C.byref(cam), x0, y0, w, h, spp_begin, spp_count, spp_total, max_vertices,
                                        float(kd), C.byref(e), int(seed), _dptr(out), C.byref(st), C.c_void_p(stream)),
               "lh_render_pt_tile")
        return out, {k: int(getattr(st, k)) for k, _ in st._fields_}

    def render_pt_tile2(self, cam, x0, y0, w, h, spp_begin, spp_count, spp_total, max_vertices=8, flags=0, seed=1, out=None,
                        stream=None):
        """as render_pt_tile, with the accelerator's per-mesh materials and environment (set_material / set_environment)"""
        import torch
        dev = torch.device("cuda", self.device)
        if out is None:
            out = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        st = PtStats()
        _check(self.L.lh_render_pt_tile2(self.h, C.byref(cam), x0, y0, w, h, spp_begin, spp_count, spp_total, max_vertices,
                                         int(flags), int(seed), _dptr(out), C.byref(st), C.c_void_p(stream)), "lh_render_pt_tile2")
        return out, {k: int(getattr(st, k)) for k, _ in st._fields_}

    def render_pt_bands(self, cam, y0_first, band_rows, band_stride, nbands, spp_begin, spp_count, spp_total, max_vertices=8, flags=0,
                        override=None, seed=1, out=None, stream=None):
        """lh_render_pt_bands: nbands full-width bands (band k starts at frame line y0_first + k * band_stride) as ONE pass ->
        (float32 [nbands, band_rows, W, 3] CUDA tensor, accumulated into `out`; every band in image orientation; stats).
        override: a Material for every mesh, None: the accelerator's own; environment: the accelerator's."""
        import torch
        dev = torch.device("cuda", self.device)
        if out is None:
            out = torch.zeros((nbands, band_rows, cam.width, 3), dtype=torch.float32, device=dev)
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        st = PtStats()
        _check(self.L.lh_render_pt_bands(self.h, C.byref(cam), int(y0_first), int(band_rows), int(band_stride), int(nbands), int(spp_begin),
                                         int(spp_count), int(spp_total), int(max_vertices), int(flags),
                                         C.byref(override) if override is not None else None, int(seed), _dptr(out), C.byref(st),
                                         C.c_void_p(stream)), "lh_render_pt_bands")
        return out, {k: int(getattr(st, k)) for k, _ in st._fields_}

    def set_material(self, mesh, material):
        _check(self.L.lh_accel_set_material(self.h, int(mesh), C.byref(material)), "lh_accel_set_material")

    def reset_environment(self):
        """back to the default environment of the path tracer (constant white)"""
        _check(self.L.lh_accel_set_environment(self.h, None), "lh_accel_set_environment")
        self._env = None

    def set_environment(self, rgb=(1.0, 1.0, 1.0), envmap=None):
        """envmap: [H,W,4] float32 angular-map light probe or None (constant radiance rgb).  An explicit (0, 0, 0) is black."""
        self._env = (tuple(float(x) for x in rgb), envmap)
        e = Environment()
        for k in range(3):
            e.rgb[k] = float(rgb[k])
        m = None
        if envmap is not None:
            m = np.ascontiguousarray(envmap, np.float32)
            assert m.ndim == 3 and m.shape[2] == 4
            e.map_rgba = m.ctypes.data; e.height, e.width = m.shape[0], m.shape[1]
        _check(self.L.lh_accel_set_environment(self.h, C.byref(e)), "lh_accel_set_environment")

    def set_attribute(self, mesh, kind, data):
        D = _np(data, np.float64)
        if D.ndim != 2:
            raise ValueError("attribute data must be [n, components]")
        _check(self.L.lh_accel_set_attribute(self.h, int(mesh), int(kind), D.ctypes.data, D.shape[1] * 8, D.shape[0]),
               "lh_accel_set_attribute")

    def state_build(self, org, dr, prim, t, u, v):
        """host arrays -> [n, 24] ri_intersection_state_build records (zeros for misses)"""
        o = _np(org, np.float64).reshape(-1, 3); d = _np(dr, np.float64).reshape(-1, 3)
        p = _np(prim, np.uint32); tt = _np(t, np.float64); uu = _np(u, np.float64); vv = _np(v, np.float64)
        st = np.zeros((o.shape[0], STATE_DOUBLES))
        _check(self.L.lh_accel_state_build_host(self.h, o.shape[0], o.ctypes.data, d.ctypes.data, p.ctypes.data, tt.ctypes.data,
                                                uu.ctypes.data, vv.ctypes.data, st.ctypes.data), "lh_accel_state_build_host")
        return st

    def scratch(self, which, dtype, width):
        """view (copy to host) of a scratch buffer of the last render_ao_tile call"""
        p = C.c_void_p(); n = C.c_size_t()
        _check(self.L.lh_render_scratch(self.h, which, C.byref(p), C.byref(n)), "lh_render_scratch")
        count = n.value * width
        host = np.empty(count, dtype)
        if count:
            hip = C.CDLL("libamdhip64.so")
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            assert hip.hipMemcpy(host.ctypes.data, p, host.nbytes, 2) == 0
        return host.reshape(n.value, width) if width > 1 else host


class _AccelView(HipAccel):
    """a replica of a HipMulti as a HipAccel (borrowed handle: closing it does nothing)"""

    def __init__(self, handle, device):        # noqa: D401 -- no lh_accel_create here
        self.L = lib(); self.h = C.c_void_p(handle); self.device = int(device); self.committed = True; self._npos = []

    def close(self):
        self.h = None


class HipMulti:
    """lh_multi_t: the G GPUs of one node from one process -- one host build, replicated; frames through a tile
    queue with the slabs gathered on device 0; ray dumps in contiguous slices."""

    def __init__(self, devices=None):
        self.L = lib()
        self.h = C.c_void_p()
        if devices is None:
            _check(self.L.lh_multi_create(C.byref(self.h), 0, None), "lh_multi_create")
        else:
            arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            _check(self.L.lh_multi_create(C.byref(self.h), len(devices), arr), "lh_multi_create")
        self.n = int(self.L.lh_multi_ndevices(self.h))
        self.devices = list(devices) if devices is not None else list(range(self.n))
        _live.add(self)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.lh_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_mesh(self, positions, indices):
        P = _np(positions, np.float64); I = _np(indices, np.uint32).reshape(-1)
        _check(self.L.lh_multi_add_mesh(self.h, P.shape[0], P.ctypes.data, P.shape[1] * 8, I.shape[0], I.ctypes.data), "lh_multi_add_mesh")

    def set_normals(self, mesh, normals, two_side=0):
        N = _np(normals, np.float64) if normals is not None else None
        _check(self.L.lh_multi_set_normals(self.h, int(mesh), N.ctypes.data if N is not None else None,
                                           (N.shape[1] * 8) if N is not None else 24, int(two_side)), "lh_multi_set_normals")

    def commit(self, build_threads=0):
        _check(self.L.lh_multi_commit(self.h, int(build_threads)), "lh_multi_commit")

    def accel(self, k):
        return _AccelView(self.L.lh_multi_accel(self.h, int(k)), self.devices[k])

    def set_material(self, mesh, material):
        _check(self.L.lh_multi_set_material(self.h, int(mesh), C.byref(material)), "lh_multi_set_material")

    def set_environment(self, rgb=(1.0, 1.0, 1.0), envmap=None):
        e = Environment()
        for k in range(3):
            e.rgb[k] = float(rgb[k])
        m = None
        if envmap is not None:
            m = np.ascontiguousarray(envmap, np.float32)
            e.map_rgba = m.ctypes.data; e.height, e.width = m.shape[0], m.shape[1]
        _check(self.L.lh_multi_set_environment(self.h, C.byref(e)), "lh_multi_set_environment")

    def intersect_host(self, org, dr, mode=MODE_CLOSEST):
        o = _np(org, np.float64).reshape(-1, 3); d = _np(dr, np.float64).reshape(-1, 3)
        n = o.shape[0]
        if mode == MODE_CLOSEST:
            prim = np.empty(n, np.uint32); t = np.empty(n); u = np.empty(n); v = np.empty(n)
            _check(self.L.lh_multi_intersect_host(self.h, n, o.ctypes.data, d.ctypes.data, prim.ctypes.data, t.ctypes.data,
                                                  u.ctypes.data, v.ctypes.data, None, mode), "lh_multi_intersect_host")
            return prim, t, u, v
        occ = np.empty(n, np.uint8)
        _check(self.L.lh_multi_intersect_host(self.h, n, o.ctypes.data, d.ctypes.data, None, None, None, None, occ.ctypes.data, mode),
               "lh_multi_intersect_host")
        return occ

    def render_ao_frame(self, cam, pixel_samples, gather_nsamples, seed=1, tile=512):
        """-> (rgb [H,W,3] float32 host array, stats, per-replica busy seconds)"""
        rgb = np.empty((cam.height, cam.width, 3), np.float32); st = TileStats(); secs = (C.c_double * self.n)()
        _check(self.L.lh_multi_render_ao_frame_host(self.h, C.byref(cam), pixel_samples, gather_nsamples, int(seed), int(tile),
                                                    rgb.ctypes.data, C.byref(st), secs), "lh_multi_render_ao_frame_host")
        return rgb, {k: int(getattr(st, k)) for k, _ in st._fields_}, list(secs)

    def render_pt_frame(self, cam, spp, spp_chunk=0, max_vertices=8, flags=0, seed=1, tile=512):
        rgb = np.empty((cam.height, cam.width, 3), np.float32); st = PtStats(); secs = (C.c_double * self.n)()
        _check(self.L.lh_multi_render_pt_frame_host(self.h, C.byref(cam), int(spp), int(spp_chunk), int(max_vertices), int(flags),
                                                    int(seed), int(tile), rgb.ctypes.data, C.byref(st), secs),
               "lh_multi_render_pt_frame_host")
        return rgb, {k: int(getattr(st, k)) for k, _ in st._fields_}, list(secs)


DIST_RCCL, DIST_SHM = 0, 1


def pack_records16(prim, t, u, v, out, n=None, stream=None):
    """lh_dist_pack_records16: the first n hit records (prim i32/u32, t, u, v f64 CUDA tensors) -> n 16-byte wire records
    {prim u32, t, u, v f32} in `out` (a CUDA uint8 tensor of >= 16 n bytes)"""
    import torch
    n = int(prim.shape[0]) if n is None else int(n)
    L = lib()
    L.lh_dist_pack_records16.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    s = stream if stream is not None else torch.cuda.current_stream(prim.device)
    _check(L.lh_dist_pack_records16(n, prim.data_ptr(), t.data_ptr(), u.data_ptr(), v.data_ptr(), out.data_ptr(), C.c_void_p(s.cuda_stream)),
           "lh_dist_pack_records16")
    return out


class HipDist:
    """lh_dist_t: this process as one rank of a one-process-per-GPU job (RCCL over xGMI through the C ABI; DIST_SHM when the
    ranks share a device).  Rendezvous: `unique_id()` on rank 0 + any out-of-band channel (torch.distributed's store), or a
    fresh file path."""

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * 128)()
        _check(lib().lh_dist_unique_id(buf), "lh_dist_unique_id")
        return bytes(buf)

    def __init__(self, rank, world, device, unique_id=None, rendezvous=None, transport=DIST_RCCL):
        self.L = lib(); self.h = C.c_void_p()
        self.L.lh_dist_init.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
        self.L.lh_dist_init_file.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
        if rendezvous is not None:
            _check(self.L.lh_dist_init_file(C.byref(self.h), str(rendezvous).encode(), int(rank), int(world), int(device)), "lh_dist_init_file")
        else:
            assert unique_id is not None and len(unique_id) == 128
            _check(self.L.lh_dist_init(C.byref(self.h), bytes(unique_id), int(rank), int(world), int(device), int(transport)), "lh_dist_init")
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        self.transport = int(self.L.lh_dist_transport(self.h))
        _live.add(self)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.lh_dist_destroy.argtypes = [C.c_void_p]
            self.L.lh_dist_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def barrier(self):
        self.L.lh_dist_barrier.argtypes = [C.c_void_p]
        _check(self.L.lh_dist_barrier(self.h), "lh_dist_barrier")

    def host_barrier(self):
        """the ranks of one node meet in shared memory (microseconds): the barrier around a timed frame"""
        self.L.lh_dist_host_barrier.argtypes = [C.c_void_p]
        _check(self.L.lh_dist_host_barrier(self.h), "lh_dist_host_barrier")

    @staticmethod
    def _stream_of(tensor, stream):
        """the stream a collective on `tensor` is enqueued on: the caller's, else torch's CURRENT stream of the tensor's device --
        behind whatever torch op produced the tensor (the communicator's private stream has no such dependency)"""
        import torch
        if stream is None:
            cur = torch.cuda.current_stream(tensor.device)
            if cur.cuda_stream == 0:
                # torch's default stream has handle 0, which lh_dist_* reads as "the communicator's own stream" -- a non-blocking
                # stream with no dependency on the legacy default stream: finish the producer first
                cur.synchronize()
            return C.c_void_p(cur.cuda_stream)
        return C.c_void_p(stream.cuda_stream) if hasattr(stream, "cuda_stream") else C.c_void_p(stream)

    def broadcast(self, tensor, stream=None):
        """in-place broadcast of a CUDA tensor from rank 0 (stream: a torch.cuda.Stream or a raw handle; default: torch's current stream)"""
        self.L.lh_dist_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        _check(self.L.lh_dist_broadcast(self.h, tensor.data_ptr(), tensor.numel() * tensor.element_size(), self._stream_of(tensor, stream)), "lh_dist_broadcast")
        return tensor

    def gather(self, tensor, stream=None):
        """equal-sized contiguous CUDA tensors to rank 0 -> [world, ...] there, None elsewhere"""
        import torch
        t = tensor.contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device) if self.rank == 0 else None
        self.L.lh_dist_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        _check(self.L.lh_dist_gather(self.h, t.data_ptr(), t.numel() * t.element_size(), out.data_ptr() if out is not None else None, self._stream_of(t, stream)), "lh_dist_gather")
        torch.cuda.synchronize(t.device)
        return out

    def broadcast_scene(self, acc):
        """rank 0: a committed HipAccel; other ranks: a fresh one (committed on return: no build on those ranks)"""
        self.L.lh_dist_broadcast_scene.argtypes = [C.c_void_p, C.c_void_p]
        _check(self.L.lh_dist_broadcast_scene(self.h, acc.h), "lh_dist_broadcast_scene")
        return acc

    def render_ao_frame(self, acc, cam, pixel_samples, gather_nsamples, seed=1, band_rows=0):
        """-> (frame [H, W, 3] float32 numpy on rank 0 | None, stats)"""
        rgb = np.empty((cam.height, cam.width, 3), np.float32) if self.rank == 0 else None
        st = TileStats()
        self.L.lh_dist_render_ao_frame_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
        _check(self.L.lh_dist_render_ao_frame_host(self.h, acc.h, C.byref(cam), int(pixel_samples), int(gather_nsamples), int(seed), int(band_rows),
                                                   rgb.ctypes.data if rgb is not None else None, C.byref(st)), "lh_dist_render_ao_frame_host")
        return rgb, {"primary_rays": st.primary_rays, "primary_hits": st.primary_hits, "ao_rays": st.ao_rays, "ao_occluded": st.ao_occluded}
