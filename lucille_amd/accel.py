"""ctypes view of the host mirror in include/lucille_accel.h (lh_host.c): lucille's
own plugin API -- ri_geom_* / ri_scene_* / ri_accel_bind / ri_raytrace -- bound to the
HIP accelerator.  Lets tests be written the way a lucille program is."""
import ctypes as C

import numpy as np

from . import binding

RI_ACCEL_UGRID, RI_ACCEL_BVH, RI_ACCEL_HIP = 0, 1, 2
Vec = C.c_double * 4


class RiGeom(C.Structure):
    _fields_ = [("positions", C.POINTER(Vec)), ("npositions", C.c_uint), ("normals", C.POINTER(Vec)),
                ("nnormals", C.c_uint), ("indices", C.POINTER(C.c_uint)), ("nindices", C.c_uint),
                ("two_side", C.c_int)]


class RiRay(C.Structure):
    _fields_ = [("org", Vec), ("dir", Vec), ("t", C.c_float), ("dir_sign", C.c_int * 3), ("invdir", Vec),
                ("thread_num", C.c_int)]


class RiState(C.Structure):
    _fields_ = [("P", Vec), ("Ng", Vec), ("Ns", Vec), ("E", Vec), ("I", Vec), ("t", C.c_double),
                ("inside", C.c_char), ("geom", C.POINTER(RiGeom)), ("index", C.c_uint32), ("color", Vec),
                ("tangent", Vec), ("binormal", Vec), ("stqr", Vec), ("u", C.c_double), ("v", C.c_double)]


class RiAccel(C.Structure):
    _fields_ = [("build", C.c_void_p), ("free", C.c_void_p), ("intersect", C.c_void_p), ("data", C.c_void_p)]


class RiScene(C.Structure):
    _fields_ = [("geom_list", C.POINTER(C.POINTER(RiGeom))), ("ngeoms", C.c_uint), ("accel", C.POINTER(RiAccel))]


class RiRender(C.Structure):
    _fields_ = [("scene", C.POINTER(RiScene)), ("nrays", C.c_uint64), ("device", C.c_int)]


_ready = False


def api():
    global _ready
    L = binding.lib()
    if not _ready:
        L.ri_geom_new.restype = C.POINTER(RiGeom)
        L.ri_geom_free.argtypes = [C.POINTER(RiGeom)]
        L.ri_geom_add_positions.argtypes = [C.POINTER(RiGeom), C.c_uint, C.c_void_p]
        L.ri_geom_add_normals.argtypes = [C.POINTER(RiGeom), C.c_uint, C.c_void_p]
        L.ri_geom_add_indices.argtypes = [C.POINTER(RiGeom), C.c_uint, C.c_void_p]
        L.ri_scene_new.restype = C.POINTER(RiScene)
        L.ri_scene_free.argtypes = [C.POINTER(RiScene)]
        L.ri_scene_add_geom.argtypes = [C.POINTER(RiScene), C.POINTER(RiGeom)]
        L.ri_scene_build_accel.argtypes = [C.POINTER(RiScene)]
        L.ri_accel_new.restype = C.POINTER(RiAccel)
        L.ri_accel_free.argtypes = [C.POINTER(RiAccel)]
        L.ri_accel_bind.argtypes = [C.POINTER(RiAccel), C.c_int]
        L.ri_render_get.restype = C.POINTER(RiRender)
        L.ri_raytrace.argtypes = [C.POINTER(RiRender), C.POINTER(RiRay), C.POINTER(RiState)]
        L.ri_raytrace_batch.argtypes = [C.POINTER(RiRender), C.c_size_t, C.POINTER(RiRay), C.POINTER(RiState),
                                        C.POINTER(C.c_int)]
        L.ri_raytrace_batch.restype = C.c_long
        L.ri_accel_intersect_batch.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 7 + [C.c_int]
        L.ri_accel_prim_lookup.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.POINTER(RiGeom)), C.POINTER(C.c_uint32)]
        L.ri_intersection_state_build.argtypes = [C.POINTER(RiState), C.c_void_p, C.c_void_p]
        _ready = True
    return L


def make_geom(positions, indices, normals=None):
    """ri_geom_new + ri_geom_add_positions/indices (positions padded to double[4])"""
    L = api()
    P = np.zeros((len(positions), 4)); P[:, :3] = np.asarray(positions, np.float64)[:, :3]
    I = np.ascontiguousarray(indices, np.uint32)
    g = L.ri_geom_new()
    L.ri_geom_add_positions(g, P.shape[0], P.ctypes.data)
    L.ri_geom_add_indices(g, I.shape[0], I.ctypes.data)
    if normals is not None:
        N = np.zeros((len(normals), 4)); N[:, :3] = np.asarray(normals, np.float64)[:, :3]
        L.ri_geom_add_normals(g, N.shape[0], N.ctypes.data)
    return g
