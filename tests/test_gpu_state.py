"""GPU: the device-side hit epilogue (lh_accel_state_build_*: P, Ng, Ns, tangent, binormal, colour, st, I, inside from
SoA per-primitive attributes) against the compiled reference's records and the oracle, bit for bit; and the host
mirror of lucille's plugin API carrying the same attributes (ri_geom_add_colors ... -> ri_raytrace -> state)."""
import ctypes as C

import numpy as np
import pytest

import lucille_amd as la
from lucille_amd import binding
from oracle import pyoracle as po
from tests.golden.make_golden import apply_state_scene, state_scene
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu


class _Prod:
    """adapter: the fixture's scene description onto a HipAccel"""

    def __init__(self):
        self.acc = la.HipAccel(0)

    def add_mesh(self, P, idx):
        self.acc.add_mesh(P, idx)

    def set_normals(self, k, N, two_side):
        self.acc.set_normals(k, N, two_side)

    def set_attribute(self, k, kind, data):
        self.acc.set_attribute(k, kind, data)


def test_device_state_records_equal_the_reference():
    g = load_golden("state_attr")
    meshes, org, dr = state_scene(int(g["seed"]))
    p = _Prod(); apply_state_scene(p, meshes, False); p.acc.commit()
    prim, t, u, v = p.acc.intersect_host(org, dr)
    assert np.array_equal(prim, g["prim"])
    st = p.acc.state_build(org, dr, prim, t, u, v)
    assert np.array_equal(st, g["state"])
    # a second seed against the oracle (no fixture): same arithmetic on other data
    meshes, org, dr = state_scene(5)
    p2 = _Prod(); apply_state_scene(p2, meshes, False); p2.acc.commit()
    o = po.Oracle(); apply_state_scene(o, meshes, False); o.build()
    prim, t, u, v = p2.acc.intersect_host(org, dr)
    op, ost = o.state_batch(org, dr)
    assert np.array_equal(prim, op) and np.array_equal(p2.acc.state_build(org, dr, prim, t, u, v), ost)
    with pytest.raises(la.LucilleHipError, match="needs"):
        bad = la.HipAccel(0); bad.add_mesh(meshes[0]["P"], meshes[0]["idx"]); bad.set_attribute(0, la.ATTR_COLOR, np.zeros((3, 3)))
    p.acc.close(); p2.acc.close()


def test_host_mirror_state_carries_the_attributes():
    """include/lucille_accel.h: ri_geom_add_colors / _tangents / _binormals / _texcoords(_unshared) ->
    ri_scene_build_accel(RI_ACCEL_HIP) -> ri_raytrace: the ri_intersection_state_t the caller gets == the reference's"""
    g = load_golden("state_attr")
    meshes, org, dr = state_scene(int(g["seed"]))
    L = binding.lib()
    vec4 = lambda a: np.ascontiguousarray(np.concatenate([a, np.zeros((a.shape[0], 1))], 1))

    class Geom(C.Structure):
        _fields_ = [("positions", C.c_void_p), ("npositions", C.c_uint), ("normals", C.c_void_p), ("nnormals", C.c_uint),
                    ("indices", C.c_void_p), ("nindices", C.c_uint), ("two_side", C.c_int),
                    ("tangents", C.c_void_p), ("ntangents", C.c_uint), ("binormals", C.c_void_p), ("nbinormals", C.c_uint),
                    ("colors", C.c_void_p), ("ncolors", C.c_uint), ("texcoords", C.c_void_p), ("texcoords_unshared", C.c_void_p),
                    ("ntexcoords", C.c_uint)]

    class State(C.Structure):
        _fields_ = [("P", C.c_double * 4), ("Ng", C.c_double * 4), ("Ns", C.c_double * 4), ("E", C.c_double * 4), ("I", C.c_double * 4),
                    ("t", C.c_double), ("inside", C.c_char), ("geom", C.c_void_p), ("index", C.c_uint32),
                    ("color", C.c_double * 4), ("tangent", C.c_double * 4), ("binormal", C.c_double * 4), ("stqr", C.c_double * 4),
                    ("u", C.c_double), ("v", C.c_double)]

    class Ray(C.Structure):
        _fields_ = [("org", C.c_double * 4), ("dir", C.c_double * 4), ("t", C.c_float), ("dir_sign", C.c_int * 3),
                    ("invdir", C.c_double * 4), ("thread_num", C.c_int)]

    L.ri_geom_new.restype = C.POINTER(Geom)
    L.ri_scene_new.restype = C.c_void_p
    L.ri_render_get.restype = C.c_void_p
    for f in ("ri_geom_add_positions", "ri_geom_add_normals", "ri_geom_add_tangents", "ri_geom_add_binormals", "ri_geom_add_colors"):
        getattr(L, f).argtypes = [C.POINTER(Geom), C.c_uint, C.c_void_p]
    L.ri_geom_add_texcoords.argtypes = [C.POINTER(Geom), C.c_uint, C.c_void_p]
    L.ri_geom_add_texcoords_unshared.argtypes = [C.POINTER(Geom), C.c_uint, C.c_void_p]
    L.ri_geom_add_indices.argtypes = [C.POINTER(Geom), C.c_uint, C.c_void_p]
    L.ri_scene_add_geom.argtypes = [C.c_void_p, C.POINTER(Geom)]
    L.ri_scene_build_accel.argtypes = [C.c_void_p]
    L.ri_accel_bind.argtypes = [C.c_void_p, C.c_int]
    L.ri_raytrace.argtypes = [C.c_void_p, C.POINTER(Ray), C.POINTER(State)]
    L.ri_render_init()

    class Render(C.Structure):
        _fields_ = [("scene", C.c_void_p)]
    class Scene(C.Structure):
        _fields_ = [("geom_list", C.c_void_p), ("ngeoms", C.c_uint), ("accel", C.c_void_p)]
    render = C.cast(L.ri_render_get(), C.POINTER(Render))
    scene = render.contents.scene
    keep = []
    for m in meshes:
        gm = L.ri_geom_new()
        P4 = vec4(m["P"]); keep.append(P4)
        L.ri_geom_add_positions(gm, P4.shape[0], P4.ctypes.data)
        idx = np.ascontiguousarray(m["idx"], np.uint32); keep.append(idx)
        L.ri_geom_add_indices(gm, idx.shape[0], idx.ctypes.data)
        for key, fn in (("N", L.ri_geom_add_normals), ("T", L.ri_geom_add_tangents), ("B", L.ri_geom_add_binormals), ("C", L.ri_geom_add_colors)):
            if key in m:
                a = vec4(m[key]); keep.append(a); fn(gm, a.shape[0], a.ctypes.data)
        if "ST" in m:
            a = np.ascontiguousarray(m["ST"]); keep.append(a); L.ri_geom_add_texcoords(gm, a.shape[0], a.ctypes.data)
        if "STU" in m:
            a = np.ascontiguousarray(m["STU"]); keep.append(a); L.ri_geom_add_texcoords_unshared(gm, a.shape[0], a.ctypes.data)
        gm.contents.two_side = m["two_side"]
        L.ri_scene_add_geom(scene, gm)
    sc = C.cast(scene, C.POINTER(Scene))
    assert L.ri_accel_bind(sc.contents.accel, 2) == 0 and L.ri_scene_build_accel(scene) == 0
    exp_p, exp = g["prim"], g["state"]
    n = 600
    for i in range(n):
        ray = Ray(); st = State()
        for k in range(3):
            ray.org[k] = org[i, k]; ray.dir[k] = dr[i, k]
        hit = L.ri_raytrace(render, C.byref(ray), C.byref(st))
        assert bool(hit) == (exp_p[i] != po.MISS)
        if hit:
            got = np.concatenate([st.P[:3], st.Ng[:3], st.Ns[:3], st.tangent[:3], st.binormal[:3], st.color[:3], st.stqr[:2], st.I[:3],
                                  [float(ord(st.inside))]])
            assert np.array_equal(got, exp[i]), (i, got, exp[i])
    L.ri_render_free()
