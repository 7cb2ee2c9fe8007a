"""ctypes bindings for the CHECKERS under oracle/ (test infrastructure only).

  Oracle  -> oracle/liblucille_oracle.so   this repo's CPU restatement
  RefLib  -> oracle/_ref/liblucille_ref*.so the compiled reference (built only
             where /root/reference exists; the .so travels to the GPU box)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  Nothing under lucille_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MISS = 0xFFFFFFFF

_dp = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)


def _p(a, ty):
    return a.ctypes.data_as(ty) if a is not None else None


def build_oracle(force=False):
    so = os.path.join(HERE, "liblucille_oracle.so")
    srcs = [os.path.join(HERE, f) for f in ("lucille_oracle.c", "lucille_oracle_ao.c", "lucille_oracle_beam.c", "lucille_oracle_pt.c", "lucille_oracle.h")]
    stale = (not os.path.exists(so)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", HERE, "liblucille_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def build_ref():
    """(Re)build oracle/_ref from /root/reference when that tree is present."""
    if os.path.isdir(os.environ.get("LUCILLE_REF", "/root/reference") + "/src/render"):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)
        return True
    return False


def ref_available(stat=False):
    return os.path.exists(os.path.join(HERE, "_ref", "liblucille_ref_stat.so" if stat else "liblucille_ref.so"))


class Counters(C.Structure):
    _fields_ = [("ninner", C.c_uint64), ("nleaf", C.c_uint64), ("ntested", C.c_uint64),
                ("nhit", C.c_uint64), ("nrays", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class TreeStats(C.Structure):
    _fields_ = [("ninner", C.c_uint64), ("nleaf", C.c_uint64), ("max_depth", C.c_uint64),
                ("max_leaf_tris", C.c_uint64), ("ntriangles", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Camera(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("rh", C.c_int), ("ortho", C.c_int),
                ("flength", C.c_double), ("cam2world", C.c_double * 16)]

    @classmethod
    def from_ref(cls, cam20, width=None, height=None):
        """from the 20 doubles lref_camera_get captures: c2w[16], flength, w, h, is_rh"""
        c = cls()
        for i in range(16):
            c.cam2world[i] = float(cam20[i])
        c.flength = float(cam20[16])
        c.width = int(cam20[17]) if width is None else width
        c.height = int(cam20[18]) if height is None else height
        c.rh = int(cam20[19])
        return c


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build_oracle())
        L.lo_scene_new.restype = C.c_void_p
        L.lo_scene_free.argtypes = [C.c_void_p]
        L.lo_scene_add_mesh.argtypes = [C.c_void_p, C.c_uint32, _dp, C.c_uint32, _u32p]
        L.lo_scene_build.argtypes = [C.c_void_p]
        L.lo_scene_ntriangles.argtypes = [C.c_void_p]
        L.lo_scene_ntriangles.restype = C.c_uint64
        L.lo_scene_tree_stats.argtypes = [C.c_void_p, C.POINTER(TreeStats)]
        L.lo_scene_bbox.argtypes = [C.c_void_p, _dp, _dp]
        L.lo_scene_get_triangles.argtypes = [C.c_void_p, _dp, _u32p, _u32p]
        L.lo_intersect_batch.argtypes = [C.c_void_p, C.c_size_t, _dp, _dp, _u32p, _dp, _dp, _dp,
                                         C.POINTER(Counters), C.c_int]
        L.lo_brute_force_batch.argtypes = [C.c_void_p, C.c_size_t, _dp, _dp, _u32p, _dp, _dp, _dp, C.c_int]
        L.lo_beam_visibility_batch.argtypes = [C.c_void_p, C.c_size_t, _dp, _dp, C.POINTER(C.c_int32)]
        L.lo_scene_leaf_order.argtypes = [C.c_void_p, _u32p, _u32p]
        L.lo_beam_raster.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, _dp, _u64p]
        L.lo_count_equal_t_batch.argtypes = [C.c_void_p, C.c_size_t, _dp, _dp, _dp, _u32p]
        L.lo_scene_set_normals.argtypes = [C.c_void_p, C.c_uint32, _dp, C.c_int]
        L.lo_camera_ray.argtypes = [C.POINTER(Camera), C.c_double, C.c_double, _dp, _dp]
        L.lo_state_build.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_double, _dp, _dp, _dp, _dp, _dp,
                                     C.POINTER(C.c_int)]
        L.lo_ao_rays.argtypes = [_dp, _dp, C.c_uint32, C.c_uint32, _dp, _dp, _dp]
        L.lo_mt_stream.argtypes = [C.c_ulong, C.c_size_t, _dp]
        L.lo_bucket_order.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint)]
        L.lo_render_ao.restype = C.c_size_t
        L.lo_render_ao.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(C.c_float), _dp, _dp, _u32p, _dp, _dp, _dp, C.c_size_t]
        L.lo_scene_set_attribute.argtypes = [C.c_void_p, C.c_uint32, C.c_int, _dp]
        L.lo_state_batch.argtypes = [C.c_void_p, C.c_size_t, _dp, _dp, _u32p, _dp]
        L.lo_render_pt.restype = C.c_uint64
        L.lo_render_pt.argtypes = [C.c_void_p, C.POINTER(Camera)] + [C.c_int] * 8 + [_u32p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                   C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_uint64,
                                   C.POINTER(C.c_float), C.POINTER(C.c_uint16), _u64p]
        L.lo_soup_triangles.argtypes = [_u64p, C.c_uint32, C.c_double, _dp, _u32p]
        L.lo_soup_rays.argtypes = [_u64p, C.c_size_t, _dp, _dp]
        _lib = L
    return _lib


SOUP_SEED = 88172645463325252


def soup(ntri, nrays, half_extent=0.005, seed=SOUP_SEED):
    """S-soup generator: triangles then rays from ONE
    xorshift64 stream.  Returns (positions[3*ntri,3], indices[3*ntri], org[n,3], dir[n,3])."""
    L = lib()
    st = C.c_uint64(seed)
    P = np.empty((3 * ntri, 3), np.float64)
    idx = np.empty(3 * ntri, np.uint32)
    L.lo_soup_triangles(C.byref(st), ntri, half_extent, _p(P, _dp), _p(idx, _u32p))
    org = np.empty((nrays, 3), np.float64)
    dr = np.empty((nrays, 3), np.float64)
    L.lo_soup_rays(C.byref(st), nrays, _p(org, _dp), _p(dr, _dp))
    return P, idx, org, dr


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class Oracle:
    """The CPU restatement (oracle/lucille_oracle.c)."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.lo_scene_new())

    def __del__(self):
        try:
            self.L.lo_scene_free(self.h)
        except Exception:
            pass

    def add_mesh(self, positions, indices):
        P = _c(positions, np.float64).reshape(-1, 3)
        I = _c(indices, np.uint32).reshape(-1)
        self.L.lo_scene_add_mesh(self.h, P.shape[0], _p(P, _dp), I.shape[0], _p(I, _u32p))

    def set_normals(self, mesh, normals, two_side=0):
        N = _c(normals, np.float64).reshape(-1, 3) if normals is not None else None
        self.L.lo_scene_set_normals(self.h, mesh, _p(N, _dp), int(two_side))

    def set_attribute(self, mesh, kind, data):
        """kind 0 colors, 1 tangents, 2 binormals ([npos,3]); 3 texcoords ([npos,2]); 4 texcoords_unshared ([nidx,2])"""
        D = _c(data, np.float64)
        assert self.L.lo_scene_set_attribute(self.h, int(mesh), int(kind), _p(D, _dp)) == 0

    def state_batch(self, org, dr):
        """closest hit + the whole ri_intersection_state_build record (24 doubles, LH_STATE_DOUBLES layout)"""
        org = _c(org, np.float64).reshape(-1, 3); dr = _c(dr, np.float64).reshape(-1, 3)
        prim = np.empty(org.shape[0], np.uint32); st = np.zeros((org.shape[0], 24))
        self.L.lo_state_batch(self.h, org.shape[0], _p(org, _dp), _p(dr, _dp), _p(prim, _u32p), _p(st, _dp))
        return prim, st

    def build(self):
        self.L.lo_scene_build(self.h)

    def render_ao(self, cam, pixel_samples, gather_nsamples, record=True, bucket_size=32):
        """the reference's frame loop + AO transport, single thread (lo_render_ao)"""
        W, H = cam.width, cam.height
        img = np.zeros((H, W, 3), np.float32)
        n_ao = int(np.sqrt(gather_nsamples)) ** 2
        cap = W * H * pixel_samples * pixel_samples * (1 + n_ao) if record else 0
        ro = np.empty((cap, 3)); rd = np.empty((cap, 3)); rp = np.empty(cap, np.uint32)
        rt = np.empty(cap); ru = np.empty(cap); rv = np.empty(cap)
        n = self.L.lo_render_ao(self.h, C.byref(cam), pixel_samples, pixel_samples, gather_nsamples, bucket_size,
                                img.ctypes.data_as(C.POINTER(C.c_float)), _p(ro, _dp) if record else None,
                                _p(rd, _dp) if record else None, _p(rp, _u32p) if record else None,
                                _p(rt, _dp) if record else None, _p(ru, _dp) if record else None,
                                _p(rv, _dp) if record else None, cap)
        if not record:
            return img, n
        return img, {"org": ro[:n], "dir": rd[:n], "prim": rp[:n], "t": rt[:n], "u": ru[:n], "v": rv[:n]}

    def render_pt(self, cam, x0, y0, w, h, s0, spp, spp_total, max_vertices=8, materials=None, override=None,
                  env_rgb=(1.0, 1.0, 1.0), env_map=None, ref_weights=0, seed=1, out=None):
        """lo_render_pt (lucille_oracle_pt.c): the path-traced tile, one path at a time.  materials: [nmesh, 10] float32
        (kd, ks, kt, ior) or override: 10 floats for every mesh.  -> (rgb [h, w, 3] float32 (accumulated into `out`),
        {"rays", "max_depth_reached", "paths"}, rays per path [h * w, spp] uint16)"""
        fp = C.POINTER(C.c_float)
        rgb = np.zeros((h, w, 3), np.float32) if out is None else out
        _, g, _ = self.triangles()
        g = _c(g, np.uint32)
        mats = None if materials is None else _c(materials, np.float32).reshape(-1, 10)
        ov = None if override is None else _c(override, np.float32).reshape(10)
        assert (mats is None) != (ov is None)
        if mats is not None:
            assert mats.shape[0] > int(g.max())
        env = _c(env_rgb, np.float32).reshape(3)
        em = None if env_map is None else _c(env_map, np.float32)
        per = np.zeros((h * w, spp), np.uint16); longest = C.c_uint64(0)
        rays = self.L.lo_render_pt(self.h, C.byref(cam), x0, y0, w, h, s0, spp, spp_total, max_vertices, _p(g, _u32p),
                                   None if mats is None else mats.ctypes.data_as(fp), None if ov is None else ov.ctypes.data_as(fp),
                                   env.ctypes.data_as(fp), None if em is None else em.ctypes.data_as(fp),
                                   0 if em is None else em.shape[1], 0 if em is None else em.shape[0], int(ref_weights), int(seed),
                                   rgb.ctypes.data_as(fp), per.ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(longest))
        return rgb, {"paths": w * h * spp, "rays": int(rays), "max_depth_reached": int(longest.value)}, per

    def render_ptref(self, cam, x0, y0, w, h, nsamples, max_vertices=0, materials=None, override=None, env_rgb=(1.0, 1.0, 1.0),
                     env_map=None, mt_seed=4357):
        """lo_render_ptref (lucille_oracle_ptref.c): src/transport/pathtrace.c AS WRITTEN -> (rgb [h, w, 3] float32, rays traced)"""
        fp = C.POINTER(C.c_float)
        rgb = np.zeros((h, w, 3), np.float32)
        _, g, _ = self.triangles()
        g = _c(g, np.uint32)
        mats = None if materials is None else _c(materials, np.float32).reshape(-1, 10)
        ov = None if override is None else _c(override, np.float32).reshape(10)
        assert (mats is None) != (ov is None)
        env = _c(env_rgb, np.float32).reshape(3)
        em = None if env_map is None else _c(env_map, np.float32)
        self.L.lo_render_ptref.restype = C.c_uint64
        self.L.lo_render_ptref.argtypes = [C.c_void_p, C.POINTER(Camera)] + [C.c_int] * 6 + [_u32p, fp, fp, fp, fp, C.c_int, C.c_int, C.c_ulong, fp]
        rays = self.L.lo_render_ptref(self.h, C.byref(cam), x0, y0, w, h, int(nsamples), int(max_vertices), _p(g, _u32p),
                                      None if mats is None else mats.ctypes.data_as(fp), None if ov is None else ov.ctypes.data_as(fp),
                                      env.ctypes.data_as(fp), None if em is None else em.ctypes.data_as(fp),
                                      0 if em is None else em.shape[1], 0 if em is None else em.shape[0], int(mt_seed), rgb.ctypes.data_as(fp))
        return rgb, int(rays)

    @property
    def ntriangles(self):
        return int(self.L.lo_scene_ntriangles(self.h))

    def tree_stats(self):
        s = TreeStats()
        self.L.lo_scene_tree_stats(self.h, C.byref(s))
        return s.as_dict()

    def bbox(self):
        a = np.empty(3); b = np.empty(3)
        self.L.lo_scene_bbox(self.h, _p(a, _dp), _p(b, _dp))
        return a, b

    def triangles(self):
        n = self.ntriangles
        v9 = np.empty((n, 9), np.float64); g = np.empty(n, np.uint32); ix = np.empty(n, np.uint32)
        self.L.lo_scene_get_triangles(self.h, _p(v9, _dp), _p(g, _u32p), _p(ix, _u32p))
        return v9, g, ix

    def _run(self, fn, org, dr, counters, nthreads):
        org = _c(org, np.float64).reshape(-1, 3); dr = _c(dr, np.float64).reshape(-1, 3)
        n = org.shape[0]
        prim = np.empty(n, np.uint32); t = np.empty(n); u = np.empty(n); v = np.empty(n)
        if fn is self.L.lo_intersect_batch:
            c = Counters() if counters else None
            fn(self.h, n, _p(org, _dp), _p(dr, _dp), _p(prim, _u32p), _p(t, _dp), _p(u, _dp), _p(v, _dp),
               C.byref(c) if c is not None else None, nthreads)
            return (prim, t, u, v, c.as_dict()) if counters else (prim, t, u, v)
        fn(self.h, n, _p(org, _dp), _p(dr, _dp), _p(prim, _u32p), _p(t, _dp), _p(u, _dp), _p(v, _dp), nthreads)
        return prim, t, u, v

    def intersect(self, org, dr, counters=False, nthreads=1):
        return self._run(self.L.lo_intersect_batch, org, dr, counters, nthreads)

    def beam_visibility(self, org, dirs):
        org = _c(org, np.float64).reshape(-1, 3); dirs = _c(dirs, np.float64).reshape(-1, 4, 3)
        res = np.empty(org.shape[0], np.int32)
        self.L.lo_beam_visibility_batch(self.h, org.shape[0], _p(org, _dp), _p(dirs, _dp), res.ctypes.data_as(C.POINTER(C.c_int32)))
        return res

    def beam_raster(self, org, dirs, width, height, frame, corner, eye, fov):
        """ri_beam_set + ri_raster_plane_setup + ri_bvh_intersect_beam for ONE beam -> (rc, t[height, width], flags[4]);
        rc -1: ri_beam_set refuses; flags: lo_beam_raster (lucille_oracle.h)"""
        a = [_c(x, np.float64).reshape(-1) for x in (org, dirs, frame, corner, eye)]
        t = np.zeros((height, width)); fl = np.zeros(4, np.uint64)
        rc = self.L.lo_beam_raster(self.h, _p(a[0], _dp), _p(a[1], _dp), int(width), int(height), _p(a[2], _dp), _p(a[3], _dp),
                                   _p(a[4], _dp), float(fov), _p(t, _dp), _p(fl, _u64p))
        return rc, t, fl

    def leaf_order(self):
        n = self.ntriangles
        lp = np.empty(n, np.uint32); pf = np.empty(n, np.uint32)
        self.L.lo_scene_leaf_order(self.h, _p(lp, _u32p), _p(pf, _u32p))
        return lp, pf

    def count_equal_t(self, org, dr, t_ref):
        """per ray: number of triangles hit at exactly t_ref (brute force, small scenes)"""
        org = _c(org, np.float64).reshape(-1, 3); dr = _c(dr, np.float64).reshape(-1, 3)
        t_ref = _c(t_ref, np.float64)
        cnt = np.zeros(org.shape[0], np.uint32)
        self.L.lo_count_equal_t_batch(self.h, org.shape[0], _p(org, _dp), _p(dr, _dp), _p(t_ref, _dp), _p(cnt, _u32p))
        return cnt

    def brute_force(self, org, dr, nthreads=1):
        return self._run(self.L.lo_brute_force_batch, org, dr, False, nthreads)


class RefLib:
    """The compiled reference (oracle/_ref).  One global scene per process."""

    def __init__(self, stat=False):
        name = "liblucille_ref_stat.so" if stat else "liblucille_ref.so"
        path = os.path.join(HERE, "_ref", name)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        L = C.CDLL(path)
        L.lref_scene_add_mesh.argtypes = [C.c_uint32, _dp, C.c_uint32, _u32p]
        L.lref_intersect_batch.argtypes = [C.c_size_t, _dp, _dp, _u32p, _dp, _dp, _dp, _dp]
        L.lref_tree_stats.argtypes = [_u64p]
        L.lref_counters_get.argtypes = [_u64p]
        L.lref_scene_bbox.argtypes = [_dp, _dp]
        L.lref_record_stop.restype = C.c_size_t
        L.lref_record_size.restype = C.c_size_t
        L.lref_record_copy.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        L.lref_beam_visibility_batch.argtypes = [C.c_size_t, _dp, _dp, C.POINTER(C.c_int32)]
        L.lref_scene_set_attribute.argtypes = [C.c_uint32, C.c_int, _dp, C.c_uint32, C.c_int]
        L.lref_state_batch.argtypes = [C.c_size_t, _dp, _dp, _u32p, _dp]
        self.L = L
        L.lref_init()
        L.lref_scene_reset()

    def reset(self):
        self.L.lref_scene_reset()

    def add_mesh(self, positions, indices):
        P = _c(positions, np.float64).reshape(-1, 3)
        I = _c(indices, np.uint32).reshape(-1)
        self.L.lref_scene_add_mesh(P.shape[0], _p(P, _dp), I.shape[0], _p(I, _u32p))

    def set_attribute(self, mesh, kind, data, two_side=0):
        """kind -1 normals (+ two_side), 0 colors, 1 tangents, 2 binormals, 3 texcoords, 4 texcoords_unshared"""
        D = _c(data, np.float64) if data is not None else None
        n = 0 if D is None else D.shape[0]
        assert self.L.lref_scene_set_attribute(int(mesh), int(kind), _p(D, _dp), n, int(two_side)) == 0

    def state_batch(self, org, dr):
        org = _c(org, np.float64).reshape(-1, 3); dr = _c(dr, np.float64).reshape(-1, 3)
        prim = np.empty(org.shape[0], np.uint32); st = np.zeros((org.shape[0], 24))
        self.L.lref_state_batch(org.shape[0], _p(org, _dp), _p(dr, _dp), _p(prim, _u32p), _p(st, _dp))
        return prim, st

    def build(self):
        assert self.L.lref_scene_build() == 0

    def tree_stats(self):
        o = np.zeros(5, np.uint64)
        self.L.lref_tree_stats(_p(o, _u64p))
        return dict(zip(("ninner", "nleaf", "max_depth", "max_leaf_tris", "ntriangles"), map(int, o)))

    def bbox(self):
        a = np.empty(3); b = np.empty(3)
        self.L.lref_scene_bbox(_p(a, _dp), _p(b, _dp))
        return a, b

    def beam_visibility(self, org, dirs):
        org = _c(org, np.float64).reshape(-1, 3); dirs = _c(dirs, np.float64).reshape(-1, 4, 3)
        res = np.empty(org.shape[0], np.int32)
        self.L.lref_beam_visibility_batch(org.shape[0], _p(org, _dp), _p(dirs, _dp), res.ctypes.data_as(C.POINTER(C.c_int32)))
        return res

    def beam_raster(self, org, dirs, width, height, frame, corner, eye, fov, invalidate=True):
        """the reference's own beam-raster path for ONE beam (ref_harness.c lref_beam_raster) -> (rc, t[height, width]).
        NOT SAFE in-process: a beam whose footprint leaves the raster window makes the reference write outside plane->t and
        its asserts abort -- use ref_beam_raster_child()."""
        self.L.lref_beam_raster.argtypes = [_dp, _dp, C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, C.c_int, _dp]
        a = [_c(x, np.float64).reshape(-1) for x in (org, dirs, frame, corner, eye)]
        t = np.zeros((height, width))
        rc = self.L.lref_beam_raster(_p(a[0], _dp), _p(a[1], _dp), int(width), int(height), _p(a[2], _dp), _p(a[3], _dp),
                                     _p(a[4], _dp), float(fov), 1 if invalidate else 0, _p(t, _dp))
        return rc, t

    def intersect(self, org, dr, state=False, counters=False):
        org = _c(org, np.float64).reshape(-1, 3); dr = _c(dr, np.float64).reshape(-1, 3)
        n = org.shape[0]
        prim = np.empty(n, np.uint32); t = np.empty(n); u = np.empty(n); v = np.empty(n)
        st = np.empty((n, 9)) if state else None
        if counters:
            self.L.lref_counters_clear()
        self.L.lref_intersect_batch(n, _p(org, _dp), _p(dr, _dp), _p(prim, _u32p), _p(t, _dp), _p(u, _dp),
                                    _p(v, _dp), _p(st, _dp))
        out = [prim, t, u, v]
        if state:
            out.append(st)
        if counters:
            o = np.zeros(5, np.uint64)
            self.L.lref_counters_get(_p(o, _u64p))
            out.append(dict(zip(("ninner", "nleaf", "ntested", "nhit", "nrays"), map(int, o))))
        return tuple(out)


# ---- beam-raster helpers (test infrastructure) ---------------------------------------------------------------------
def beam_camera(eye, lookat, up, width, height, fov):
    """the testbed's camera set-up (src/testbed/simplerender.cpp:37-70): -> corner, du, dv, dw"""
    eye = np.asarray(eye, np.float64); lookat = np.asarray(lookat, np.float64); up = np.asarray(up, np.float64)
    flen = 0.5 * width / np.tan(0.5 * (fov * np.pi / 180.0))
    dw = lookat - eye
    du = np.cross(dw, up); du = du / np.linalg.norm(du)
    dv = np.cross(dw, du); dv = dv / np.linalg.norm(dv)
    dw = dw / np.linalg.norm(dw)
    return flen * dw - 0.5 * (width * du + height * dv), du, dv, dw


def beam_dirs(corner, du, dv, size, s, t):
    """the four corner directions of the beam over pixels [s, s + size) x [t, t + size) (simplerender.cpp:88-120)"""
    return np.array([corner + (s + a) * du + (t + b) * dv for a, b in ((0, 0), (size, 0), (size, size), (0, size))])


def _ref_beam_child(meshes, beams, q):
    ref = RefLib()
    for P, idx in meshes:
        ref.add_mesh(P, idx)
    ref.build()
    out = []
    for b in beams:
        out.append(ref.beam_raster(*b))
        q.put((len(out) - 1, out[-1]))
    q.put(None)


def ref_beam_raster_child(meshes, beams, timeout=120):
    """Run the compiled reference's beam-raster path in a CHILD process, one scene, a list of beams (argument tuples of
    RefLib.beam_raster).  -> list of (rc, t) per beam, None from the first beam on that killed the child (assert, heap
    corruption, hang)."""
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=_ref_beam_child, args=(meshes, beams, q))
    p.start()
    res = [None] * len(beams)
    try:
        while True:
            item = q.get(timeout=timeout)
            if item is None:
                break
            res[item[0]] = item[1]
    except Exception:
        pass
    p.join(5)
    if p.is_alive():
        p.kill(); p.join()
    return res
