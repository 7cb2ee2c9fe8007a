"""Shared test helpers (test infrastructure)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MODEL_DIR = os.path.join(ROOT, "tests", "cpu_model")
CSRC = os.path.join(ROOT, "lucille_amd", "csrc")

_dp = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)


def _fma_flag():
    try:
        return ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []
    except OSError:
        return []


def build_model():
    """tests/cpu_model/liblh_model.so: host model of the kernel algorithm over
    the product's own BVH builder (lh_bvh.c) and arithmetic (lh_filter.h)."""
    so = os.path.join(MODEL_DIR, "liblh_model.so")
    srcs = [os.path.join(MODEL_DIR, "lh_model.c"), os.path.join(CSRC, "lh_bvh.c"), os.path.join(CSRC, "lh_refbvh.c"),
            os.path.join(CSRC, "lh_hostwalk.c"),           # the product's one-ray host walk, as it is (lhm_hostwalk)
            os.path.join(CSRC, "lh_bvh.h"), os.path.join(CSRC, "lh_filter.h"), os.path.join(CSRC, "lh_refbvh.h"),
            os.path.join(CSRC, "lh_reftrace.h")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        tmp = "%s.%d.tmp" % (so, os.getpid())              # pytest-xdist workers build at once: link aside, rename whole
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fPIC", "-shared"] + _fma_flag() +
                              ["-I" + CSRC, srcs[0], srcs[1], srcs[2], srcs[3], "-o", tmp, "-lm", "-lpthread"])
        os.replace(tmp, so)
    return so


class Model:
    """Host model of the HIP kernel (see tests/cpu_model/lh_model.c)."""
    _L = None

    @classmethod
    def lib(cls):
        if cls._L is None:
            L = C.CDLL(build_model())
            L.lhm_build.restype = C.c_void_p
            L.lhm_build.argtypes = [C.c_uint32, _dp, C.c_uint32, _u32p, C.c_int]
            L.lhm_free.argtypes = [C.c_void_p]
            L.lhm_info.argtypes = [C.c_void_p, _u32p]
            L.lhm_nodes.restype = C.c_void_p
            L.lhm_nodes.argtypes = [C.c_void_p]
            L.lhm_tri32.restype = C.c_void_p
            L.lhm_tri32.argtypes = [C.c_void_p]
            L.lhm_trace.argtypes = [C.c_void_p, C.c_size_t, _dp, _dp, _u32p, _dp, _dp, _dp,
                                    C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint64), C.c_int]
            cls._L = L
        return cls._L

    def __init__(self, positions, indices, nthreads=4):
        L = self.lib()
        P = np.ascontiguousarray(positions, np.float64).reshape(-1, 3)
        I = np.ascontiguousarray(indices, np.uint32).reshape(-1)
        self.h = L.lhm_build(P.shape[0], P.ctypes.data_as(_dp), I.shape[0], I.ctypes.data_as(_u32p), nthreads)
        assert self.h, "lh_bvh_build failed"
        info = np.zeros(4, np.uint32)
        L.lhm_info(self.h, info.ctypes.data_as(_u32p))
        self.ntris, self.nnodes, self.max_depth, self.nleaves = map(int, info)

    def __del__(self):
        try:
            self.lib().lhm_free(self.h)
        except Exception:
            pass

    # ---- reference-order tree (lh_refbvh.c) ------------------------------------
    def ref_build(self, nthreads=4, use_for_ties=True):
        L = self.lib()
        L.lhm_ref_build.restype = C.c_void_p; L.lhm_ref_build.argtypes = [C.c_void_p, C.c_int]
        L.lhm_ref_use.argtypes = [C.c_void_p]; L.lhm_ref_info.argtypes = [C.c_void_p, _u32p]
        L.lhm_ref_leaf_order.argtypes = [C.c_void_p, _u32p, _u32p]; L.lhm_ref_bbox.argtypes = [C.c_void_p, _dp]
        self.ref = L.lhm_ref_build(self.h, nthreads)
        assert self.ref
        L.lhm_ref_use(self.ref if use_for_ties else None)
        info = np.zeros(4, np.uint32); L.lhm_ref_info(self.ref, info.ctypes.data_as(_u32p))
        return dict(zip(("ninner", "nleaf", "max_depth", "ntriangles"), map(int, info)))

    def ref_leaf_order(self):
        lp = np.empty(self.ntris, np.uint32); pf = np.empty(self.ntris, np.uint32)
        self.lib().lhm_ref_leaf_order(self.ref, lp.ctypes.data_as(_u32p), pf.ctypes.data_as(_u32p))
        return lp, pf

    def ref_bbox(self):
        b = np.empty(6); self.lib().lhm_ref_bbox(self.ref, b.ctypes.data_as(_dp)); return b[:3], b[3:]

    @classmethod
    def ref_off(cls):
        cls.lib().lhm_ref_use.argtypes = [C.c_void_p]; cls.lib().lhm_ref_use(None)

    def nodes(self):
        if self.nnodes == 0:
            return np.zeros((0, 16), np.float32)
        buf = (C.c_float * (16 * self.nnodes)).from_address(self.lib().lhm_nodes(self.h))
        return np.frombuffer(buf, np.float32).reshape(-1, 16).copy()

    def tri32(self):
        if self.ntris == 0:
            return np.zeros((0, 12), np.float32)
        buf = (C.c_float * (12 * self.ntris)).from_address(self.lib().lhm_tri32(self.h))
        return np.frombuffer(buf, np.float32).reshape(-1, 12).copy()

    def q4info(self):
        L = self.lib(); o = np.zeros(2, np.uint32)
        L.lhm_info4.argtypes = [C.c_void_p, _u32p]; L.lhm_info4(self.h, o.ctypes.data_as(_u32p))
        return int(o[0]), int(o[1])

    def grid(self):
        """(grid_lo[3], grid_step[3]) of the scene's 16-bit grid, as float64"""
        L = self.lib()
        L.lhm_grid.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        g = np.zeros(6, np.float32); L.lhm_grid(self.h, g.ctypes.data_as(C.POINTER(C.c_float)))
        return g[:3].astype(np.float64), g[3:].astype(np.float64)

    def q8info(self):
        L = self.lib(); o = np.zeros(2, np.uint32)
        L.lhm_infoq8.argtypes = [C.c_void_p, _u32p]; L.lhm_infoq8(self.h, o.ctypes.data_as(_u32p))
        return int(o[0]), int(o[1])

    def q8nodes(self):
        """lh_q8node_t records as uint32 [n, 32]: 8 x (x, y, z) words (lo | hi << 16), then 8 child references"""
        L = self.lib(); n, _ = self.q8info()
        L.lhm_q8nodes.restype = C.c_void_p; L.lhm_q8nodes.argtypes = [C.c_void_p]
        buf = (C.c_char * (128 * n)).from_address(L.lhm_q8nodes(self.h))
        return np.frombuffer(buf, np.uint32).reshape(-1, 32).copy()

    def q4nodes(self):
        L = self.lib(); n, _ = self.q4info()
        L.lhm_q4nodes.restype = C.c_void_p; L.lhm_q4nodes.argtypes = [C.c_void_p]
        buf = (C.c_uint16 * (32 * n)).from_address(L.lhm_q4nodes(self.h))
        return np.frombuffer(buf, np.uint16).reshape(-1, 32).copy()

    def trace(self, org, dr, anyhit=False, nthreads=4, qnodes=2):
        """qnodes: 0 fp32 2-wide nodes (the textbook variant), 2 16-bit grid 4-wide (the kernel's default),
        4 8-wide on the 16-bit grid (128-byte records: ray dumps over scenes larger than the Infinity Cache)"""
        qnodes = int(qnodes)
        assert qnodes in (0, 2, 4)
        org = np.ascontiguousarray(org, np.float64).reshape(-1, 3)
        dr = np.ascontiguousarray(dr, np.float64).reshape(-1, 3)
        n = org.shape[0]
        cnt = np.zeros(4, np.uint64)
        cp = cnt.ctypes.data_as(C.POINTER(C.c_uint64))
        if anyhit:
            occ = np.empty(n, np.uint8)
            self.lib().lhm_trace(self.h, n, org.ctypes.data_as(_dp), dr.ctypes.data_as(_dp), None, None, None, None,
                                 occ.ctypes.data_as(C.POINTER(C.c_uint8)), 1 | (qnodes << 1), cp, nthreads)
            return occ, dict(zip(("nodes", "tris", "exact", "rays"), map(int, cnt)))
        prim = np.empty(n, np.uint32); t = np.empty(n); u = np.empty(n); v = np.empty(n)
        self.lib().lhm_trace(self.h, n, org.ctypes.data_as(_dp), dr.ctypes.data_as(_dp), prim.ctypes.data_as(_u32p),
                             t.ctypes.data_as(_dp), u.ctypes.data_as(_dp), v.ctypes.data_as(_dp), None, qnodes << 1, cp, nthreads)
        return (prim, t, u, v), dict(zip(("nodes", "tris", "exact", "rays"), map(int, cnt)))

    def hostwalk(self, org, dr, use_ref=True):
        """the PRODUCT's one-ray host walk (lh_hostwalk.c) over this model's trees, ray by ray -> (prim, t, u, v).
        use_ref: with lucille's own tree (ref_build() first), as lh_accel_intersect1 calls it"""
        org = np.ascontiguousarray(org, np.float64).reshape(-1, 3); dr = np.ascontiguousarray(dr, np.float64).reshape(-1, 3)
        n = org.shape[0]
        prim = np.empty(n, np.uint32); t = np.empty(n); u = np.empty(n); v = np.empty(n)
        L = self.lib()
        L.lhm_hostwalk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, _dp, _dp, _u32p, _dp, _dp, _dp]
        L.lhm_hostwalk(self.h, self.ref if (use_ref and getattr(self, "ref", None)) else None, n, org.ctypes.data_as(_dp), dr.ctypes.data_as(_dp),
                       prim.ctypes.data_as(_u32p), t.ctypes.data_as(_dp), u.ctypes.data_as(_dp), v.ctypes.data_as(_dp))
        return prim, t, u, v

    def trace_diag(self, org, dr, qnodes=2):
        """closest-hit walk with PER-RAY counts -> ((prim, t, u, v), uint32 [n, 4]: node visits, leaf visits, triangle records
        through the fp32 filter, fp64 tests)"""
        org = np.ascontiguousarray(org, np.float64).reshape(-1, 3); n = org.shape[0]
        diag = np.zeros((n, 4), np.uint32)
        L = self.lib(); L.lhm_set_diag.argtypes = [C.c_void_p]
        L.lhm_set_diag(diag.ctypes.data)
        try:
            out, _ = self.trace(org, dr, nthreads=1, qnodes=qnodes)
        finally:
            L.lhm_set_diag(None)
        return out, diag


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def random_rays(rng, n, lo=-0.2, hi=1.2):
    org = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return org, d


def grid_mesh(nx, ny, z=0.0, size=1.0):
    """axis-aligned quad grid split 0-1-2 / 0-2-3 like polygon.c; shared vertices,
    so rays through vertices/edges/diagonals exercise the tolerance band"""
    xs = np.linspace(0.0, size, nx + 1); ys = np.linspace(0.0, size, ny + 1)
    P = np.array([[x, y, z] for y in ys for x in xs], np.float64)
    idx = []
    for j in range(ny):
        for i in range(nx):
            a = j * (nx + 1) + i; b = a + 1; c = a + nx + 2; d = a + nx + 1
            idx += [a, b, c, a, c, d]
    return P, np.array(idx, np.uint32)


def assert_hits_equal(got, exp, what=""):
    names = ("prim", "t", "u", "v")
    for k in range(4):
        g = np.asarray(got[k]); e = np.asarray(exp[k])
        if k == 0:
            g = g.view(np.uint32) if g.dtype == np.int32 else g
        bad = np.nonzero(g != e)[0]
        assert bad.size == 0, "%s: %s differs at %d rays, first %s: got %r expected %r" % (
            what, names[k], bad.size, bad[:5], g[bad[:5]], e[bad[:5]])


def random_beams(rng, n, spread, lo=-0.5, hi=1.5):
    """n beams: origin, 4 corner directions around an axis aimed into the unit cube
    (a small share straddles an octant, which ri_beam_set refuses with -1)"""
    org = rng.uniform(lo, hi, (n, 3))
    c = rng.uniform(0, 1, (n, 3)) - org
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    a = np.cross(c, rng.normal(size=(n, 3))); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = np.cross(c, a)
    d = np.stack([c - spread * a - spread * b, c + spread * a - spread * b, c + spread * a + spread * b,
                  c - spread * a + spread * b], 1)
    return org, d


def chain_scene(n, ratio=0.5 ** 0.5):
    """n triangles of geometrically shrinking size along a line: SAH peels them off one by one, so the
    tree is about as deep as it can get (4-wide depth 26+ from n = 120), and every triangle is isolated
    (a ray aimed at one of its vertices grazes box corners of the reference's tree)"""
    k = np.arange(n); s_ = ratio ** k
    base = np.stack([s_, s_ * 0.3, s_ * 0.1], 1)
    P = np.concatenate([base, base + np.stack([s_ * 0.4, 0 * s_, 0 * s_], 1), base + np.stack([0 * s_, s_ * 0.4, s_ * 0.1], 1)], 1).reshape(-1, 3)
    return P, np.arange(3 * n, dtype=np.uint32)


def vertex_aimed_rays(rng, P, idx, n, spread=1.2):
    """origins around the scene, targets EXACTLY on vertices / edge midpoints / centroids of triangles"""
    T = P[idx].reshape(-1, 3, 3)
    lo, hi = P.min(0), P.max(0)
    org = rng.uniform(lo - (spread - 1) * (hi - lo) - 0.1, hi + (spread - 1) * (hi - lo) + 0.1, (n, 3))
    pick = rng.integers(0, T.shape[0], n)
    tgt = T[pick, rng.integers(0, 3, n)].copy()
    tgt[1::3] = 0.5 * (T[pick[1::3], 0] + T[pick[1::3], 1])
    tgt[2::5] = T[pick[2::5]].mean(axis=1)
    return org, tgt - org


# ---- the beam-raster path (ri_bvh_intersect_beam): seeded cases ----------------------------------------------------
def raster_camera(eye, lookat, up, width, height, fov):
    """the testbed's camera (src/testbed/simplerender.cpp:37-70): -> corner, frame [du, dv, dw]"""
    eye = np.asarray(eye, np.float64); lookat = np.asarray(lookat, np.float64); up = np.asarray(up, np.float64)
    flen = 0.5 * width / np.tan(0.5 * (fov * np.pi / 180.0))
    dw = lookat - eye
    du = np.cross(dw, up); du = du / np.linalg.norm(du)
    dv = np.cross(dw, du); dv = dv / np.linalg.norm(dv)
    dw = dw / np.linalg.norm(dw)
    return flen * dw - 0.5 * (width * du + height * dv), np.stack([du, dv, dw])


def raster_beam_dirs(corner, frame, size, s, t):
    """corner directions of the beam over pixels [s, s + size) x [t, t + size) (simplerender.cpp:88-120)"""
    du, dv = frame[0], frame[1]
    return np.array([corner + (s + a) * du + (t + b) * dv for a, b in ((0, 0), (size, 0), (size, size), (0, size))])


def raster_case(seed, ntri, width, height, nbeams, eye=(0.0, 0.0, 0.0), fov=45.0, tri_size=0.6, inside=True):
    """A seeded scene in front of a camera looking down +z and `nbeams` square beams, each inside ONE quadrant of the
    window (ri_beam_set refuses beams whose corner directions differ in sign).  inside=False lets beams hang over the
    window's edge (undefined in the reference: it writes outside plane->t).
    -> dict(P, idx, eye, fov, width, height, frame, corner, org [n,3], dirs [n,4,3], corners [n,3])"""
    rng = np.random.default_rng(seed)
    c = np.stack([rng.uniform(-2, 2, ntri), rng.uniform(-2, 2, ntri), rng.uniform(4, 8, ntri)], 1)[:, None, :]
    P = (c + rng.uniform(-tri_size, tri_size, (ntri, 3, 3))).reshape(-1, 3)
    idx = np.arange(3 * ntri, dtype=np.uint32)
    eye = np.asarray(eye, np.float64)
    corner, frame = raster_camera(eye, eye + np.array([0.0, 0.0, 1.0]), [0.0, 1.0, 0.0], width, height, fov)
    dirs = []
    hw, hh = width // 2, height // 2
    for _ in range(nbeams):
        size = int(rng.integers(2, min(hw, hh) - 1))
        qx, qy = (int(v) for v in rng.integers(0, 2, 2))
        s = int(rng.integers(0, hw - size)) + qx * hw
        t = int(rng.integers(0, hh - size)) + qy * hh
        if not inside:                         # push the beam over the outer edge of its quadrant
            s = (width - size // 2) if qx else -(size // 2)
        dirs.append(raster_beam_dirs(corner, frame, size, s, t))
    n = nbeams
    return dict(P=P, idx=idx, eye=eye, fov=fov, width=width, height=height, frame=frame, corner=corner,
                org=np.repeat(eye[None], n, 0), dirs=np.array(dirs), corners=np.repeat(corner[None], n, 0))


# the cases tests/golden/beam_raster.npz holds the compiled reference's planes for (tests/golden/make_golden.py)
RASTER_GOLDEN_CASES = (
    dict(seed=7, ntri=200, width=64, height=64, nbeams=12, eye=(0.2, -0.1, 0.3)),      # eye off the origin: project_triangles' quirk shows
    dict(seed=11, ntri=40, width=32, height=32, nbeams=8),
    dict(seed=13, ntri=1500, width=48, height=48, nbeams=6, tri_size=0.25),
    dict(seed=17, ntri=300, width=128, height=128, nbeams=6, eye=(-0.4, 0.3, -0.2), fov=60.0),   # two columns per lane on the device
)
