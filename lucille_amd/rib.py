"""ctypes view of the RIB-subset reader / .hdr writer of liblucille_hip.so (lh_rib.c) and of
the `lsh_hip` driver: what `lsh scene.rib` needs either side of the ray-query path."""
import ctypes as C
import os

import numpy as np

from . import binding


class RibInfo(C.Structure):
    _fields_ = [("nmeshes", C.c_uint32), ("ntriangles", C.c_uint64), ("camera", binding.Camera),
                ("perspective", C.c_int), ("fov", C.c_float), ("pixel_samples", C.c_int * 2),
                ("gather_nsamples", C.c_int), ("accel_method", C.c_int), ("nthreads", C.c_int),
                ("world_complete", C.c_int), ("nskipped", C.c_uint32), ("nrequests", C.c_uint32),
                ("nunknown", C.c_uint32), ("display_name", C.c_char * 1024), ("display_type", C.c_char * 64)]


_ready = False


def api():
    global _ready
    L = binding.lib()
    if not _ready:
        dpp, upp = C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_uint32))
        L.lh_rib_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.lh_rib_free.argtypes = [C.c_void_p]; L.lh_rib_free.restype = None
        L.lh_rib_last_error.restype = C.c_char_p
        L.lh_rib_messages.argtypes = [C.c_void_p]; L.lh_rib_messages.restype = C.c_char_p
        L.lh_rib_info.argtypes = [C.c_void_p, C.POINTER(RibInfo)]
        L.lh_rib_mesh.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), dpp, C.POINTER(C.c_uint32), upp, dpp,
                                  C.POINTER(C.c_int)]
        L.lh_accel_add_rib_scene.argtypes = [C.c_void_p, C.c_void_p]
        L.lh_hdr_write.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        L.lh_render_ao_frame_host.argtypes = [C.c_void_p, C.POINTER(binding.Camera), C.c_int, C.c_int, C.c_uint64, C.c_int,
                                              C.c_void_p, C.POINTER(binding.TileStats)]
        _ready = True
    return L


class RibScene:
    """lh_rib_load: geoms (world space, RIB order) + camera + options of one RIB file"""

    def __init__(self, path):
        L = api()
        self.h = C.c_void_p()
        if L.lh_rib_load(os.fsencode(path), C.byref(self.h)) != 0:
            raise binding.LucilleHipError(L.lh_rib_last_error().decode())
        self.info = RibInfo()
        L.lh_rib_info(self.h, C.byref(self.info))
        self.messages = L.lh_rib_messages(self.h).decode()

    def close(self):
        if self.h:
            api().lh_rib_free(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def camera(self):
        return self.info.camera

    def mesh(self, m):
        """-> dict(positions [n,4], indices [k], normals [n,4] | None, two_side) as copies"""
        L = api()
        npos, nidx, two = C.c_uint32(), C.c_uint32(), C.c_int()
        pp, nn, ip = C.POINTER(C.c_double)(), C.POINTER(C.c_double)(), C.POINTER(C.c_uint32)()
        if L.lh_rib_mesh(self.h, m, C.byref(npos), C.byref(pp), C.byref(nidx), C.byref(ip), C.byref(nn), C.byref(two)) != 0:
            raise binding.LucilleHipError(L.lh_rib_last_error().decode())
        P = np.ctypeslib.as_array(pp, (npos.value, 4)).copy() if npos.value else np.zeros((0, 4))
        I = np.ctypeslib.as_array(ip, (nidx.value,)).copy() if nidx.value else np.zeros(0, np.uint32)
        N = np.ctypeslib.as_array(nn, (npos.value, 4)).copy() if (npos.value and bool(nn)) else None
        return {"positions": P, "indices": I, "normals": N, "two_side": two.value}

    def meshes(self):
        return [self.mesh(m) for m in range(self.info.nmeshes)]

    def add_to(self, acc):
        """meshes (+ normals) into a HipAccel, in order (lh_accel_add_rib_scene); caller commits"""
        if api().lh_accel_add_rib_scene(acc.h, self.h) != 0:
            raise binding.LucilleHipError(binding.lib().lh_last_error().decode())


def hdr_write(path, rgb):
    """rgb [H,W,3] float32, top row first -> Radiance .hdr as the reference's file driver writes it"""
    a = np.ascontiguousarray(rgb, np.float32)
    if api().lh_hdr_write(os.fsencode(path), a.shape[1], a.shape[0], a.ctypes.data) != 0:
        raise binding.LucilleHipError(api().lh_rib_last_error().decode())


def hdr_read(path):
    """decode a run-length coded Radiance .hdr (the standard format) -> [H,W,3] float32; test helper
    and a way to look at lsh_hip's output without other tools"""
    raw = open(path, "rb").read()
    end = raw.index(b"\n\n") + 2
    nl = raw.index(b"\n", end)
    tok = raw[end:nl].split()
    H, W = int(tok[1]), int(tok[3])
    p = nl + 1
    out = np.zeros((H, W, 4), np.uint8)
    for y in range(H):
        if W < 8 or W > 0x7fff or raw[p] != 2 or raw[p + 1] != 2:
            out[y] = np.frombuffer(raw, np.uint8, 4 * W, p).reshape(W, 4); p += 4 * W
            continue
        p += 4
        for c in range(4):
            x = 0
            while x < W:
                n = raw[p]; p += 1
                if n > 128:
                    out[y, x:x + n - 128, c] = raw[p]; p += 1; x += n - 128
                else:
                    out[y, x:x + n, c] = np.frombuffer(raw, np.uint8, n, p); p += n; x += n
    e = out[..., 3].astype(np.int32)
    f = np.where(e > 0, np.ldexp(1.0, e - 136), 0.0)
    return (out[..., :3] * f[..., None]).astype(np.float32)


def lsh_hip_path():
    return os.path.join(binding.CSRC, "lsh_hip")
