"""Drive the COMPILED REFERENCE's RenderMan C API (Ri*V entry points in
oracle/_ref/liblucille_ref*.so) from a RIB file -- test infrastructure that stands
in for `lsh` (which needs flex/bison, absent here).  Verbs: the subset the
reference's example scenes use.

    render_rib(path, width, height, gather_nsamples, pixel_samples, accel_method) ->
        dict(image, geoms, camera, records)

Each render runs in the calling process and leaves the renderer finished (the
reference frees its scene in render_frame_cleanup): call it from a fresh
subprocess (render_rib_subprocess) when more than one render is needed.
"""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def tokenize(text):
    text = re.sub(r"#[^\n]*", "", text)
    for m in re.finditer(r'"([^"]*)"|(\[)|(\])|([^\s\[\]"]+)', text):
        if m.group(1) is not None:
            yield ("s", m.group(1))
        elif m.group(2):
            yield ("[", None)
        elif m.group(3):
            yield ("]", None)
        else:
            tok = m.group(4)
            try:
                yield ("n", float(tok))
            except ValueError:
                yield ("w", tok)


def _find_archive(name, base, root):
    """Option "searchpath" "archive" of the example scenes lists sub-directories of the scene
    directory: look next to the including file, then anywhere under the top-level RIB's directory"""
    cand = os.path.join(base, name)
    if os.path.exists(cand):
        return cand
    for dp, _, files in os.walk(root):
        if name in files:
            return os.path.join(dp, name)
    raise FileNotFoundError(name)


def parse(path, _depth=0, _root=None):
    """-> list of (verb, args) with arrays as python lists; ReadArchive is inlined"""
    toks = list(tokenize(open(path).read()))
    out, i = [], 0
    base = os.path.dirname(path)
    root = _root or base
    while i < len(toks):
        kind, val = toks[i]
        assert kind == "w", "expected a RIB verb, got %r" % (toks[i],)
        verb, args = val, []
        i += 1
        while i < len(toks) and toks[i][0] != "w":
            if toks[i][0] == "[":
                arr = []; i += 1
                while toks[i][0] != "]":
                    arr.append(toks[i][1]); i += 1
                i += 1
                args.append(arr)
            else:
                args.append(toks[i][1]); i += 1
        if verb == "ReadArchive":
            out += parse(_find_archive(args[0], base, root), _depth + 1, root)
        else:
            out.append((verb, args))
    return out


def _farr(v):
    return (C.c_float * len(v))(*[float(x) for x in v])


def _iarr(v):
    return (C.c_int * len(v))(*[int(x) for x in v])


def _params(args):
    """token/value pairs -> (n, tokens[], params[]) with floats as RtFloat arrays"""
    toks, vals, keep = [], [], []
    for k in range(0, len(args) - 1, 2):
        toks.append(args[k].encode())
        v = args[k + 1]
        if isinstance(v, list) and v and isinstance(v[0], str):
            s = (C.c_char_p * len(v))(*[x.encode() for x in v]); keep.append(s); vals.append(C.cast(s, C.c_void_p))
        elif isinstance(v, str):
            s = (C.c_char_p * 1)(v.encode()); keep.append(s); vals.append(C.cast(s, C.c_void_p))
        else:
            a = _farr(v if isinstance(v, list) else [v]); keep.append(a); vals.append(C.cast(a, C.c_void_p))
    n = len(toks)
    T = (C.c_char_p * max(n, 1))(*toks); V = (C.c_void_p * max(n, 1))(*vals)
    return n, T, V, keep


def render_rib(path, width, height, gather_nsamples, pixel_samples=1, accel_method=1, nthreads=1,
               lib="liblucille_ref.so", record=True):
    return render_verbs(parse(path), width, height, gather_nsamples, pixel_samples, accel_method, nthreads, lib, record)


def scene_verbs(geoms, world_to_camera, fov=45.0, rh=True):
    """RIB verbs for a scene given as triangle meshes (positions [n,3], indices) + camera:
    how tests feed the reference's renderer without a RIB file (no /root/reference needed)"""
    v = [("Display", ["out.hdr", "file", "rgb"]), ("Projection", ["perspective", "fov", [fov]])]
    if rh:
        v.append(("Orientation", ["rh"]))
    v.append(("ConcatTransform", [[float(x) for x in np.asarray(world_to_camera).reshape(16)]]))
    v.append(("WorldBegin", []))
    for P, I in geoms:
        I = np.asarray(I).reshape(-1)
        v.append(("PointsPolygons", [[3] * (len(I) // 3), [int(i) for i in I], "P",
                                     [float(x) for x in np.asarray(P, np.float64).reshape(-1)]]))
    v.append(("WorldEnd", []))
    return v


def render_verbs(verbs, width, height, gather_nsamples, pixel_samples=1, accel_method=1, nthreads=1,
                 lib="liblucille_ref.so", record=True):
    L = C.CDLL(os.path.join(HERE, "_ref", lib))
    L.lref_init(); L.lref_capture_display()
    L.RiFormat.argtypes = [C.c_int, C.c_int, C.c_float]
    L.RiPixelSamples.argtypes = [C.c_float, C.c_float]
    L.RiShutter.argtypes = [C.c_float, C.c_float]
    L.RiBegin.argtypes = [C.c_char_p]
    L.RiOrientation.argtypes = [C.c_char_p]
    Mat = (C.c_float * 4) * 4
    cwd = os.getcwd(); tmp = tempfile.mkdtemp(); os.chdir(tmp)
    try:
        L.RiBegin(None)
        for verb, a in verbs:
            if verb == "Display":
                n, T, V, keep = _params(a[3:])
                L.RiDisplayV(b"capture.hdr", b"file", b"rgb", n, T, V)   # .hdr name keeps the type "file" (display.c:148-185) = our capture driver
            elif verb == "Format":
                pass                                   # overridden below (lsh applies CLI overrides the same way)
            elif verb == "PixelSamples":
                pass
            elif verb == "Shutter":
                L.RiShutter(a[0], a[1])
            elif verb == "Projection":
                n, T, V, keep = _params(a[1:])
                L.RiProjectionV(a[0].encode(), n, T, V)
            elif verb == "Orientation":
                L.RiOrientation(a[0].encode())
            elif verb in ("ConcatTransform", "Transform"):
                m = Mat(*[(C.c_float * 4)(*a[0][4 * r:4 * r + 4]) for r in range(4)])
                getattr(L, "Ri" + verb)(m)
            elif verb == "WorldBegin":
                L.RiFormat(width, height, 1.0)
                L.RiPixelSamples(float(pixel_samples), float(pixel_samples))
                L.RiWorldBegin()
                L.lref_set_options(accel_method, nthreads, gather_nsamples)
                if record:
                    L.lref_record_start()
            elif verb == "WorldEnd":
                L.RiWorldEnd()
            elif verb in ("AttributeBegin", "AttributeEnd", "TransformBegin", "TransformEnd"):
                getattr(L, "Ri" + verb)()
            elif verb == "PointsPolygons":
                nverts, verts = _iarr(a[0]), _iarr(a[1])
                n, T, V, keep = _params(a[2:])
                L.RiPointsPolygonsV(len(a[0]), nverts, verts, n, T, V)
            elif verb == "Polygon":
                n, T, V, keep = _params(a)
                p_arr = [a[k + 1] for k in range(0, len(a) - 1, 2) if a[k] == "P"][0]
                L.RiPolygonV(len(p_arr) // 3, n, T, V)
            elif verb == "Identity":
                L.RiIdentity()
            elif verb in ("Translate", "Scale"):
                f = getattr(L, "Ri" + verb); f.argtypes = [C.c_float] * 3
                f(*[float(x) for x in (a[0] if isinstance(a[0], list) else a[:3])])
            elif verb == "Rotate":
                L.RiRotate.argtypes = [C.c_float] * 4
                L.RiRotate(*[float(x) for x in (a[0] if isinstance(a[0], list) else a[:4])])
            elif verb == "Sides":
                L.RiSides.argtypes = [C.c_int]
                L.RiSides(int(a[0]))
            elif verb in ("Surface", "ShadingInterpolation", "Atmosphere", "Imager", "ShadingRate", "Option",
                          "Attribute", "Color", "Opacity", "Declare", "FrameBegin", "FrameEnd",
                          "Exposure", "Quantize", "Clipping", "ScreenWindow", "LightSource", "version"):
                pass                                   # no effect on the ray-query path
            else:
                raise ValueError("RIB verb not handled by the test driver: " + verb)
    finally:
        os.chdir(cwd)
    out = {}
    L.lref_record_stop.restype = C.c_size_t; L.lref_record_size.restype = C.c_size_t
    L.lref_record_copy.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
    nrec = L.lref_record_stop() if record else 0
    rec_dt = np.dtype([("org", "f8", 3), ("dir", "f8", 3), ("t", "f8"), ("u", "f8"), ("v", "f8"),
                       ("hit", "u4"), ("geom", "u4"), ("index", "u4"), ("pad", "u4")])
    assert L.lref_record_size() == rec_dt.itemsize
    rec = np.zeros(nrec, rec_dt)
    if nrec:
        L.lref_record_copy(rec.ctypes.data, 0, nrec)
    out["records"] = rec
    w, h = C.c_int(), C.c_int()
    L.lref_image_size(C.byref(w), C.byref(h))
    img = np.zeros((h.value, w.value, 3), np.float32)
    L.lref_image_copy.argtypes = [C.c_void_p]; L.lref_image_copy(img.ctypes.data)
    out["image"] = img
    cam = np.zeros(20); L.lref_camera_get.argtypes = [C.c_void_p]; L.lref_camera_get(cam.ctypes.data)
    out["camera"] = cam
    out["ortho"] = int(L.lref_camera_is_ortho())
    geoms = []
    L.lref_scene_geom_sizes.argtypes = [C.c_uint32] + [C.POINTER(C.c_uint32)] * 4
    L.lref_scene_geom_copy.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    for g in range(L.lref_scene_ngeoms()):
        npos, nidx, hn, ts = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        L.lref_scene_geom_sizes(g, C.byref(npos), C.byref(nidx), C.byref(hn), C.byref(ts))
        P = np.zeros((npos.value, 3)); I = np.zeros(nidx.value, np.uint32); N = np.zeros((npos.value, 3)) if hn.value else None
        L.lref_scene_geom_copy(g, P.ctypes.data, I.ctypes.data, N.ctypes.data if N is not None else None)
        geoms.append({"positions": P, "indices": I, "normals": N, "two_side": int(ts.value)})
    out["geoms"] = geoms
    return out


def render_scene_subprocess(scene_npz, outfile, env=None, **kw):
    """like render_rib_subprocess for a scene stored as .npz {ngeoms, pos%d, idx%d, w2c, fov};
    env: extra environment variables of the child (RI_HIP_RENDER ...)"""
    code = ("import sys; sys.path.insert(0, %r); import numpy as np; from oracle import ref_rib as r; "
            "g = np.load(%r); "
            "verbs = r.scene_verbs([(g['pos%%d' %% i], g['idx%%d' %% i]) for i in range(int(g['ngeoms']))], g['w2c'], float(g['fov'])); "
            "o = r.render_verbs(verbs, **%r); "
            "np.savez(%r, image=o['image'], camera=o['camera'], records=o['records'])") % (
                os.path.dirname(HERE), scene_npz, kw, outfile)
    e = dict(os.environ)
    e.update(env or {})
    subprocess.check_call([sys.executable, "-c", code], stdout=subprocess.DEVNULL, env=e)
    return np.load(outfile)


def render_rib_subprocess(path, outfile, **kw):
    """fresh process per render; result saved as .npz at outfile"""
    code = ("import sys; sys.path.insert(0, %r); import numpy as np; from oracle import ref_rib as r; "
            "o = r.render_rib(%r, **%r); "
            "d = {'image': o['image'], 'camera': o['camera'], 'ortho': o['ortho'], 'records': o['records'], 'ngeoms': len(o['geoms'])}; "
            "[d.update({'pos%%d' %% i: g['positions'], 'idx%%d' %% i: g['indices'], 'two_side%%d' %% i: g['two_side']}) for i, g in enumerate(o['geoms'])]; "
            "[d.update({'nrm%%d' %% i: g['normals']}) for i, g in enumerate(o['geoms']) if g['normals'] is not None]; "
            "np.savez(%r, **d)") % (os.path.dirname(HERE), path, kw, outfile)
    subprocess.check_call([sys.executable, "-c", code], stdout=subprocess.DEVNULL)
    return np.load(outfile)
