"""ctypes binding of oracle/libft_oracle.so — TEST INFRASTRUCTURE ONLY.

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  It implements
the same small backend protocol as fraytracer_amd.api.Device, so one scene description can be
realised on both and the results compared.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libft_oracle.so")


class Counters(C.Structure):
    _fields_ = [("prim", C.c_uint64 * 8)] + [(k, C.c_uint64) for k in (
        "root_evals", "march_steps", "rays_primary", "rays_shadow", "rays_ext", "hits_primary", "hits_shadow",
        "smooth_children", "union_candidates", "flags", "union_tested")]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "prim"}
        d["prim"] = list(self.prim)
        return d


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    f3 = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int)
    sig = {
        "orc_last_error": (C.c_char_p, []), "orc_reset": (None, []), "orc_set_libm": (None, [C.c_int]),
        "orc_expf": (C.c_float, [C.c_float]), "orc_logf": (C.c_float, [C.c_float]), "orc_sqrtf": (C.c_float, [C.c_float]),
        "orc_divf": (C.c_float, [C.c_float, C.c_float]),
        "orc_mathf_min": (C.c_float, [C.c_float, C.c_float]), "orc_mathf_max": (C.c_float, [C.c_float, C.c_float]),
        "orc_expf_array": (None, [C.c_void_p, C.c_void_p, C.c_int64]), "orc_logf_array": (None, [C.c_void_p, C.c_void_p, C.c_int64]),
        "orc_sqrtf_array": (None, [C.c_void_p, C.c_void_p, C.c_int64]),
        "orc_form_sphere": (C.c_int, [f3, C.c_float]), "orc_form_capsule": (C.c_int, [f3, f3, C.c_float]),
        "orc_form_torus": (C.c_int, [f3, f3, C.c_float, C.c_float]), "orc_form_triangle": (C.c_int, [f3, f3, f3, C.c_float]),
        "orc_form_box": (C.c_int, [f3, f3]),
        "orc_form_union": (C.c_int, [ip, C.c_int]), "orc_form_subtract": (C.c_int, [C.c_int, C.c_int]),
        "orc_form_intersect": (C.c_int, [ip, C.c_int]), "orc_form_union_smooth": (C.c_int, [C.c_float, ip, C.c_int]),
        "orc_form_distance": (C.c_float, [C.c_int, f3]), "orc_form_boundary": (C.c_int, [C.c_int, f3]),
        "orc_form_grid_info": (C.c_int64, [C.c_int, f3, ip]),
        "orc_form_grid_dump": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "orc_material_solid": (C.c_int, [f3]), "orc_object_create": (C.c_int, [C.c_int, C.c_int]),
        "orc_object_union": (C.c_int, [ip, C.c_int]), "orc_object_subtract": (C.c_int, [C.c_int, C.c_int]),
        "orc_object_intersect": (C.c_int, [C.c_int, ip, C.c_int]), "orc_object_form": (C.c_int, [C.c_int]),
        "orc_object_color": (C.c_int, [C.c_int, f3, f3, f3]),
        "orc_light_directional": (C.c_int, [f3, f3]), "orc_light_point": (C.c_int, [f3, f3]),
        "orc_scene_create": (C.c_int, [C.c_int, f3, ip, C.c_int]),
        "orc_lens_create": (C.c_float, [C.c_float]), "orc_camera_lookat": (None, [f3, f3, f3, C.c_float, f3]),
        "orc_trace_rays": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(Counters)]),
        "orc_form_try_trace": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(Counters)]),
        "orc_object_try_trace": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(Counters)]),
        "orc_render": (C.c_int, [C.c_int, f3, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int, C.POINTER(Counters)]),
        "orc_render_strided": (C.c_int, [C.c_int, f3, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int, C.POINTER(Counters)]),
        "orc_render_ext": (C.c_int, [C.c_int, f3, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(Counters)]),
        "orc_material_glass": (C.c_int, [f3, C.c_float, C.c_float]),
        "orc_spectral_table": (C.c_int, [C.c_int, C.c_void_p]), "orc_glass_hash": (C.c_uint32, [C.c_uint32, C.c_uint32]),
        "orc_fresnel": (None, [C.c_float, C.c_float, f3, f3, C.c_void_p]),
        "orc_powf": (C.c_float, [C.c_float, C.c_float]),
        "orc_libm_checksums": (None, [C.c_int, C.c_float, C.c_uint32, C.c_int, C.c_void_p, C.c_int]),
        "orc_libm_array": (None, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
        "orc_tone_map": (C.c_float, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_uint32, C.c_int, C.c_void_p]),
        "orc_pixel_ray": (None, [f3, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, f3]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def _f3(v):
    return (C.c_float * 3)(*v)


def _hs(hs):
    return (C.c_int * len(hs))(*hs), len(hs)


class OracleError(RuntimeError):
    pass


def _ck(h):
    if h < 0:
        raise OracleError(lib.orc_last_error().decode())
    return h


class Oracle:
    """Backend protocol over the oracle's global arena (single-threaded construction)."""

    def reset(self): lib.orc_reset()
    def sphere(self, c, r): return _ck(lib.orc_form_sphere(_f3(c), r))
    def capsule(self, a, b, r): return _ck(lib.orc_form_capsule(_f3(a), _f3(b), r))
    def torus(self, c, n, R, r): return _ck(lib.orc_form_torus(_f3(c), _f3(n), R, r))
    def triangle(self, a, b, c, r): return _ck(lib.orc_form_triangle(_f3(a), _f3(b), _f3(c), r))
    def box(self, c, h): return _ck(lib.orc_form_box(_f3(c), _f3(h)))
    def form_union(self, hs): return _ck(lib.orc_form_union(*_hs(hs)))
    def form_subtract(self, a, b): return _ck(lib.orc_form_subtract(a, b))
    def form_intersect(self, hs): return _ck(lib.orc_form_intersect(*_hs(hs)))
    def form_union_smooth(self, k, hs): return _ck(lib.orc_form_union_smooth(k, *_hs(hs)))
    def material_solid(self, rgb): return _ck(lib.orc_material_solid(_f3(rgb)))
    def material_glass(self, tint, ior, dispersion): return _ck(lib.orc_material_glass(_f3(tint), ior, dispersion))
    def object_create(self, m, f): return _ck(lib.orc_object_create(m, f))
    def object_union(self, hs): return _ck(lib.orc_object_union(*_hs(hs)))
    def object_subtract(self, o, f): return _ck(lib.orc_object_subtract(o, f))
    def object_intersect(self, o, hs): return _ck(lib.orc_object_intersect(o, *_hs(hs)))
    def light_directional(self, d, c): return _ck(lib.orc_light_directional(_f3(d), _f3(c)))
    def light_point(self, p, c): return _ck(lib.orc_light_point(_f3(p), _f3(c)))
    def object_form(self, o): return _ck(lib.orc_object_form(o))

    def form_boundary(self, h):
        out = (C.c_float * 4)()
        _ck(lib.orc_form_boundary(h, out))
        return tuple(out)

    def form_distance(self, h, pts):
        pts = np.asarray(pts, np.float32).reshape(-1, 3)
        return np.array([lib.orc_form_distance(h, _f3(p)) for p in pts], np.float32)

    def object_color(self, o, p, n=(0, 0, 1)):
        out = (C.c_float * 3)()
        _ck(lib.orc_object_color(o, _f3(p), _f3(n), out))
        return tuple(out)

    def grid(self, form):
        info = (C.c_float * 9)()
        counts = (C.c_int * 3)()
        total = lib.orc_form_grid_info(form, info, counts)
        if total < 0:
            raise OracleError("form has no grid")
        nc = counts[0] * counts[1] * counts[2]
        cell_start = np.empty(nc + 1, np.uint32)
        centers = np.empty((nc, 3), np.float32)
        lower = np.empty(total, np.float32)
        item = np.empty(total, np.int32)
        _ck(lib.orc_form_grid_dump(form, *(a.ctypes.data_as(C.c_void_p) for a in (cell_start, centers, lower, item))))
        return {"aabbMin": np.array(info[0:3], np.float32), "cellSize": np.array(info[3:6], np.float32),
                "cellSizeInv": np.array(info[6:9], np.float32), "counts": tuple(counts),
                "cell_start": cell_start, "centers": centers, "lower": lower, "child": item}

    def scene(self, scene):
        from fraytracer_amd.api import realise
        memo = {}
        obj = realise(scene.Object, self, memo)
        lights = [realise(l, self, memo) for l in scene.Lights]
        hs, n = _hs(lights)
        return OracleScene(_ck(lib.orc_scene_create(obj, _f3(scene.BackgroundColor), hs, n)), obj)


class OracleScene:
    def __init__(self, handle, obj):
        self.handle = handle
        self.object = obj

    def render(self, epsilon, length, W, H, cam12, x0=0, x1=None, nthreads=None, xstep=1, spp=1, ao_samples=0, ao_radius=0.0,
               max_bounces=0, spectral=0):
        x1 = W if x1 is None else x1
        nthreads = nthreads or min(32, os.cpu_count() or 1)
        ncols = (x1 - x0 + xstep - 1) // xstep
        out = np.empty((ncols, H, 3), np.float32)
        cnt = Counters()
        cam = (C.c_float * 12)(*[float(v) for v in cam12])
        _ck(lib.orc_render_ext(self.handle, cam, W, H, x0, x1, xstep, epsilon, length, spp, ao_samples, ao_radius,
                               max_bounces, spectral, out.ctypes.data_as(C.c_void_p), nthreads, C.byref(cnt)))
        return out, cnt.as_dict()

    def form_try_trace(self, rays):
        """SdfForm.tryTrace scene.Object.Form: float32 [n, 10] = Ray (8), Distance, hit (int32 bits); misses are zeros"""
        return _try_trace(lib.orc_form_try_trace, self.handle, rays, 10)

    def object_try_trace(self, rays):
        """SdfObject.tryTrace scene.Object: float32 [n, 16] = Ray (8), Normal (3), Color (3), hit (int32 bits), 0"""
        return _try_trace(lib.orc_object_try_trace, self.handle, rays, 16)

    def trace_rays(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.empty((rays.shape[0], 3), np.float32)
        cnt = Counters()
        _ck(lib.orc_trace_rays(self.handle, rays.ctypes.data_as(C.c_void_p), rays.shape[0], out.ctypes.data_as(C.c_void_p), C.byref(cnt)))
        return out, cnt.as_dict()


def _try_trace(fn, handle, rays, width):
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    out = np.empty((rays.shape[0], width), np.float32)
    cnt = Counters()
    _ck(fn(handle, rays.ctypes.data_as(C.c_void_p), rays.shape[0], out.ctypes.data_as(C.c_void_p), C.byref(cnt)))
    return out, cnt.as_dict()


def spectral_table(nw):
    """EXTENSION: (nw, 4) float32 — rgb weight and Cauchy term of every wavelength bin"""
    out = np.empty((nw, 4), np.float32)
    _ck(lib.orc_spectral_table(nw, out.ctypes.data_as(C.c_void_p)))
    return out


def fresnel(n1, n2, N, D):
    """EXTENSION: {'reflectance', 'total', 'reflect', 'transmit'} of the oracle's repaired Light.fs:30-59"""
    out = np.empty(8, np.float32)
    lib.orc_fresnel(n1, n2, _f3(N), _f3(D), out.ctypes.data_as(C.c_void_p))
    return {"reflectance": float(out[0]), "total": bool(out[1]), "reflect": out[2:5].copy(), "transmit": out[5:8].copy()}


def glass_hash(seed, bounce):
    return int(lib.orc_glass_hash(seed & 0xFFFFFFFF, bounce))


def lens_create(fov):
    return lib.orc_lens_create(fov)


def camera_lookat(pos, look, up, near_plane_size):
    out = (C.c_float * 12)()
    lib.orc_camera_lookat(_f3(pos), _f3(look), _f3(up), near_plane_size, out)
    return np.array(out, np.float32)


def pixel_ray(cam12, W, H, x, y, epsilon, length):
    out = (C.c_float * 8)()
    cam = (C.c_float * 12)(*[float(v) for v in cam12])
    lib.orc_pixel_ray(cam, W, H, x, y, epsilon, length, out)
    return np.array(out, np.float32)


def expf(x):
    x = np.ascontiguousarray(x, np.float32); y = np.empty_like(x)
    lib.orc_expf_array(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.size); return y


def logf(x):
    x = np.ascontiguousarray(x, np.float32); y = np.empty_like(x)
    lib.orc_logf_array(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.size); return y


def sqrtf(x):
    x = np.ascontiguousarray(x, np.float32); y = np.empty_like(x)
    lib.orc_sqrtf_array(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.size); return y


def tone_map(image, gamma=2.2, seed=None, bmp_order=False):
    """Image.toColors gamma rng image (+ Image.toBitmap's buffer order): (uint8 [X, Y, 3] R,G,B or [Y, X, 3] B,G,R, max)"""
    img = np.ascontiguousarray(image, np.float32)
    X, Y = img.shape[0], img.shape[1]
    out = np.empty((Y, X, 3) if bmp_order else (X, Y, 3), np.uint8)
    mx = lib.orc_tone_map(img.ctypes.data_as(C.c_void_p), X, Y, gamma, 0 if seed is None else 1, 0 if seed is None else seed & 0xFFFFFFFF,
                          1 if bmp_order else 0, out.ctypes.data_as(C.c_void_p))
    return out, float(mx)


def powf(x, g):
    return lib.orc_powf(x, g)


def libm_checksums(op, y=0.0, lo_bits=0, n_chunks=256, nthreads=8):
    """this machine's expf (op 0) / logf (1) / powf(x, y) (2): one checksum per 2^24 consecutive float bit patterns (orc_libm_checksums)"""
    sums = np.zeros(n_chunks, np.uint64)
    lib.orc_libm_checksums(int(op), float(y), int(lo_bits), int(n_chunks), sums.ctypes.data_as(C.c_void_p), int(nthreads))
    return sums


def libm_array(op, x, y=None):
    x = np.ascontiguousarray(x, np.float32); out = np.empty_like(x)
    yy = np.ascontiguousarray(y if y is not None else np.zeros_like(x), np.float32)
    lib.orc_libm_array(int(op), x.ctypes.data_as(C.c_void_p), yy.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), x.size)
    return out
