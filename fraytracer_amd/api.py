"""Host-side mirror of FrayTracer's scene-composition API (the reference is F#; no .NET toolchain
exists in the build image, so the host layer above the C ABI is written in Python — INTEGRATION.md
shows the F# binding a maintainer would add).

Names, argument order and behaviour follow the reference modules:

    SdfForm.Primitive.sphere/capsule/torus/triangle   src/FrayTracer/SdfForm.fs:117-268
    SdfForm.union/subtract/intersect/unionSmooth      src/FrayTracer/SdfForm.fs:14-91
    SdfMaterial.createSolid                           src/FrayTracer/SdfMaterial.fs:4-7
    SdfObject.create/union/subtract/intersect         src/FrayTracer/SdfObject.fs:6-64
    SdfLight.directional/point                        src/FrayTracer/SdfLight.fs:6-42
    SdfScene (record) / SdfScene.trace                src/FrayTracer/Types.fs:74-79, SdfScene.fs:7-28
    Lens.create / Camera.lookAt                       src/FrayTracer/Camera.fs:11-42
    ImageSize / Image.render                          src/FrayTracer/Image.fs:8-35

Scene values are immutable descriptions (the reference's goal "immutable scenes", README.md:8).
They are realised through libfraytracer_hip's constructor twins the first time they are rendered
on a device.  The same descriptions can be realised on any object offering the small backend
protocol below (tests realise them on the CPU oracle to compare results).
"""
import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import lib, check, FrayTracerError


def _v3(v):
    x, y, z = (float(np.float32(c)) for c in v)
    return (x, y, z)


class FColor(tuple):
    """FColor.fs:8-33 — an RGB triple of float32."""

    @staticmethod
    def ofRGB(r, g, b):
        return FColor(_v3((r, g, b)))


# --------------------------------------------------------------------------------------------------
# descriptions
# --------------------------------------------------------------------------------------------------
class _Desc:
    __slots__ = ("kind", "args", "kids", "__weakref__")

    def __init__(self, kind, args=(), kids=()):
        object.__setattr__(self, "kind", kind)
        object.__setattr__(self, "args", tuple(args))
        object.__setattr__(self, "kids", tuple(kids))

    def __setattr__(self, *_):
        raise AttributeError("scene descriptions are immutable")

    def __repr__(self):
        return f"<{type(self).__name__} {self.kind} kids={len(self.kids)}>"


class Form(_Desc):
    pass


class Material(_Desc):
    pass


class Object(_Desc):
    pass


class Light(_Desc):
    pass


class SdfForm:
    class Primitive:
        @staticmethod
        def sphere(Center, Radius):
            return Form("sphere", (_v3(Center), float(np.float32(Radius))))

        @staticmethod
        def capsule(From, To, Radius):
            return Form("capsule", (_v3(From), _v3(To), float(np.float32(Radius))))

        @staticmethod
        def torus(Center, Normal, MajorRadius, MinorRadius):
            return Form("torus", (_v3(Center), _v3(Normal), float(np.float32(MajorRadius)), float(np.float32(MinorRadius))))

        @staticmethod
        def triangle(V1, V2, V3, Radius):
            return Form("triangle", (_v3(V1), _v3(V2), _v3(V3), float(np.float32(Radius))))

        @staticmethod
        def box(Center, HalfExtent):
            """EXTENSION — not in the reference (BASELINE.json config 2 asks for boxes)."""
            return Form("box", (_v3(Center), _v3(HalfExtent)))

    @staticmethod
    def union(forms):
        forms = list(forms)
        if not forms:
            raise ValueError("No SdfObjects given.")            # SdfForm.fs:16
        return forms[0] if len(forms) == 1 else Form("union", (), forms)

    @staticmethod
    def subtract(a, b):
        return Form("subtract", (), (a, b))

    @staticmethod
    def intersect(forms):
        forms = list(forms)
        if not forms:
            raise ValueError("No SdfObjects given.")            # SdfForm.fs:53
        return forms[0] if len(forms) == 1 else Form("intersect", (), forms)

    @staticmethod
    def unionSmooth(strength, forms):
        forms = list(forms)
        if not forms:
            raise ValueError("blub")                            # SdfForm.fs:71
        return forms[0] if len(forms) == 1 else Form("unionSmooth", (float(np.float32(strength)),), forms)

    @staticmethod
    def tryTrace(form, rays, device=None):
        """SdfForm.tryTrace (SdfForm.fs:93-104) over a ray buffer on the GPU, see form_try_trace"""
        return form_try_trace(form, rays, device)

    @staticmethod
    def distance(form, points, device=None):
        """sdf.Distance at points [n, 3] on the GPU -> float32 [n]"""
        return _form_scene(form, device).eval_distance(points)[0]

    @staticmethod
    def tryDistance(form, points, device=None):
        """SdfForm.tryDistance (SdfForm.fs:7-12): sdf.Distance where SdfBoundary.isInside (DistanceSquared(Center, p) <
        Radius * Radius, SdfBoundary.fs:56), NaN standing for ValueNone elsewhere"""
        F = np.float32
        pts = np.ascontiguousarray(points, dtype=F).reshape(-1, 3)
        ds = _form_scene(form, device)
        b = ds.boundary()
        d = pts - np.asarray(b[0:3], F)
        inside = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]) < F(b[3]) * F(b[3])
        out = ds.eval_distance(pts)[0]
        out[~inside] = np.nan
        return out

    @staticmethod
    def normalFromRay(form, rays, device=None):
        """SdfForm.normalFromRay (SdfForm.fs:106-115) at the position of each ray [n, 8] -> float32 [n, 3]: forward
        differences with h = Epsilon / 8 at Ray.get(-Epsilon), evaluated on the GPU"""
        F = np.float32
        r = np.ascontiguousarray(rays, dtype=F).reshape(-1, 8)
        p = r[:, 0:3] + r[:, 3:6] * (-r[:, 7:8])
        h = r[:, 7] * F(0.125)
        probes = np.repeat(p[:, None, :], 4, axis=1)
        for k in range(3):
            probes[:, k, k] = p[:, k] + h
        d = _form_scene(form, device).eval_distance(probes.reshape(-1, 3))[0].reshape(-1, 4)
        g = d[:, 0:3] - d[:, 3:4]
        with np.errstate(all="ignore"):
            return g / np.sqrt((g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]) + g[:, 2] * g[:, 2])[:, None]


_trace_cache = []          # most recent first: (description, light-less scene around it); a few entries


def _cached_scene(key, make):
    for i, (k, scene) in enumerate(_trace_cache):
        if k is key:
            if i:
                _trace_cache.insert(0, _trace_cache.pop(i))
            return scene
    scene = make()
    _trace_cache.insert(0, (key, scene))
    del _trace_cache[8:]
    return scene


def _object_scene(object, device):
    """device scene holding just `object` (no lights): what the tryTrace entries need"""
    scene = _cached_scene(object, lambda: SdfScene(object, (0.0, 0.0, 0.0), []))
    dev = device if device is not None else Device.default(0)
    return dev.scene(scene)


def _form_scene(form, device):
    scene = _cached_scene(form, lambda: SdfScene(SdfObject.create(SdfMaterial.createSolid((0.0, 0.0, 0.0)), form), (0.0, 0.0, 0.0), []))
    dev = device if device is not None else Device.default(0)
    return dev.scene(scene)


def form_try_trace(form, rays, device=None):
    """SdfForm.tryTrace sdf ray (SdfForm.fs:93-104) over rays [n, 8] on the GPU -> float32 [n, 10]
    (Ray at the hit, Distance, hit flag as int32 bits); a miss (ValueNone) is a row of zeros."""
    return _form_scene(form, device).form_try_trace(rays)[0]


class SdfMaterial:
    @staticmethod
    def createSolid(color):
        return Material("solid", (_v3(color),))

    @staticmethod
    def createGlass(tint, ior, dispersion=0.0):
        """EXTENSION (not in the reference): refracting material, Cauchy dispersion in um^2."""
        return Material("glass", (_v3(tint), float(np.float32(ior)), float(np.float32(dispersion))))


class SdfObject:
    @staticmethod
    def create(material, form):
        return Object("create", (), (material, form))

    @staticmethod
    def union(objects):
        objects = list(objects)
        if not objects:
            raise ValueError("No SdfObjects given.")            # SdfObject.fs:14
        return objects[0] if len(objects) == 1 else Object("union", (), objects)

    @staticmethod
    def subtract(object, form):
        return Object("subtract", (), (object, form))

    @staticmethod
    def intersect(object, forms):
        return Object("intersect", (), (object,) + tuple(forms))

    @staticmethod
    def tryTrace(object, rays, device=None):
        """SdfObject.tryTrace object ray (SdfObject.fs:66-78) over rays [n, 8] on the GPU -> float32 [n, 16]
        (Ray pulled back by epsilon, Normal, Color, hit flag as int32 bits, 0); a miss (ValueNone) is a row of zeros."""
        return _object_scene(object, device).object_try_trace(rays)[0]


class SdfLight:
    @staticmethod
    def directional(direction, color):
        return Light("directional", (_v3(direction), _v3(color)))

    @staticmethod
    def point(position, color):
        return Light("point", (_v3(position), _v3(color)))


class SdfScene:
    """Types.fs:74-79 record {Object; BackgroundColor; Lights}."""

    def __init__(self, Object, BackgroundColor, Lights=()):
        self.Object = Object
        self.BackgroundColor = _v3(BackgroundColor)
        self.Lights = tuple(Lights)
        self._realised = {}
        self._lock = threading.Lock()

    @staticmethod
    def trace(scene, device=None):
        """SdfScene.trace scene : Ray -> FColor (SdfScene.fs:7-8), evaluated on the GPU."""
        return SceneTrace(scene, device)


def realise(node, backend, memo=None):
    """Build `node` on a backend (libfraytracer_hip context or the test oracle); returns its handle."""
    memo = {} if memo is None else memo
    key = id(node)
    if key in memo:
        return memo[key]
    k = node.kind
    kids = [realise(c, backend, memo) for c in node.kids]
    if isinstance(node, Form):
        if k == "sphere": h = backend.sphere(*node.args)
        elif k == "capsule": h = backend.capsule(*node.args)
        elif k == "torus": h = backend.torus(*node.args)
        elif k == "triangle": h = backend.triangle(*node.args)
        elif k == "box": h = backend.box(*node.args)
        elif k == "union": h = backend.form_union(kids)
        elif k == "subtract": h = backend.form_subtract(kids[0], kids[1])
        elif k == "intersect": h = backend.form_intersect(kids)
        elif k == "unionSmooth": h = backend.form_union_smooth(node.args[0], kids)
        else: raise ValueError(k)
    elif isinstance(node, Material):
        h = backend.material_solid(node.args[0]) if k == "solid" else backend.material_glass(*node.args)
    elif isinstance(node, Object):
        if k == "create": h = backend.object_create(kids[0], kids[1])
        elif k == "union": h = backend.object_union(kids)
        elif k == "subtract": h = backend.object_subtract(kids[0], kids[1])
        elif k == "intersect": h = backend.object_intersect(kids[0], kids[1:])
        else: raise ValueError(k)
    elif isinstance(node, Light):
        h = backend.light_directional(*node.args) if k == "directional" else backend.light_point(*node.args)
    else:
        raise TypeError(type(node))
    memo[key] = h
    return h


# --------------------------------------------------------------------------------------------------
# libfraytracer_hip backend
# --------------------------------------------------------------------------------------------------
def _f3(v):
    return (C.c_float * 3)(*v)


def _handles(hs):
    return (C.c_int32 * len(hs))(*hs), len(hs)


def _vec(v):
    return _lib.Vec3(*v)


_GLIBC_PROBES = ((0x4202422F, 0x56FC9F1C, 0x56FC9F1B), (0xC27C65D9, 0x11FA2993, 0x11FA2992))   # expf input bits, FMA build's result, SSE2 build's result


def glibc_build_of_this_host():
    """FT_MATH_GLIBC_FMA (1) or FT_MATH_GLIBC_SSE2 (2): which build of expf / logf / powf the C runtime of this machine resolves to — the
    value to give Device.set_option("math", ...) for results that equal the reference's CPU path on this host.  Decided by asking the running
    libm: the two builds of glibc 2.35's expf differ on exactly two of the 2^32 inputs (found by comparing the restatements of
    csrc/ft_libm.h exhaustively; logf never differs), so expf of those two tells which one the ifunc resolver picked — whatever
    /proc/cpuinfo, GLIBC_TUNABLES or the OS's XSAVE state made it pick.  Raises if the answers match neither build (another libm: then
    neither glibc mode restates this host's arithmetic)."""
    import struct
    libm = C.CDLL("libm.so.6")
    libm.expf.restype, libm.expf.argtypes = C.c_float, [C.c_float]
    got = [struct.unpack("<I", struct.pack("<f", libm.expf(struct.unpack("<f", struct.pack("<I", x))[0])))[0] for x, _, _ in _GLIBC_PROBES]
    if got == [f for _, f, _ in _GLIBC_PROBES]:
        return _lib.FT_MATH_GLIBC_FMA
    if got == [s_ for _, _, s_ in _GLIBC_PROBES]:
        return _lib.FT_MATH_GLIBC_SSE2
    raise RuntimeError("this host's expf is neither build of glibc 2.35's (probe results %s): FT_OPT_MATH's glibc modes do not restate it" % [hex(g) for g in got])


class Device:
    """One ft_ctx: a GPU (index >= 0) or a host-only context (index -1: construction and
    introspection only — rendering raises, there is no CPU fallback)."""

    _default = {}
    _default_lock = threading.RLock()          # re-entrant: default() constructs a Device, whose __init__ takes the lock for its serial
    _serial = 0

    def __init__(self, index=0):
        p = C.c_void_p()
        check(lib.ft_ctx_create(int(index), C.byref(p)))
        self._ctx = p
        self.index = int(index)
        self._scenes = []                      # DeviceScenes created on this context (closed with it)
        with Device._default_lock:
            Device._serial += 1
            self.serial = Device._serial       # cache key for SdfScene._realised (id() values get reused)

    @classmethod
    def default(cls, index=0):
        with cls._default_lock:
            if index not in cls._default:
                cls._default[index] = Device(index)
            return cls._default[index]

    def close(self):
        if self._ctx:
            for s in self._scenes:             # scenes hold device memory of this context: release them first
                s.close()
            self._scenes = []
            lib.ft_ctx_destroy(self._ctx)
            self._ctx = None

    def host_register(self, array):
        """page-lock a numpy array that is reused as render(..., out=array) destination (ft_host_register)"""
        check(lib.ft_host_register(self._ctx, array.ctypes.data_as(C.c_void_p), array.nbytes))

    def host_unregister(self, array):
        check(lib.ft_host_unregister(self._ctx, array.ctypes.data_as(C.c_void_p)))

    def set_stream(self, hip_stream):
        check(lib.ft_ctx_set_stream(self._ctx, C.c_void_p(hip_stream)))

    OPTIONS = {"refill_min": _lib.FT_OPT_REFILL_MIN, "max_blocks_per_cu": _lib.FT_OPT_MAX_BLOCKS_PER_CU,
               "host_chunks": _lib.FT_OPT_HOST_CHUNKS, "host_pin": _lib.FT_OPT_HOST_PIN, "math": _lib.FT_OPT_MATH,
               "tail_k": _lib.FT_OPT_TAIL_K, "guided": _lib.FT_OPT_GUIDED, "chunk": _lib.FT_OPT_CHUNK, "cull": _lib.FT_OPT_CULL, "escape": _lib.FT_OPT_ESCAPE, "lazy_union": _lib.FT_OPT_LAZY_UNION, "carved": _lib.FT_OPT_CARVED, "reuse": _lib.FT_OPT_REUSE}

    def set_option(self, name, value):
        """ft_ctx_set_option: per-context switches (the library reads no environment variables)"""
        check(lib.ft_ctx_set_option(self._ctx, self.OPTIONS[name], int(value)))

    def get_option(self, name):
        v = C.c_int32()
        check(lib.ft_ctx_get_option(self._ctx, self.OPTIONS[name], C.byref(v)))
        return int(v.value)

    # constructor twins ---------------------------------------------------------------------------
    def sphere(self, c, r): return check(lib.ft_form_sphere(self._ctx, C.byref(_lib.Sphere(_vec(c), r))))
    def capsule(self, a, b, r): return check(lib.ft_form_capsule(self._ctx, C.byref(_lib.Capsule(_vec(a), _vec(b), r))))
    def torus(self, c, n, R, r): return check(lib.ft_form_torus(self._ctx, C.byref(_lib.Torus(_vec(c), _vec(n), R, r))))
    def triangle(self, a, b, c, r): return check(lib.ft_form_triangle(self._ctx, C.byref(_lib.Triangle(_vec(a), _vec(b), _vec(c), r))))
    def box(self, c, h): return check(lib.ft_form_box(self._ctx, C.byref(_lib.Box(_vec(c), _vec(h)))))
    def form_union(self, hs): return check(lib.ft_form_union(self._ctx, *_handles(hs)))
    def form_subtract(self, a, b): return check(lib.ft_form_subtract(self._ctx, a, b))
    def form_intersect(self, hs): return check(lib.ft_form_intersect(self._ctx, *_handles(hs)))
    def form_union_smooth(self, k, hs): return check(lib.ft_form_union_smooth(self._ctx, k, *_handles(hs)))
    def material_solid(self, rgb): return check(lib.ft_material_solid(self._ctx, _f3(rgb)))
    def material_glass(self, tint, ior, dispersion): return check(lib.ft_material_glass(self._ctx, _f3(tint), ior, dispersion))
    def object_create(self, m, f): return check(lib.ft_object_create(self._ctx, m, f))
    def object_union(self, hs): return check(lib.ft_object_union(self._ctx, *_handles(hs)))
    def object_subtract(self, o, f): return check(lib.ft_object_subtract(self._ctx, o, f))
    def object_intersect(self, o, hs): return check(lib.ft_object_intersect(self._ctx, o, *_handles(hs)))
    def light_directional(self, d, c): return check(lib.ft_light_directional(self._ctx, _f3(d), _f3(c)))
    def light_point(self, p, c): return check(lib.ft_light_point(self._ctx, _f3(p), _f3(c)))

    def form_boundary(self, h):
        b = _lib.Boundary()
        check(lib.ft_form_boundary(self._ctx, h, C.byref(b)))
        return (b.center.x, b.center.y, b.center.z, b.radius)

    def object_form(self, h): return check(lib.ft_object_form(self._ctx, h))

    def scene(self, scene):
        """realise + flatten + upload an SdfScene; cached per device."""
        with scene._lock:
            got = scene._realised.get(self.serial)
            if got is None or got._scene is None:
                memo = {}
                obj = realise(scene.Object, self, memo)
                lights = [realise(l, self, memo) for l in scene.Lights]
                got = DeviceScene(self, obj, scene.BackgroundColor, lights)
                scene._realised[self.serial] = got
                self._scenes.append(got)
            return got

    def selftest_libm(self, op, variant, y=0.0, lo_bits=0, n_chunks=256):
        """ft_selftest_libm: checksums of the device restatement of glibc's expf (op 0) / logf (1) / powf(x, y) (2) per 2^24 inputs"""
        sums = (C.c_uint64 * n_chunks)()
        check(lib.ft_selftest_libm(self._ctx, int(op), int(variant), float(y), int(lo_bits), int(n_chunks), sums))
        return np.array(sums, np.uint64)

    def selftest_fastmath(self):
        m = (C.c_uint64 * 3)()
        check(lib.ft_selftest_fastmath(self._ctx, m))
        return {"sqrt": int(m[0]), "exp": int(m[1]), "exp_near": int(m[2])}

    def math_eval(self, op, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        yp = None
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.float32)
            yp = y.ctypes.data_as(C.c_void_p)
        check(lib.ft_math_eval(self._ctx, op, x.ctypes.data_as(C.c_void_p), yp, x.size, out.ctypes.data_as(C.c_void_p)))
        return out


class DeviceScene:
    """ft_scene: the flattened immutable scene resident in HBM."""

    def __init__(self, device, obj, bg, lights):
        self.device = device
        self._object = obj
        p = C.c_void_p()
        hs, n = _handles(lights)
        check(lib.ft_scene_create(device._ctx, obj, _f3(bg), hs, n, C.byref(p)))
        self._scene = p

    def close(self):
        if self._scene:
            lib.ft_scene_destroy(self._scene)
            self._scene = None

    def boundary(self):
        """scene.Object.Form.Boundary as (cx, cy, cz, radius)"""
        return self.device.form_boundary(self.device.object_form(self._object))

    def info(self):
        i = _lib.SceneInfo()
        check(lib.ft_scene_info_get(self._scene, C.byref(i)))
        return {k: getattr(i, k) for k, _ in i._fields_}

    def support_sphere(self):
        """(cx, cy, cz, radius) of the sphere outside of which (grown by epsilon) no evaluation can be a hit; radius < 0: none known"""
        cr = (C.c_float * 4)()
        check(lib.ft_scene_support_sphere(self._scene, cr))
        return tuple(float(v) for v in cr)

    def grid(self, g=0):
        info = (C.c_float * 6)()
        counts = (C.c_int32 * 3)()
        nc, ni = C.c_int32(), C.c_int32()
        check(lib.ft_scene_grid_shape(self._scene, g, info, counts, C.byref(nc), C.byref(ni)))
        cell_start = np.empty(nc.value + 1, np.uint32)
        centers = np.empty((nc.value, 3), np.float32)
        lower = np.empty(ni.value, np.float32)
        child = np.empty(ni.value, np.int32)
        check(lib.ft_scene_grid_dump(self._scene, g, *(a.ctypes.data_as(C.c_void_p) for a in (cell_start, centers, lower, child))))
        return {"aabbMin": np.array(info[0:3], np.float32), "cellSizeInv": np.array(info[3:6], np.float32),
                "counts": tuple(counts), "cell_start": cell_start, "centers": centers, "lower": lower, "child": child}

    def _params(self, imageSize, epsilon, length, x0=0, n_columns=None, stripe_width=None, stripe_ranks=1, stripe_rank=0,
                spp=1, ao_samples=0, ao_radius=0.0, max_bounces=0, spectral=0):
        """spp / ao_samples / ao_radius / max_bounces / spectral are EXTENSIONS (not in the reference);
        defaults = the reference."""
        W, H = int(imageSize.X), int(imageSize.Y)
        if n_columns is None:
            n_columns = W - x0 if stripe_ranks == 1 else W // stripe_ranks
        if stripe_width is None:
            stripe_width = n_columns
        return _lib.RenderParams(W, H, int(x0), int(n_columns), int(stripe_width), int(stripe_ranks), int(stripe_rank),
                                 int(spp), float(epsilon), float(length), int(ao_samples), float(ao_radius),
                                 int(max_bounces), int(spectral))

    def render(self, epsilon, length, imageSize, camera, out=None, **tiling):
        """Image.render (Image.fs:26-35) -> (FColor[X,Y] as float32 [n_columns, Y, 3], stats dict).  `out`: a float32 array
        of that shape to render into (e.g. one page-locked with Device.host_register)."""
        p = self._params(imageSize, epsilon, length, **tiling)
        if out is None:
            out = np.empty((p.n_columns, p.height, 3), np.float32)
        elif out.dtype != np.float32 or out.shape != (p.n_columns, p.height, 3) or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float32 array of shape (n_columns, Y, 3)")
        st = _lib.Stats()
        check(lib.ft_render(self.device._ctx, self._scene, C.byref(camera._c), C.byref(p), out.ctypes.data_as(C.c_void_p), C.byref(st)))
        return out, st.as_dict()

    def render_device(self, epsilon, length, imageSize, camera, d_out_ptr, **tiling):
        """asynchronous render into device memory (pointer as int); pair with collect_stats()."""
        p = self._params(imageSize, epsilon, length, **tiling)
        check(lib.ft_render_device(self.device._ctx, self._scene, C.byref(camera._c), C.byref(p), C.c_void_p(d_out_ptr)))
        return p.n_columns

    def collect_stats(self):
        st = _lib.Stats()
        check(lib.ft_collect_stats(self.device._ctx, C.byref(st)))
        return st.as_dict()

    def render_colors(self, epsilon, length, imageSize, camera, gamma=2.2, seed=None, bmp_order=False, **ext):
        """Program.fs:90-100 on the device: Image.render, then Image.toColors gamma (and, with bmp_order, the scan-line
        order of Image.toBitmap) — only 3 bytes per pixel leave the GPU.  seed None: no dithering noise (u = 0.5).
        Returns (uint8 [X, Y, 3] R,G,B — or [Y, X, 3] B,G,R rows from the top with bmp_order —, max, stats)."""
        p = self._params(imageSize, epsilon, length, **ext)
        tm = _tonemap_params(gamma, seed, bmp_order)
        out = np.empty((p.height, p.width, 3) if bmp_order else (p.width, p.height, 3), np.uint8)
        st = _lib.Stats()
        mx = C.c_float()
        check(lib.ft_render_colors(self.device._ctx, self._scene, C.byref(camera._c), C.byref(p), C.byref(tm),
                                   out.ctypes.data_as(C.c_void_p), C.byref(mx), C.byref(st)))
        return out, float(mx.value), st.as_dict()

    def trace_rays(self, rays):
        """SdfScene.trace over n rays given as float32 [n, 8] (Origin, Direction, Length, Epsilon)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.empty((rays.shape[0], 3), np.float32)
        st = _lib.Stats()
        check(lib.ft_trace_rays(self.device._ctx, self._scene, rays.ctypes.data_as(C.c_void_p), rays.shape[0],
                                out.ctypes.data_as(C.c_void_p), C.byref(st)))
        return out, st.as_dict()

    def _try_trace(self, fn, rays, width):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.empty((rays.shape[0], width), np.float32)
        st = _lib.Stats()
        check(fn(self.device._ctx, self._scene, rays.ctypes.data_as(C.c_void_p), rays.shape[0], out.ctypes.data_as(C.c_void_p), C.byref(st)))
        return out, st.as_dict()

    def form_try_trace(self, rays):
        """SdfForm.tryTrace scene.Object.Form over rays [n, 8] (SdfForm.fs:93-104) -> float32 [n, 10]:
        Ray at the hit (8), Distance, hit flag (int32 bits; 0 = ValueNone, row is zeros)"""
        return self._try_trace(lib.ft_form_try_trace, rays, 10)

    def object_try_trace(self, rays):
        """SdfObject.tryTrace scene.Object over rays [n, 8] (SdfObject.fs:66-78) -> float32 [n, 16]:
        Ray pulled back by epsilon (8), Normal (3), Color (3), hit flag (int32 bits), 0"""
        return self._try_trace(lib.ft_object_try_trace, rays, 16)

    def eval_distance(self, points):
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        d = np.empty(pts.shape[0], np.float32)
        m = np.empty(pts.shape[0], np.int32)
        check(lib.ft_eval_distance(self.device._ctx, self._scene, pts.ctypes.data_as(C.c_void_p), pts.shape[0],
                                   d.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p)))
        return d, m


def _tonemap_params(gamma, seed, bmp_order):
    return _lib.TonemapParams(float(gamma), 0 if seed is None else 1, 0 if seed is None else int(seed) & 0xFFFFFFFF, 1 if bmp_order else 0)


def tone_map_device(device, d_frame_ptr, X, Y, gamma=2.2, seed=None, bmp_order=False, d_out_ptr=None):
    """Image.toColors on a frame that sits in HBM (pointer as int).  With d_out_ptr the 8-bit image stays on the device
    (asynchronous, returns None); otherwise it is copied to the host: (uint8 array, max)."""
    tm = _tonemap_params(gamma, seed, bmp_order)
    if d_out_ptr is not None:
        check(lib.ft_tone_map_device(device._ctx, C.c_void_p(d_frame_ptr), int(X), int(Y), C.byref(tm), C.c_void_p(d_out_ptr)))
        return None
    out = np.empty((Y, X, 3) if bmp_order else (X, Y, 3), np.uint8)
    mx = C.c_float()
    check(lib.ft_tone_map(device._ctx, C.c_void_p(d_frame_ptr), int(X), int(Y), C.byref(tm), out.ctypes.data_as(C.c_void_p), C.byref(mx)))
    return out, float(mx.value)


def spectral_table(nw):
    """EXTENSION: the library's wavelength table, float32 [nw, 4] = rgb weight, Cauchy term (ft_spectral_table)."""
    out = np.empty((int(nw), 4), np.float32)
    check(lib.ft_spectral_table(int(nw), out.ctypes.data_as(C.c_void_p)))
    return out


def render_multi(devices, scene, epsilon, length, imageSize, camera, stripe_width=16):
    """ft_render_multi: single-process multi-GPU render (one host thread + context per device inside the
    library, column stripes, ONE ncclGather to devices[0], de-interleave on the way to the host).
    Returns (float32 [X, Y, 3], stats)."""
    scenes = [devices[0].scene(scene)]
    for d in devices[1:]:
        p = C.c_void_p()
        check(lib.ft_scene_clone(scenes[0]._scene, d._ctx, C.byref(p)))
        clone = DeviceScene.__new__(DeviceScene)
        clone.device, clone._scene = d, p
        scenes.append(clone)
    n = len(devices)
    ctxs = (C.c_void_p * n)(*[d._ctx for d in devices])
    scs = (C.c_void_p * n)(*[s._scene for s in scenes])
    W, H = int(imageSize.X), int(imageSize.Y)
    p = _lib.RenderParams(W, H, 0, W, int(stripe_width), 1, 0, 1, float(epsilon), float(length), 0, 0.0, 0, 0)
    out = np.empty((W, H, 3), np.float32)
    st = _lib.Stats()
    try:
        check(lib.ft_render_multi(ctxs, scs, n, C.byref(camera._c), C.byref(p), out.ctypes.data_as(C.c_void_p), C.byref(st)))
    finally:
        for s in scenes[1:]:
            s.close()
    return out, st.as_dict()


class SceneTrace:
    """The value `SdfScene.trace scene`: callable on one ray (8 floats) -> FColor."""

    def __init__(self, scene, device=None):
        self.scene = scene
        self.device = device

    def resolve(self):
        dev = self.device if self.device is not None else Device.default(0)
        return dev.scene(self.scene)

    def __call__(self, ray):
        out, _ = self.resolve().trace_rays(np.asarray(ray, np.float32).reshape(1, 8))
        return FColor(tuple(float(c) for c in out[0]))


# --------------------------------------------------------------------------------------------------
# Camera.fs / Image.fs
# --------------------------------------------------------------------------------------------------
class Lens:
    def __init__(self, NearPlaneSize):
        self.NearPlaneSize = float(NearPlaneSize)

    @staticmethod
    def create(fieldOfView):
        """Camera.fs:11-14: sin(fov * 0.5) — radians, as in the reference (Program.fs:21 passes 60.0f)."""
        return Lens(lib.ft_lens_create(float(np.float32(fieldOfView))))


class Camera:
    def __init__(self, c):
        self._c = c

    @staticmethod
    def lookAt(Position, LookAt, Up, Lens):
        c = _lib.CameraS()
        check(lib.ft_camera_look_at(_f3(_v3(Position)), _f3(_v3(LookAt)), _f3(_v3(Up)), Lens.NearPlaneSize, C.byref(c)))
        return Camera(c)

    def as_array(self):
        return np.frombuffer(bytes(self._c), dtype=np.float32).copy()

    @staticmethod
    def uniformPixelToRay(epsilon, length, camera, position):
        """Camera.uniformPixelToRay (Camera.fs:44-54), host side, float32 operation by operation:
        Direction = normalize(Forward + (px - 0.5) * RightScaled + (py - 0.5) * UpScaled) -> ray as 8 floats
        (Origin, Direction, Length, Epsilon; Types.fs:9-17)."""
        F = np.float32
        a = camera.as_array()
        pos, fw, up, rt = a[0:3], a[3:6], a[6:9], a[9:12]
        px, py = F(position[0]), F(position[1])
        d = (fw + F(px - F(0.5)) * rt) + F(py - F(0.5)) * up
        d = d / np.sqrt(F(F(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]))
        return np.concatenate([pos, d, [F(length), F(epsilon)]]).astype(F)

    Position = property(lambda s: (s._c.position.x, s._c.position.y, s._c.position.z))
    Forward = property(lambda s: (s._c.forward.x, s._c.forward.y, s._c.forward.z))
    UpScaled = property(lambda s: (s._c.up_scaled.x, s._c.up_scaled.y, s._c.up_scaled.z))
    RightScaled = property(lambda s: (s._c.right_scaled.x, s._c.right_scaled.y, s._c.right_scaled.z))


class ImageSize:
    def __init__(self, X, Y):
        self.X, self.Y = int(X), int(Y)

    @staticmethod
    def getUniformPixelPos(size):
        """ImageSize.getUniformPixelPos (Image.fs:17-23): (x, y) -> (x / max(X, Y), y / max(X, Y)) in float32"""
        m = np.float32(max(size.X, size.Y))
        return lambda x, y: (np.float32(x) / m, np.float32(y) / m)


class Ray:
    """Ray.fs:6-15 on rays stored as 8 floats (Origin, Direction, Length, Epsilon)"""

    @staticmethod
    def get(length, ray):
        r = np.asarray(ray, np.float32)
        return r[0:3] + r[3:6] * np.float32(length)

    @staticmethod
    def move(length, ray):
        r = np.asarray(ray, np.float32).copy()
        r[0:3] = Ray.get(length, ray)
        r[6] = r[6] - np.float32(length)
        return r

    @staticmethod
    def setDirection(direction, ray):
        r = np.asarray(ray, np.float32).copy()
        r[3:6] = np.asarray(direction, np.float32)
        return r


class Image:
    @staticmethod
    def render(epsilon, length, imageSize, camera, trace):
        """Image.render epsilon length imageSize camera trace (Image.fs:26-35).  `trace` must be the
        value of SdfScene.trace (an opaque Python closure cannot run on the GPU); returns the
        FColor[X,Y] image as a float32 array [X, Y, 3]."""
        if not isinstance(trace, SceneTrace):
            raise TypeError("Image.render needs `SdfScene.trace scene`; arbitrary closures have no GPU form")
        img, _ = trace.resolve().render(epsilon, length, imageSize, camera)
        return img

    @staticmethod
    def renderScene(epsilon, length, imageSize, camera, scene, device=None):
        return Image.render(epsilon, length, imageSize, camera, SdfScene.trace(scene, device))

    @staticmethod
    def toColors(gamma, rng, image, device=None, bmp_order=False):
        """Image.toColors gamma rng image (Image.fs:37-50) on the GPU for a host FColor[X,Y] (float32 [X, Y, 3]) ->
        uint8 [X, Y, 3].  `rng`: None = no dithering noise (u = 0.5); an int = seed of the counter-based noise (the
        reference's shared System.Random is racy, so its low bit is not reproducible: +-1 LSB comparable)."""
        img = np.ascontiguousarray(image, dtype=np.float32)
        X, Y = img.shape[0], img.shape[1]
        dev = device if device is not None else Device.default(0)
        tm = _tonemap_params(gamma, rng, bmp_order)
        out = np.empty((Y, X, 3) if bmp_order else (X, Y, 3), np.uint8)
        check(lib.ft_tone_map_host(dev._ctx, img.ctypes.data_as(C.c_void_p), X, Y, C.byref(tm), out.ctypes.data_as(C.c_void_p), None))
        return out

    @staticmethod
    def saveBitmap(path, colors):
        """Image.saveBitmap (Image.fs:88-90) for the uint8 [X, Y, 3] value of toColors: 24-bpp BMP, host side"""
        from .postprocess import saveBitmap
        saveBitmap(path, colors)
