"""ctypes binding of libfraytracer_hip.so (include/fraytracer_hip.h).

The library is the product; this module only marshals.  There is no fallback of any kind: if the
shared object is missing the import fails, and if no GPU is visible every render / trace call
raises FrayTracerError (the C side returns FT_ERR_NO_DEVICE).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FRAYTRACER_HIP_LIB: diagnostic builds only (tools/union_divergence.py); the product is the in-tree library
LIB_PATH = os.environ.get("FRAYTRACER_HIP_LIB") or os.path.join(_HERE, "libfraytracer_hip.so")

FT_OPT_REFILL_MIN, FT_OPT_MAX_BLOCKS_PER_CU, FT_OPT_HOST_CHUNKS, FT_OPT_HOST_PIN, FT_OPT_TAIL_K, FT_OPT_MATH, FT_OPT_GUIDED, FT_OPT_CHUNK, FT_OPT_CULL, FT_OPT_ESCAPE, FT_OPT_LAZY_UNION, FT_OPT_CARVED, FT_OPT_REUSE = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13
FT_MATH_FIXED, FT_MATH_GLIBC_FMA, FT_MATH_GLIBC_SSE2 = 0, 1, 2
FT_OK, FT_ERR_INVALID, FT_ERR_NO_DEVICE, FT_ERR_HIP, FT_ERR_UNSUPPORTED, FT_ERR_EMPTY, FT_ERR_COMM = 0, -1, -2, -3, -4, -5, -6


class FrayTracerError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libfraytracer_hip error {code}: {message}")
        self.code = code


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Ray(C.Structure):                      # Types.fs:9-17
    _fields_ = [("origin", Vec3), ("direction", Vec3), ("length", C.c_float), ("epsilon", C.c_float)]


class Boundary(C.Structure):                 # Types.fs:19-24
    _fields_ = [("center", Vec3), ("radius", C.c_float)]


class Sphere(C.Structure):
    _fields_ = [("center", Vec3), ("radius", C.c_float)]


class Capsule(C.Structure):
    _fields_ = [("from_", Vec3), ("to", Vec3), ("radius", C.c_float)]


class Torus(C.Structure):
    _fields_ = [("center", Vec3), ("normal", Vec3), ("major_radius", C.c_float), ("minor_radius", C.c_float)]


class Triangle(C.Structure):
    _fields_ = [("v1", Vec3), ("v2", Vec3), ("v3", Vec3), ("radius", C.c_float)]


class Box(C.Structure):
    _fields_ = [("center", Vec3), ("half_extent", Vec3)]


class CameraS(C.Structure):                  # Camera.fs:16-22
    _fields_ = [("position", Vec3), ("forward", Vec3), ("up_scaled", Vec3), ("right_scaled", Vec3)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("x0", C.c_int32), ("n_columns", C.c_int32),
                ("stripe_width", C.c_int32), ("stripe_ranks", C.c_int32), ("stripe_rank", C.c_int32),
                ("spp", C.c_int32), ("epsilon", C.c_float), ("length", C.c_float),
                ("ao_samples", C.c_int32), ("ao_radius", C.c_float),
                ("max_bounces", C.c_int32), ("spectral", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_ext", C.c_uint64),
                ("hits_primary", C.c_uint64), ("hits_shadow", C.c_uint64), ("sdf_evals", C.c_uint64),
                ("flags", C.c_uint64), ("kernel_ms", C.c_float), ("culled_fraction", C.c_float), ("wave_evals", C.c_uint64),
                ("shader_mhz", C.c_float), ("tail_fraction", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("reserved")}


class TonemapParams(C.Structure):
    _fields_ = [("gamma", C.c_float), ("dither", C.c_int32), ("seed", C.c_uint32), ("bmp_order", C.c_int32)]


class SceneInfo(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("n_instr", "n_slots", "n_consts", "n_grids", "n_children", "n_cells",
                                          "n_items", "n_lights", "n_materials", "fast_path", "cull_pc")]


# every symbol include/fraytracer_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_H = C.c_int32
_F3 = C.POINTER(C.c_float)
SYMBOLS = {
    "ft_abi_version": (C.c_int, []),
    "ft_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "ft_ctx_destroy": (None, [_P]),
    "ft_last_error": (C.c_char_p, []),
    "ft_ctx_set_stream": (C.c_int, [_P, _P]),
    "ft_build_info": (C.c_char_p, []),
    "ft_ctx_set_option": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "ft_ctx_get_option": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32)]),
    "ft_form_sphere": (_H, [_P, C.POINTER(Sphere)]),
    "ft_form_capsule": (_H, [_P, C.POINTER(Capsule)]),
    "ft_form_torus": (_H, [_P, C.POINTER(Torus)]),
    "ft_form_triangle": (_H, [_P, C.POINTER(Triangle)]),
    "ft_form_box": (_H, [_P, C.POINTER(Box)]),
    "ft_form_union": (_H, [_P, C.POINTER(_H), C.c_int32]),
    "ft_form_subtract": (_H, [_P, _H, _H]),
    "ft_form_intersect": (_H, [_P, C.POINTER(_H), C.c_int32]),
    "ft_form_union_smooth": (_H, [_P, C.c_float, C.POINTER(_H), C.c_int32]),
    "ft_form_boundary": (C.c_int, [_P, _H, C.POINTER(Boundary)]),
    "ft_material_solid": (_H, [_P, _F3]),
    "ft_material_glass": (_H, [_P, _F3, C.c_float, C.c_float]),
    "ft_form_try_trace": (C.c_int, [_P, _P, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(Stats)]),
    "ft_object_try_trace": (C.c_int, [_P, _P, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(Stats)]),
    "ft_spectral_table": (C.c_int, [C.c_int32, C.c_void_p]),
    "ft_object_create": (_H, [_P, _H, _H]),
    "ft_object_union": (_H, [_P, C.POINTER(_H), C.c_int32]),
    "ft_object_subtract": (_H, [_P, _H, _H]),
    "ft_object_intersect": (_H, [_P, _H, C.POINTER(_H), C.c_int32]),
    "ft_object_form": (_H, [_P, _H]),
    "ft_light_directional": (_H, [_P, _F3, _F3]),
    "ft_light_point": (_H, [_P, _F3, _F3]),
    "ft_scene_create": (C.c_int, [_P, _H, _F3, C.POINTER(_H), C.c_int32, C.POINTER(_P)]),
    "ft_scene_destroy": (None, [_P]),
    "ft_lens_create": (C.c_float, [C.c_float]),
    "ft_camera_look_at": (C.c_int, [_F3, _F3, _F3, C.c_float, C.POINTER(CameraS)]),
    "ft_render": (C.c_int, [_P, _P, C.POINTER(CameraS), C.POINTER(RenderParams), _P, C.POINTER(Stats)]),
    "ft_host_register": (C.c_int, [_P, _P, C.c_uint64]),
    "ft_host_unregister": (C.c_int, [_P, _P]),
    "ft_render_device": (C.c_int, [_P, _P, C.POINTER(CameraS), C.POINTER(RenderParams), _P]),
    "ft_collect_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "ft_tone_map_device": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.POINTER(TonemapParams), _P]),
    "ft_tone_map": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.POINTER(TonemapParams), _P, C.POINTER(C.c_float)]),
    "ft_tone_map_host": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.POINTER(TonemapParams), _P, C.POINTER(C.c_float)]),
    "ft_render_colors": (C.c_int, [_P, _P, C.POINTER(CameraS), C.POINTER(RenderParams), C.POINTER(TonemapParams), _P, C.POINTER(C.c_float), C.POINTER(Stats)]),
    "ft_trace_rays": (C.c_int, [_P, _P, _P, C.c_int64, _P, C.POINTER(Stats)]),
    "ft_eval_distance": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P]),
    "ft_scene_clone": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "ft_render_multi": (C.c_int, [C.POINTER(_P), C.POINTER(_P), C.c_int32, C.POINTER(CameraS), C.POINTER(RenderParams), _P, C.POINTER(Stats)]),
    "ft_scene_info_get": (C.c_int, [_P, C.POINTER(SceneInfo)]),
    "ft_scene_grid_shape": (C.c_int, [_P, C.c_int32, _F3, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ft_scene_grid_dump": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P]),
    "ft_scene_support_sphere": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "ft_math_eval": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int64, _P]),
    "ft_selftest_fastmath": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "ft_selftest_libm": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_float, C.c_uint32, C.c_int32, C.POINTER(C.c_uint64)]),
}


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C fraytracer_amd/csrc`.  fraytracer_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()


def build_info():
    """{'src': hash of the sources the loaded library was built from, 'kind': product | profile | experiment}"""
    return dict(kv.split("=", 1) for kv in lib.ft_build_info().decode().split(";"))


def source_hash():
    """the same hash computed from the source tree next to this package (csrc/source_hash.py)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ft_source_hash", os.path.join(_HERE, "csrc", "source_hash.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.source_hash()


def last_error():
    m = lib.ft_last_error()
    return m.decode("utf-8", "replace") if m else ""


def check(rc):
    if rc < 0:
        raise FrayTracerError(rc, last_error())
    return rc
