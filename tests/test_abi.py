"""The drop-in boundary: libfraytracer_hip.so loads and exports every symbol include/*.h declares, with
the struct layouts of the reference records.  No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import fraytracer_amd as ft
from fraytracer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fraytracer_hip.h")


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(ft_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 35
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (ft_[a-z0-9_]+)\b", out))
    missing = [n for n in names if n not in exported]
    assert not missing, f"declared in the header but not exported: {missing}"
    unbound = [n for n in names if n not in _lib.SYMBOLS]
    assert not unbound, f"declared in the header but not bound by the Python mirror: {unbound}"
    for n in names:
        getattr(_lib.lib, n)


def test_blittable_layouts_match_the_reference_records():
    # Types.fs:9-24 Ray 32 B / SdfBoundary 16 B; Camera.fs:16-22 48 B; primitive structs SdfForm.fs:118-212
    assert C.sizeof(_lib.Ray) == 32 and C.sizeof(_lib.Boundary) == 16 and C.sizeof(_lib.CameraS) == 48
    assert C.sizeof(_lib.Sphere) == 16 and C.sizeof(_lib.Capsule) == 28 and C.sizeof(_lib.Torus) == 32 and C.sizeof(_lib.Triangle) == 40
    assert C.sizeof(_lib.RenderParams) == 56 and C.sizeof(_lib.Stats) == 80
    assert _lib.lib.ft_abi_version() == 5


def test_the_library_is_built_in_tree_and_is_not_the_oracle():
    assert _lib.LIB_PATH.startswith(os.path.join(ROOT, "fraytracer_amd"))
    deps = subprocess.check_output(["ldd", _lib.LIB_PATH], text=True)
    assert "libamdhip64" in deps and "ft_oracle" not in deps
    assert "librccl" not in deps            # RCCL is bound lazily by ft_render_multi (one RCCL per process)
    # the product never references the oracle directory
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fraytracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                assert "oracle" not in open(os.path.join(dirpath, f), errors="ignore").read().replace("the oracle", "").replace("test oracle", "").replace("CPU oracle", ""), f


def test_the_loaded_library_was_built_from_the_sources_next_to_it():
    """ft_build_info() carries the hash of the sources the shared object was compiled from (csrc/source_hash.py, written into
    build_hash.h by the Makefile).  A library left over from an experiment — edited sources rebuilt and then reverted, or a
    `make profile` / `make experiment` build put in the product's place — fails here instead of quietly being the thing the GPU
    suite and the bench measure (that happened in round 2: DESIGN.md section 10).  Also run on the GPU box (`-m gpu` selects
    test_gpu_parity.py::test_product_build_is_the_one_under_test, which asserts the same)."""
    info = ft.build_info()
    assert info["kind"] == "product", info
    assert info["src"] == ft.source_hash(), (info, ft.source_hash(), "rebuild: make -C fraytracer_amd/csrc")


def test_the_library_reads_no_environment_variables():
    """tuning switches are per-context options (ft_ctx_set_option), not hidden global state"""
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", _lib.LIB_PATH], text=True)
    assert not re.search(r"\b(secure_)?getenv\b", undefined), undefined
    for f in ("capi.cpp", "scene.cpp", "multi.cpp", "kernels.hip"):
        assert "getenv" not in open(os.path.join(ROOT, "fraytracer_amd", "csrc", f)).read(), f
    host = ft.Device(-1)                       # options are plain context state: usable without a GPU
    try:
        assert host.get_option("refill_min") == 64 and host.get_option("host_pin") == 1
        host.set_option("refill_min", 32); host.set_option("host_chunks", 2); host.set_option("host_pin", 0); host.set_option("max_blocks_per_cu", 3)
        assert [host.get_option(k) for k in ("refill_min", "host_chunks", "host_pin", "max_blocks_per_cu")] == [32, 2, 0, 3]
        assert [host.get_option(k) for k in ("math", "tail_k", "guided", "chunk")] == [0, -1, 0, 64]
        host.set_option("math", 2); host.set_option("tail_k", 8); host.set_option("guided", 1); host.set_option("chunk", 16)
        assert [host.get_option(k) for k in ("math", "tail_k", "guided", "chunk")] == [2, 8, 1, 16]
        # the three exact shortcuts are on unless switched off
        assert [host.get_option(k) for k in ("cull", "escape", "lazy_union")] == [1, 1, 1]
        for k in ("cull", "escape", "lazy_union"):
            host.set_option(k, 0); assert host.get_option(k) == 0
            host.set_option(k, 1)
        for name, bad in (("refill_min", 0), ("refill_min", 65), ("host_chunks", 17), ("host_pin", 2), ("max_blocks_per_cu", -1),
                          ("math", 3), ("tail_k", 65), ("tail_k", -2), ("guided", 2), ("chunk", 48), ("cull", 2), ("escape", -1), ("lazy_union", 2)):
            try:
                host.set_option(name, bad)
            except ft.FrayTracerError as e:
                assert e.code == _lib.FT_ERR_INVALID
            else:
                raise AssertionError((name, bad))
    finally:
        host.close()


C99_PROBE = r"""
#include <stddef.h>
#include <stdio.h>
#include "fraytracer_hip.h"
#define S(t) printf("sizeof " #t " %zu\n", sizeof(t))
#define O(t, f) printf("offsetof " #t "." #f " %zu\n", offsetof(t, f))
int main(void) {
    S(ft_vec3); S(ft_ray); S(ft_boundary); S(ft_form_trace_result); S(ft_object_trace_result); S(ft_sphere); S(ft_capsule);
    S(ft_torus); S(ft_triangle); S(ft_box); S(ft_camera); S(ft_render_params); S(ft_stats); S(ft_scene_info); S(ft_handle);
    S(ft_tonemap_params);
    O(ft_ray, direction); O(ft_ray, length); O(ft_ray, epsilon); O(ft_boundary, radius);
    O(ft_form_trace_result, distance); O(ft_form_trace_result, hit);
    O(ft_object_trace_result, normal); O(ft_object_trace_result, color); O(ft_object_trace_result, hit);
    O(ft_capsule, to); O(ft_capsule, radius); O(ft_torus, normal); O(ft_torus, major_radius); O(ft_torus, minor_radius);
    O(ft_triangle, v2); O(ft_triangle, v3); O(ft_triangle, radius); O(ft_box, half_extent);
    O(ft_camera, forward); O(ft_camera, up_scaled); O(ft_camera, right_scaled);
    O(ft_render_params, x0); O(ft_render_params, stripe_width); O(ft_render_params, spp); O(ft_render_params, epsilon);
    O(ft_render_params, length); O(ft_render_params, ao_samples); O(ft_render_params, ao_radius); O(ft_render_params, max_bounces);
    O(ft_render_params, spectral);
    O(ft_stats, hits_primary); O(ft_stats, sdf_evals); O(ft_stats, flags); O(ft_stats, kernel_ms); O(ft_stats, culled_fraction); O(ft_stats, wave_evals); O(ft_stats, shader_mhz);
    O(ft_tonemap_params, gamma); O(ft_tonemap_params, dither); O(ft_tonemap_params, seed); O(ft_tonemap_params, bmp_order);
    return 0;
}
"""


def test_header_is_plain_c99_with_the_reference_layouts(tmp_path):
    """The boundary is usable from C (and so from any FFI): the header compiles as strict C99 on its own, and a C
    program — not ctypes — reports the sizes / offsets the F# [<Struct>] records and the P/Invoke twins rely on
    (Types.fs:9-24, 32-37, 57-65; SdfForm.fs:118-212; FtCamera / FtRenderParams / FtStats in host/fsharp)."""
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", HEADER])
    src = tmp_path / "probe.c"
    src.write_text(C99_PROBE)
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", inc, "-o", str(exe), str(src)])
    got = {}
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        kind, name, val = line.split()
        got[(kind, name)] = int(val)
    want_sizes = {"ft_vec3": 12, "ft_ray": 32, "ft_boundary": 16, "ft_form_trace_result": 40, "ft_object_trace_result": 64,
                  "ft_sphere": 16, "ft_capsule": 28, "ft_torus": 32, "ft_triangle": 40, "ft_box": 24, "ft_camera": 48,
                  "ft_render_params": 56, "ft_stats": 80, "ft_scene_info": 44, "ft_handle": 4, "ft_tonemap_params": 16}
    for k, v in want_sizes.items():
        assert got[("sizeof", k)] == v, (k, got[("sizeof", k)], v)
    want_offsets = {"ft_ray.direction": 12, "ft_ray.length": 24, "ft_ray.epsilon": 28, "ft_boundary.radius": 12,
                    "ft_form_trace_result.distance": 32, "ft_form_trace_result.hit": 36,
                    "ft_object_trace_result.normal": 32, "ft_object_trace_result.color": 44, "ft_object_trace_result.hit": 56,
                    "ft_capsule.to": 12, "ft_capsule.radius": 24, "ft_torus.normal": 12, "ft_torus.major_radius": 24,
                    "ft_torus.minor_radius": 28, "ft_triangle.v2": 12, "ft_triangle.v3": 24, "ft_triangle.radius": 36,
                    "ft_box.half_extent": 12, "ft_camera.forward": 12, "ft_camera.up_scaled": 24, "ft_camera.right_scaled": 36,
                    "ft_render_params.x0": 8, "ft_render_params.stripe_width": 16, "ft_render_params.spp": 28,
                    "ft_render_params.epsilon": 32, "ft_render_params.length": 36, "ft_render_params.ao_samples": 40,
                    "ft_render_params.ao_radius": 44, "ft_render_params.max_bounces": 48, "ft_render_params.spectral": 52,
                    "ft_stats.hits_primary": 24, "ft_stats.sdf_evals": 40, "ft_stats.flags": 48, "ft_stats.kernel_ms": 56,
                    "ft_stats.culled_fraction": 60, "ft_stats.wave_evals": 64, "ft_stats.shader_mhz": 72,
                    "ft_tonemap_params.gamma": 0, "ft_tonemap_params.dither": 4, "ft_tonemap_params.seed": 8,
                    "ft_tonemap_params.bmp_order": 12}
    for k, v in want_offsets.items():
        assert got[("offsetof", k)] == v, (k, got[("offsetof", k)], v)
    # the ctypes mirror agrees with the C compiler
    for cname, ctype in (("ft_ray", _lib.Ray), ("ft_camera", _lib.CameraS), ("ft_render_params", _lib.RenderParams), ("ft_stats", _lib.Stats),
                         ("ft_triangle", _lib.Triangle), ("ft_torus", _lib.Torus), ("ft_capsule", _lib.Capsule)):
        assert C.sizeof(ctype) == got[("sizeof", cname)], cname


def test_the_makefile_lists_every_header_an_object_includes():
    """capi.o embeds the hash of ALL sources (ft_build_info), so an object that misses a header in its rule could stay stale while the
    hash moves on — the stale-build failure the hash exists to prevent.  Every `name.o:` rule of csrc/Makefile must list every project header its
    source includes, directly or through another header."""
    csrc = os.path.join(ROOT, "fraytracer_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read().replace("$(ABI)", "../../include/fraytracer_hip.h")

    def includes(path, seen):
        for inc in re.findall(r'^\s*#include\s+"([^"]+)"', open(path).read(), flags=re.M):
            full = os.path.normpath(os.path.join(os.path.dirname(path), inc))
            if full not in seen and os.path.exists(full):
                seen.add(full)
                includes(full, seen)
        return seen

    for obj, src in (("kernels.o", "kernels.hip"), ("scene.o", "scene.cpp"), ("capi.o", "capi.cpp"), ("multi.o", "multi.cpp")):
        rule = re.search(r"^%s:(.*)$" % re.escape(obj), mk, flags=re.M).group(1).split()
        listed = {os.path.normpath(os.path.join(csrc, r)) for r in rule}
        need = includes(os.path.join(csrc, src), set())
        need.discard(os.path.join(csrc, "build_hash.h")) if obj != "capi.o" else None
        missing = sorted(os.path.relpath(n, csrc) for n in need if n not in listed)
        assert not missing, f"{obj}: the Makefile rule does not list {missing}"


def test_glibc_build_of_this_host_asks_the_running_libm():
    """the build of glibc's expf the ifunc resolver picked is decided by evaluating the two inputs on which the FMA and SSE2 builds differ,
    not by parsing /proc/cpuinfo: a child process whose libm is switched by GLIBC_TUNABLES must report the other build."""
    import sys
    here = ft.glibc_build_of_this_host()
    assert here in (_lib.FT_MATH_GLIBC_FMA, _lib.FT_MATH_GLIBC_SSE2)
    code = "import sys; sys.path.insert(0, %r); import fraytracer_amd as ft; print(ft.glibc_build_of_this_host())" % ROOT
    env = dict(os.environ, GLIBC_TUNABLES="glibc.cpu.hwcaps=-FMA,-AVX2")
    child = int(subprocess.check_output([sys.executable, "-c", code], env=env, text=True).strip().splitlines()[-1])
    assert child == _lib.FT_MATH_GLIBC_SSE2
    if here == _lib.FT_MATH_GLIBC_FMA:
        assert child != here
