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
    assert C.sizeof(_lib.RenderParams) == 56 and C.sizeof(_lib.Stats) == 72
    assert _lib.lib.ft_abi_version() == 2


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
