#!/usr/bin/env python3
"""Times every code object in tools/_asmvar/ as the lean kernel of the diagnostic library (C3, 4096^2): kernel ms (HIP events),
the shader clock of the run and shader cycles per frame; the frame must equal the built-in kernel's bit for bit."""
import ctypes as C
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FRAYTRACER_HIP_LIB"] = os.path.join(ROOT, "fraytracer_amd", "libfraytracer_hip_exp.so")
import numpy as np
import torch
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn, _lib

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
size = 4096
scene, _ = syn.config3(size=size)
cam = syn.default_camera()
dev = ft.Device(0)
ds = dev.scene(scene)
slab = torch.empty((size, size, 3), dtype=torch.float32, device="cuda")
S = ft.ImageSize(size, size)
set_hsaco = _lib.lib.ft_debug_set_hsaco
set_hsaco.argtypes = [C.c_char_p]; set_hsaco.restype = C.c_int


def run(tag):
    ds.render_device(syn.EPSILON, syn.RAY_LENGTH, S, cam, slab.data_ptr()); ds.collect_stats()
    ms, mhz = [], []
    for _ in range(frames):
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, S, cam, slab.data_ptr())
        st = ds.collect_stats()
        ms.append(st["kernel_ms"]); mhz.append(st["shader_mhz"])
    h = int(slab.view(torch.int32).sum(dtype=torch.int64).item())
    m, f = float(np.median(ms)), float(np.median(mhz))
    print(json.dumps({"variant": tag, "ms_median": round(m, 3), "ms_min": round(min(ms), 3), "shader_mhz": round(f, 1),
                      "Gcycles_per_frame": round(m * 1e-3 * f * 1e6 / 1e9, 4), "frame_checksum": h}), flush=True)
    return h


ref = run("built-in (pad 15)")
for path in sorted(glob.glob(os.path.join(ROOT, "tools", "_asmvar", "*.hsaco"))):
    rc = set_hsaco(path.encode())
    if rc != 0:
        print(json.dumps({"variant": os.path.basename(path), "error": rc})); continue
    h = run(os.path.basename(path)[:-6])
    if h != ref:
        print(json.dumps({"variant": os.path.basename(path), "error": "frame differs from the built-in kernel"}), flush=True)
set_hsaco(b"")
