#!/usr/bin/env python3
"""One library build (FRAYTRACER_HIP_LIB), C3 at 4096^2: kernel ms of K frames (HIP events) and the shader clock the kernel
ran at.  Prints one JSON line.  Used by tools/pad_sweep.py."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
scene, _ = syn.config3(size=size)
cam = syn.default_camera()
dev = ft.Device(0)
ds = dev.scene(scene)
slab = torch.empty((size, size, 3), dtype=torch.float32, device="cuda")
S = ft.ImageSize(size, size)
ds.render_device(syn.EPSILON, syn.RAY_LENGTH, S, cam, slab.data_ptr()); ds.collect_stats()
ms, mhz = [], []
for _ in range(frames):
    ds.render_device(syn.EPSILON, syn.RAY_LENGTH, S, cam, slab.data_ptr())
    st = ds.collect_stats()
    ms.append(round(st["kernel_ms"], 3)); mhz.append(round(st["shader_mhz"], 1))
print(json.dumps({"lib": os.path.basename(os.environ.get("FRAYTRACER_HIP_LIB", "product")), 
                  "ms_min": min(ms), "ms_median": float(np.median(ms)), "ms": ms, "shader_mhz": mhz,
                  "cycles_per_frame_G": round(float(np.median(ms)) * 1e-3 * float(np.median(mhz)) * 1e6 / 1e9, 4)}), flush=True)
