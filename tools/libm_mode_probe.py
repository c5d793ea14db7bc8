#!/usr/bin/env python3
"""C3 4096^2 in the default arithmetic and in FT_OPT_MATH = glibc: kernel ms, Mrays/s, and how far the two frames are apart."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = ft.Device(0)
cam = syn.default_camera()
ds = dev.scene(syn.config3(size=n)[0])
size = ft.ImageSize(n, n)
frames = {}
out = {}
for mode in (0, ft.glibc_build_of_this_host(), 3 - ft.glibc_build_of_this_host()):
    dev.set_option("math", mode)
    buf = torch.empty((n, n, 3), dtype=torch.float32, device="cuda")
    ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr()); ds.collect_stats()
    reps = 3
    for _ in range(reps):
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr())
    st = ds.collect_stats()
    rays = (st["rays_primary"] + st["rays_shadow"]) / reps
    out[mode] = {"kernel_ms": round(st["kernel_ms"] / reps, 3), "Mrays/s": round(rays / (st["kernel_ms"] / reps) / 1e3, 1), "shader_mhz": round(st["shader_mhz"], 1)}
    frames[mode] = buf.cpu().numpy()
a = frames[0].astype(np.float64)
for mode in list(frames)[1:]:
    b = frames[mode].astype(np.float64)
    rel = (np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max(axis=2)
    out[mode].update({"pixels_identical_to_fixed": float((frames[0].view(np.uint32) == frames[mode].view(np.uint32)).all(axis=2).mean()),
                      "pixels_over_1e-4_rel": float((rel > 1e-4).mean()), "max_rel": float(rel.max())})
k = list(frames)
out["fma_vs_sse2_identical"] = bool(np.array_equal(frames[k[1]].view(np.uint32), frames[k[2]].view(np.uint32)))
print(json.dumps({"size": n, "modes": {str(k_): v for k_, v in out.items()}}))
