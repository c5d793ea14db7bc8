#!/usr/bin/env python3
"""Cost of the device tone map: C3 4096^2 through ft_render_colors (render + Image.toColors on the GPU, 50 MB of bytes to the host)
against ft_render (201 MB of floats to the host) and the kernel alone; and the tone-map kernels by themselves (HIP events via torch)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn, api

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scene, _ = syn.config3(size=size)
cam = syn.default_camera()
dev = ft.Device(0)
stream = torch.cuda.Stream()                         # a non-default stream: the library launches on it, torch events see it
torch.cuda.set_stream(stream)
dev.set_stream(stream.cuda_stream)
ds = dev.scene(scene)
S = ft.ImageSize(size, size)
frame = torch.empty((size, size, 3), dtype=torch.float32, device="cuda")
out8 = torch.empty((size, size, 3), dtype=torch.uint8, device="cuda")
ds.render_device(syn.EPSILON, syn.RAY_LENGTH, S, cam, frame.data_ptr()); torch.cuda.synchronize(); ds.collect_stats()
res = {}
for bmp in (False, True):
    api.tone_map_device(dev, frame.data_ptr(), size, size, 2.2, 19, bmp, out8.data_ptr()); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        api.tone_map_device(dev, frame.data_ptr(), size, size, 2.2, 19, bmp, out8.data_ptr())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    res["tone_map_kernels_ms_bmp_order" if bmp else "tone_map_kernels_ms"] = round(ms, 3)
    res[("bmp " if bmp else "") + "GB/s (12 B read twice + 3 B written per pixel)"] = round(size * size * 27 / ms / 1e6, 1)
host = np.zeros((size, size, 3), np.float32)
for name, fn in (("ft_render (floats to host, destination reused)", lambda: ds.render(syn.EPSILON, syn.RAY_LENGTH, S, cam, out=host)),
                 ("ft_render_colors (bytes to host)", lambda: ds.render_colors(syn.EPSILON, syn.RAY_LENGTH, S, cam, gamma=2.2, seed=19, bmp_order=True))):
    fn()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    res[name + " ms"] = round(float(np.median(ts)), 2)
ds.render_device(syn.EPSILON, syn.RAY_LENGTH, S, cam, frame.data_ptr()); torch.cuda.synchronize()
res["kernel_ms"] = round(ds.collect_stats()["kernel_ms"], 2)
print(json.dumps(res))
