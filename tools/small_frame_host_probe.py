#!/usr/bin/env python3
"""ft_render (frame in host memory, what Image.render returns) of the reference's own frame — Program.fs scene, 1000^2 — against the number of
column chunks of the host-output pipeline: wall ms per call, median of 30."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
dev = ft.Device(0)
cam = syn.default_camera()
scene, size = syn.console_scene()
ds = dev.scene(scene)
out = {}
for registered in (False, True):
    buf = np.zeros((size.X, size.Y, 3), np.float32)
    if registered:
        dev.host_register(buf)
    for chunks in (1, 2, 4):
        dev.set_option("host_chunks", chunks)
        ds.render(syn.EPSILON, syn.RAY_LENGTH, size, cam, out=buf)
        ts, ks = [], []
        for _ in range(30):
            t0 = time.perf_counter()
            _, st = ds.render(syn.EPSILON, syn.RAY_LENGTH, size, cam, out=buf)
            ts.append((time.perf_counter() - t0) * 1e3); ks.append(st["kernel_ms"])
        out[f"{'registered' if registered else 'pageable'} destination, {chunks} chunk(s)"] = {"wall_ms": round(float(np.median(ts)), 3), "sum_of_kernel_ms": round(float(np.median(ks)), 3)}
    if registered:
        dev.host_unregister(buf)
dev.set_option("host_chunks", 0)
print(json.dumps(out))
