#!/usr/bin/env python3
"""The reference's own frame (Program.fs scene, 1000^2) against the resident workgroups per CU and the rays per grab: where does a frame
with only 2.5 tiles per wave slot lose its time?  One JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
dev = ft.Device(0)
cam = syn.default_camera()
out = {}
for name, scene, n in (("Program.fs 1000^2", syn.console_scene()[0], 1000), ("C2 1024^2", syn.config2()[0], 1024), ("C3 1024^2", syn.config3(size=1024)[0], 1024)):
    ds = dev.scene(scene)
    buf = torch.empty((n, n, 3), dtype=torch.float32, device="cuda")
    size = ft.ImageSize(n, n)
    row = {}
    for cap in (0, 6, 5, 4, 3, 2):
        for chunk in (64, 32):
            dev.set_option("max_blocks_per_cu", cap); dev.set_option("chunk", chunk)
            ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr()); ds.collect_stats()
            for _ in range(5):
                ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr())
            st = ds.collect_stats()
            row[f"cap {cap} chunk {chunk}"] = round(st["kernel_ms"] / 5, 3)
    dev.set_option("max_blocks_per_cu", 0); dev.set_option("chunk", 64)
    out[name] = row
print(json.dumps(out))
