#!/usr/bin/env python3
"""What does the latency mode of the lean kernel cost per ray?  C3 at 2048^2 with every grab cut to 32 / 16 rays: all rounds of a wave then
hold at most 32 / 16 rays, i.e. run 2 / 4 lanes per ray (tail_k = 32) or, with tail_k = 0, the one-ray-per-lane path on part-filled waves."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
dev = ft.Device(0)
cam = syn.default_camera()
n = 2048
ds = dev.scene(syn.config3(size=n)[0])
buf = torch.empty((n, n, 3), dtype=torch.float32, device="cuda")
size = ft.ImageSize(n, n)
dev.set_option("guided", 0)
out = {}
for chunk in (64, 32, 16):
    for k in (0, 32):
        dev.set_option("chunk", chunk); dev.set_option("tail_k", k)
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr()); ds.collect_stats()
        for _ in range(3):
            ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr())
        st = ds.collect_stats()
        out[f"chunk {chunk}, tail_k {k}"] = {"kernel_ms": round(st["kernel_ms"] / 3, 3), "tail_fraction": round(st["tail_fraction"], 3), "lane_util": round(st["sdf_evals"] / 64 / st["wave_evals"], 3)}
print(json.dumps(out))
