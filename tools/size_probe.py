#!/usr/bin/env python3
"""Kernel time of one scene against the frame size: a small frame takes as long as its longest tile (the Program.fs scene: 1.28 ms for a 64 x 64
frame of 64 tiles, 2.2 ms for 1000 x 1000).  Usage: size_probe.py [program_fs|c2|console_like|c3]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
from _opts import apply_env_options
dev = ft.Device(0); cam = syn.default_camera()
applied = apply_env_options(dev)
which = sys.argv[1] if len(sys.argv) > 1 else "program_fs"
ds = dev.scene({"program_fs": syn.console_scene, "c2": syn.config2, "console_like": syn.console_like, "c3": syn.config3}[which]()[0])
out = {"scene": which, "options": applied}
for n in (64, 250, 500, 750, 1000, 1250, 1500, 2000, 3000, 4000):
    buf = torch.empty((n, n, 3), dtype=torch.float32, device="cuda")
    size = ft.ImageSize(n, n)
    ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr()); ds.collect_stats()
    for _ in range(5): ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr())
    st = ds.collect_stats()
    out[f"{n}^2"] = {"ms": round(st["kernel_ms"] / 5, 3), "rounds_per_tile": round(st["wave_evals"] / 5 / ((n + 7) // 8) ** 2, 1),
                     "lane_util": round(st["sdf_evals"] / 64 / st["wave_evals"], 3), "tail_fraction": round(st["tail_fraction"], 4), "mhz": round(st["shader_mhz"])}
    del buf
print(json.dumps(out))
