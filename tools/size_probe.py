#!/usr/bin/env python3
"""Kernel time of one scene against the frame size, throughput build and latency build of the general kernel (FT_OPT_WALK 0 / 1): a small frame
takes as long as its longest tile.  Usage: size_probe.py [program_fs|c2|console_like]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
dev = ft.Device(0); cam = syn.default_camera()
which = sys.argv[1] if len(sys.argv) > 1 else "program_fs"
ds = dev.scene({"program_fs": syn.console_scene, "c2": syn.config2, "console_like": syn.console_like}[which]()[0])
out = {"scene": which}
for walk in (0, 1):
  dev.set_option("walk", walk)
  for n in (64, 250, 500, 750, 1000, 1250, 1500, 2000, 3000, 4000):
    buf = torch.empty((n, n, 3), dtype=torch.float32, device="cuda")
    size = ft.ImageSize(n, n)
    ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr()); ds.collect_stats()
    for _ in range(5): ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr())
    st = ds.collect_stats()
    out[f"walk {walk}, {n}^2"] = {"ms": round(st["kernel_ms"]/5, 3), "rounds_per_tile": round(st["wave_evals"]/5/((n+7)//8)**2, 1), "lane_util": round(st["sdf_evals"]/64/st["wave_evals"], 3), "mhz": round(st["shader_mhz"])}
    del buf
print(json.dumps(out))
