#!/usr/bin/env python3
"""Mrays/s of every BASELINE.json config that fits one GPU, plus the PCIe-inclusive rate of ft_render.
Not the contract bench (that is bench.py); used to track the non-headline kernels."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn

dev = ft.Device(0)
from _opts import apply_env_options
applied = apply_env_options(dev)
cam = syn.default_camera()
cases = [("C1 sphere 256^2", syn.config1()[0], 256), ("C2 union32 1024^2", syn.config2()[0], 1024),
         ("C2 union32+boxes 1024^2", syn.config2(boxes=True)[0], 1024),
         ("console-like 1000 tori 1000^2", syn.console_like(n=1000)[0], 1000),
         ("console-like 1000 tori 4000^2", syn.console_like(n=1000)[0], 4000),
         # the reference's own workload (Program.fs:14-83: System.Random(19), 1000 tori, subtract(intersect(union)), 2 lights) at its own size and at 4000^2
         ("Program.fs scene 1000^2", syn.console_scene()[0], 1000), ("Program.fs scene 4000^2", syn.console_scene()[0], 4000),
         ("C2 union32 4096^2", syn.config2()[0], 4096),
         ("mixed nested 1024^2", syn.mixed_nested()[0], 1024),
         ("crowd of 300 combinator objects 2048^2", syn.combinator_crowd()[0], 2048),
         ("C3 smooth256 4096^2", syn.config3()[0], 4096), ("C4 smooth256 8192^2", syn.config3()[0], 8192),
         # EXTENSION (BASELINE.json config 5): glass paths
         ("C5 glass 2048^2 16spp", syn.config5()[0], 2048, dict(spp=16, spectral=16, max_bounces=4))]
only = sys.argv[1:] 
for name, scene, n, *rest in cases:
    kw = rest[0] if rest else {}
    if only and not any(o in name for o in only):
        continue
    t0 = time.perf_counter()
    ds = dev.scene(scene)
    t_build = time.perf_counter() - t0
    buf = torch.empty((n, n, 3), dtype=torch.float32, device="cuda")
    size = ft.ImageSize(n, n)
    ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr(), **kw); ds.collect_stats()
    reps = 3
    for _ in range(reps):
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr(), **kw)
    st = ds.collect_stats()
    rays = (st["rays_primary"] + st["rays_shadow"] + st["rays_ext"]) / reps
    ms = st["kernel_ms"] / reps
    t_host = float("nan")
    if not os.environ.get("FT_KERNEL_ONLY"):                                # (profile_scene.sh: only whole-frame launches in the trace)
        img, st2 = ds.render(syn.EPSILON, syn.RAY_LENGTH, size, cam, **kw)      # first call allocates staging buffers
        t0 = time.perf_counter()
        img, st2 = ds.render(syn.EPSILON, syn.RAY_LENGTH, size, cam, **kw)      # host output: includes the device->host copy
        t_host = time.perf_counter() - t0
    print(json.dumps({"scene": name, "kernel_ms": round(ms, 3), "Mrays/s": round(rays / ms / 1e3, 2),
                      "rays": int(rays), "evals_per_ray": round(st["sdf_evals"] / reps / rays, 2),
                      "lane_util": round(st["sdf_evals"] / (64.0 * st["wave_evals"]), 4), "culled_fraction": round(st["culled_fraction"], 4),
                      "host_output_ms": round(t_host * 1e3, 2), "Mrays/s_incl_pcie": round(rays / t_host / 1e6, 2),
                      "scene_build_s": round(t_build, 3), "info": ds.info()}), flush=True)
    del buf
