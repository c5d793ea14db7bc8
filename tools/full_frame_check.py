#!/usr/bin/env python3
"""Whole frames at BASELINE.json's full sizes, HIP path (through the C ABI) against the CPU oracle, float for float.
One JSON line per config.  Usage: python tools/full_frame_check.py [name ...]   (default: all)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
from oracle import binding as ob


def host_cpus():
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(int(q) / int(p)))
    except Exception:
        pass
    return os.cpu_count() or 1


CASES = {
    "c1": ("C1: one sphere, 256^2", lambda: syn.config1()[0], 256, {}),
    "c2": ("C2: SdfObject.union of 16 spheres + 16 capsules, 4096^2", lambda: syn.config2()[0], 4096, {}),
    "console": ("Program.fs scene: System.Random(19), 1000 tori, 2 lights, 4000^2", lambda: syn.console_scene()[0], 4000, {}),
    "mixed": ("nested unions / subtract / intersect / smooth, all primitive types, 2048^2", lambda: syn.mixed_nested()[0], 2048, {}),
    "zoo": ("combinator zoo, 2048^2", lambda: syn.combinator_zoo()[0], 2048, {}),
    "c3": ("C3: unionSmooth of 256 spheres, 4096^2", lambda: syn.config3()[0], 4096, {}),
    "ext_c2_ao": ("EXTENSION C2: 16 spheres + 16 boxes, 8 AO rays, 1024^2", lambda: syn.config2(boxes=True)[0], 1024, dict(ao_samples=8, ao_radius=1.0)),
    "ext_c5": ("EXTENSION C5: glass, 2048^2, 16 spp, 4 bounces, 16 wavelength bins", lambda: syn.config5()[0], 2048,
               dict(spp=16, spectral=16, max_bounces=4)),
}

cam = syn.default_camera()
dev = ft.Device(0)
threads = host_cpus()
for key in (sys.argv[1:] or CASES):
    title, make, n, kw = CASES[key]
    scene = make()
    g, st = dev.scene(scene).render(syn.EPSILON, syn.RAY_LENGTH, ft.ImageSize(n, n), cam, **kw)
    t0 = time.perf_counter()
    o, cnt = ob.Oracle().scene(scene).render(syn.EPSILON, syn.RAY_LENGTH, n, n, cam.as_array(), nthreads=threads, **kw)
    dt = time.perf_counter() - t0
    diff = g.view(np.uint32) != o.view(np.uint32)
    rays_g = st["rays_primary"] + st["rays_shadow"] + st["rays_ext"]
    rays_o = cnt["rays_primary"] + cnt["rays_shadow"] + cnt["rays_ext"]
    print(json.dumps({"config": title, "size": n, "floats": int(g.size), "differing_floats": int(diff.sum()),
                      "max_abs_delta": float(np.abs(g - o).max()), "rays_gpu": rays_g, "rays_oracle": rays_o,
                      "flags_gpu": st["flags"], "flags_oracle": cnt["flags"],
                      "oracle_seconds": round(dt, 1), "oracle_threads": threads, "oracle_Mrays/s": round(rays_o / dt / 1e6, 3),
                      "gpu_kernel_ms": round(st["kernel_ms"], 2), "gpu_Mrays/s": round(rays_g / st["kernel_ms"] / 1e3, 1)}), flush=True)
    del g, o
