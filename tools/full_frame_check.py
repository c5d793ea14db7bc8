#!/usr/bin/env python3
"""One-off: the whole C3 4096x4096 frame (16.7 M pixels) on the GPU against the CPU oracle."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
from oracle import binding as ob

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scene, _ = syn.config3()
cam = syn.default_camera()
dev = ft.Device(0)
g, st = dev.scene(scene).render(syn.EPSILON, syn.RAY_LENGTH, ft.ImageSize(n, n), cam)
t0 = time.perf_counter()
o, cnt = ob.Oracle().scene(scene).render(syn.EPSILON, syn.RAY_LENGTH, n, n, cam.as_array(), nthreads=os.cpu_count())
dt = time.perf_counter() - t0
diff = g.view(np.uint32) != o.view(np.uint32)
print(json.dumps({"size": n, "pixels": n * n, "differing_floats": int(diff.sum()), "max_abs_delta": float(np.abs(g - o).max()),
                  "rays_shadow_gpu": st["rays_shadow"], "rays_shadow_oracle": cnt["rays_shadow"], "oracle_seconds": round(dt, 1),
                  "oracle_threads": os.cpu_count(), "oracle_Mrays/s": round((cnt["rays_primary"] + cnt["rays_shadow"]) / dt / 1e6, 3),
                  "gpu_kernel_ms": round(st["kernel_ms"], 2)}))
