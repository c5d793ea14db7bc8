#!/usr/bin/env python3
"""Stress of the host-output path of ft_render / ft_render_colors (page-locking during the render, column chunks on two lanes,
copy stream): many frames of changing sizes into fresh, reused and pre-registered destinations, each compared with the first
render of the same (scene, size).  One process, one GPU.  Usage: python tools/host_output_stress.py [iterations]"""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = ft.Device(0)
cam = syn.default_camera()
scenes = {"lean64": dev.scene(syn.config3(n=64, size=1024)[0]), "union32": dev.scene(syn.config2()[0]),
          "tori150": dev.scene(syn.console_like(n=150)[0])}
rng = np.random.default_rng(123)
seen, kept, bad = {}, [], 0
t0 = time.time()
for it in range(n_iter):
    name = list(scenes)[int(rng.integers(len(scenes)))]
    w = int(rng.choice([8, 40, 64, 96, 200, 256, 512, 1000, 1024, 1536, 2048]))
    h = int(rng.choice([8, 33, 64, 128, 500, 1024, 2048]))
    mode = int(rng.integers(4))
    out = None
    if mode == 1:                                   # reused pageable destination
        out = next((a for a in kept if a.shape == (w, h, 3)), None)
        if out is None:
            out = np.empty((w, h, 3), np.float32); kept.append(out); kept[:] = kept[-8:]
    elif mode == 2:                                 # destination the caller page-locked
        out = np.empty((w, h, 3), np.float32); dev.host_register(out)
    if mode == 3:                                   # bytes through the device tone map
        img = scenes[name].render_colors(syn.EPSILON, syn.RAY_LENGTH, ft.ImageSize(w, h), cam, gamma=2.2, seed=7)
        img = img[0] if isinstance(img, tuple) else img
    else:
        img, st = scenes[name].render(syn.EPSILON, syn.RAY_LENGTH, ft.ImageSize(w, h), cam, out=out)
    key = (name, w, h, mode == 3)
    dig = hashlib.sha1(np.ascontiguousarray(img).tobytes()).hexdigest()
    if seen.setdefault(key, dig) != dig:
        bad += 1
        print(json.dumps({"mismatch": key, "iteration": it}), flush=True)
    if mode == 2:
        dev.host_unregister(out)
    if it % 50 == 49:
        print(f"... {it + 1} renders, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(json.dumps({"renders": n_iter, "distinct_configs": len(seen), "mismatches": bad, "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
