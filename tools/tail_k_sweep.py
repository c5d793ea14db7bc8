#!/usr/bin/env python3
"""Latency-mode threshold sweep (FT_OPT_TAIL_K): kernel ms per scene for several k; k = 0 is the one-ray-per-lane path only.
Usage: tail_k_sweep.py [k ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn, distributed as ftd

ks = [int(a) for a in sys.argv[1:]] or [0, 2, 4, 8, 12, 16]
dev = ft.Device(0)
cam = syn.default_camera()
c3 = syn.config3()[0]
cases = [("C3 4096^2", c3, 4096, {}), ("C3 4096^2, one rank's share at N=8", c3, 4096, ftd.tiling(4096, 8, 4, 16)),
         ("console-like 1000 tori 1000^2", syn.console_like(n=1000)[0], 1000, {}), ("console-like 1000 tori 4000^2", syn.console_like(n=1000)[0], 4000, {}),
         ("C2 union32 4096^2", syn.config2()[0], 4096, {}), ("crowd of 300 combinators 2048^2", syn.combinator_crowd()[0], 2048, {}),
         ("C5 glass 2048^2 16spp", syn.config5()[0], 2048, dict(spp=16, spectral=16, max_bounces=4))]
only = os.environ.get("FT_SWEEP_ONLY")
for name, scene, n, kw in cases:
    if only and only not in name:
        continue
    ds = dev.scene(scene)
    size = ft.ImageSize(n, n)
    cols = kw.get("n_columns", n) if "stripe_ranks" not in kw else n // kw["stripe_ranks"]
    buf = torch.empty((cols, n, 3), dtype=torch.float32, device="cuda")
    row = {"scene": name}
    for k in ks:
        dev.set_option("tail_k", abs(k)); dev.set_option("guided", 0 if k < 0 else 1)          # a negative k: that threshold without the guided hand-out
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr(), **kw); ds.collect_stats()
        reps = 4
        for _ in range(reps):
            ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr(), **kw)
        st = ds.collect_stats()
        row[f"k={k}"] = round(st["kernel_ms"] / reps, 3)
        row[f"tail_fraction k={k}"] = round(st["tail_fraction"], 4)
    row["shader_mhz"] = round(st["shader_mhz"], 1)
    print(json.dumps(row), flush=True)
    del buf
