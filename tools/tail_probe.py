#!/usr/bin/env python3
"""Kernel time of ONE rank's share of the C3 frame when it is split over N ranks (interleaved 16-column
stripes), on one GPU: shows the tail / launch overhead strong scaling pays.  FT_MAX_BLOCKS_PER_CU caps occupancy.
Usage: python tools/tail_probe.py [size]   (4096 = the metric's config, 8192 = BASELINE.json config 4)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn, distributed as ftd

dev = ft.Device(0)
from _opts import apply_env_options
applied = apply_env_options(dev)
cam = syn.default_camera()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ds = dev.scene(syn.config3(size=W)[0])
size = ft.ImageSize(W, W)
out = {}
for N in (1, 2, 4, 8):
    cols = W // N
    buf = torch.empty((cols, W, 3), dtype=torch.float32, device="cuda")
    kw = ftd.tiling(W, N, N // 2, 16)
    ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr(), **kw); ds.collect_stats()
    reps = 4
    for _ in range(reps):
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr(), **kw)
    st = ds.collect_stats()
    out[N] = round(st["kernel_ms"] / reps, 3)
print(json.dumps({"options": applied, "size": W, "build": ft.build_info()["src"], "kernel_ms_per_rank_share": out,
                  "efficiency_vs_N1": {n: round(out[1] / (n * out[n]), 3) for n in out}}))
