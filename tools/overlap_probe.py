#!/usr/bin/env python3
"""Frames of one rank's share (1/N of the C3 4096^2 frame) back to back: one stream vs two contexts on two streams
(frame k+1 starts filling the GPU while the stragglers of frame k drain).  Wall time per frame, one GPU."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn, distributed as ftd

W = 4096
cam = syn.default_camera()
scene = syn.config3(size=W)[0]
size = ft.ImageSize(W, W)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
devs = [ft.Device(0), ft.Device(0)]
for d, s in zip(devs, streams):
    d.set_stream(s.cuda_stream)
dss = [d.scene(scene) for d in devs]
K = 12
res = {}
for N in (1, 2, 4, 8):
    kw = ftd.tiling(W, N, N // 2, 16)
    bufs = [torch.empty((W // N, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
    for mode in ("one_stream", "two_streams"):
        for k in range(2):
            dss[k].render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, bufs[k].data_ptr(), **kw)
        torch.cuda.synchronize()
        for d in dss: d.collect_stats()
        t0 = time.perf_counter()
        for k in range(K):
            i = (k & 1) if mode == "two_streams" else 0
            dss[i].render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, bufs[i].data_ptr(), **kw)
        torch.cuda.synchronize()
        res[f"N={N} {mode}"] = round((time.perf_counter() - t0) / K * 1e3, 3)
        for d in dss: d.collect_stats()
print(json.dumps(res))
