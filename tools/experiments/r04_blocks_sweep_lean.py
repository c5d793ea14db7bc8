#!/usr/bin/env python3
"""round 4: the same sweep as r04_blocks_sweep.py for the lean kernel (C3 scene): kernel ms over frame size x resident workgroups per CU."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn

dev = ft.Device(0)
cam = syn.default_camera()
ds = dev.scene(syn.config3()[0])
for n in (256, 512, 768, 1024, 1448, 2048, 2896):
    buf = torch.empty((n, n, 3), dtype=torch.float32, device="cuda")
    row = {}
    for mb in (0, 1, 2, 3, 4, 5):
        dev.set_option("max_blocks_per_cu", mb)
        best = 1e9
        for rep in range(3):
            ds.render_device(syn.EPSILON, syn.RAY_LENGTH, ft.ImageSize(n, n), cam, buf.data_ptr()); st = ds.collect_stats()
            best = min(best, st["kernel_ms"])
        row[mb] = round(best, 3)
    print(json.dumps({"scene": "C3", "size": n, "tiles": ((n + 7) // 8) ** 2, "kernel_ms_by_max_blocks": row}), flush=True)
    del buf
ds.close()
