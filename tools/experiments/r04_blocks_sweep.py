#!/usr/bin/env python3
"""round 4: kernel ms over frame size x resident workgroups per CU (FT_OPT_MAX_BLOCKS_PER_CU; 0 = the occupancy limit) for the Program.fs scene
(carved-union kernel) and C2 — what a launch should ask for when the frame has only a few tiles per resident wave."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn

dev = ft.Device(0)
dev.set_option("tail_k", int(os.environ.get("FT_TAIL_K", "2")))
cam = syn.default_camera()
for name, scene in (("Program.fs", syn.console_scene()[0]), ("C2", syn.config2()[0])):
    ds = dev.scene(scene)
    for n in (250, 500, 750, 1000, 1250, 1500, 2000):
        buf = torch.empty((n, n, 3), dtype=torch.float32, device="cuda")
        row = {}
        for mb in (0, 2, 3, 4, 5, 6):
            dev.set_option("max_blocks_per_cu", mb)
            best = 1e9
            for rep in range(3):
                ds.render_device(syn.EPSILON, syn.RAY_LENGTH, ft.ImageSize(n, n), cam, buf.data_ptr()); st = ds.collect_stats()
                best = min(best, st["kernel_ms"])
            row[mb] = round(best, 3)
        print(json.dumps({"scene": name, "size": n, "tiles": ((n + 7) // 8) ** 2, "kernel_ms_by_max_blocks": row}), flush=True)
        del buf
    ds.close()
