import json, os, sys
sys.path.insert(0, ".")
import numpy as np, torch
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
dev = ft.Device(0); cam = syn.default_camera()
out = {}
for name, scene in (("program_fs", syn.console_scene()[0]), ("c3", syn.config3()[0])):
    ds = dev.scene(scene)
    # one 8x8 tile in the middle of a 1000^2 frame: x0 = 496, 8 columns, but all rows... use a 8-column strip of height 8 via a small image is different; take the centre tile of a 1000-wide, 8-high image
    for k in (0, 64):
        dev.set_option("tail_k", k)
        W, H = 1000, 1000
        buf = torch.empty((8, H, 3), dtype=torch.float32, device="cuda")
        kw = dict(x0=496, n_columns=8)
        size = ft.ImageSize(W, H)
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr(), **kw); ds.collect_stats()
        for _ in range(5): ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr(), **kw)
        st = ds.collect_stats()
        tiles = H // 8
        out[f"{name} tail_k={k}"] = {"ms": round(st["kernel_ms"]/5, 4), "tiles(=waves)": tiles, "rounds_per_tile": round(st["wave_evals"]/5/tiles, 1), "evals_per_tile": round(st["sdf_evals"]/5/tiles, 1), "mhz": round(st["shader_mhz"]), "tail_fraction": round(st["tail_fraction"], 3)}
print(json.dumps(out))
