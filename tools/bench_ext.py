#!/usr/bin/env python3
"""EXTENSION configs of BASELINE.json (no reference counterpart): C2 with boxes + 1-bounce ambient occlusion,
C3 at 4 samples per pixel, C5 glass (4 bounces, 16 wavelength bins, 16 spp).  Kernel time with the frame left in HBM."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn

dev = ft.Device(0)
cam = syn.default_camera()
for name, scene, n, kw in (("EXT C2: union of 16 spheres + 16 boxes, diffuse + 8 AO rays (radius 1), 1024^2", syn.config2(boxes=True)[0], 1024, dict(ao_samples=8, ao_radius=1.0)),
                           ("EXT C2 at 4096^2", syn.config2(boxes=True)[0], 4096, dict(ao_samples=8, ao_radius=1.0)),
                           ("EXT C3: 256-sphere smooth union, 4096^2, 4 spp", syn.config3()[0], 4096, dict(spp=4)),
                           ("EXT C3: 4096^2, 1 spp + 4 AO rays (radius 0.5)", syn.config3()[0], 4096, dict(ao_samples=4, ao_radius=0.5)),
                           ("EXT C5: 6 glass + 10 solid objects, 2048^2, 16 spp, 4 bounces, 16 wavelength bins", syn.config5()[0], 2048, dict(spp=16, spectral=16, max_bounces=4)),
                           ("EXT C5 without wavelengths: 2048^2, 16 spp, 4 bounces", syn.config5()[0], 2048, dict(spp=16, max_bounces=4)),
                           ("EXT glass blob: one glass smooth union of 24 spheres, 2048^2, 4 spp, 4 bounces, 4 bins", None, 2048, dict(spp=4, spectral=4, max_bounces=4))):
    if scene is None:
        from fraytracer_amd import SdfForm, SdfObject, SdfMaterial, SdfScene
        rng = syn.Rng(31)
        kids = [SdfForm.Primitive.sphere(rng.pointInBall(2.5), rng.range(0.4, 0.9)) for _ in range(24)]
        scene = SdfScene(SdfObject.create(SdfMaterial.createGlass((0.95, 0.9, 0.8), 1.45, 0.03), SdfForm.unionSmooth(0.25, kids)),
                         syn.BACKGROUND, syn.program_lights())
    ds = dev.scene(scene)
    buf = torch.empty((n, n, 3), dtype=torch.float32, device="cuda")
    size = ft.ImageSize(n, n)
    ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr(), **kw); ds.collect_stats()
    reps = 2
    for _ in range(reps):
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, buf.data_ptr(), **kw)
    st = ds.collect_stats()
    rays = (st["rays_primary"] + st["rays_shadow"] + st["rays_ext"]) / reps
    ms = st["kernel_ms"] / reps
    print(json.dumps({"scene": name, "kernel_ms": round(ms, 3), "Mrays/s": round(rays / ms / 1e3, 1), "primary": st["rays_primary"] // reps,
                      "shadow": st["rays_shadow"] // reps, "ext_rays": st["rays_ext"] // reps,
                      "lane_util": round(st["sdf_evals"] / (64.0 * st["wave_evals"]), 4), "culled_fraction": round(st["culled_fraction"], 4),
                      "cull_site": ds.info().get("cull_pc")}), flush=True)
