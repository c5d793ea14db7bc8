#!/usr/bin/env python3
"""Differential fuzzing aimed at the lean kernel's child culling: random smooth unions of spheres (32 - 400 children, strengths 0.03 - 1.5, radii and
spreads over two decades) seen by random cameras — far away, close by, inside the cloud — on wide, low frames (a fine pixel pitch keeps the rays of a
wave together, which is when children are dropped).  HIP path against the CPU oracle, float for float and counter for counter, with the pass on; the
share of dropped (child, ray) pairs is reported.  `wide`: scene extents from 0.1 to 3000 and strengths from 0.03 to 100 (the slack of the pass's bounds is absolute: large coordinates and weak
unions are where it is smallest relative to the arithmetic's own rounding).  `nested` (round 4): the smooth union sits inside a general scene — in a union beside
other objects, under an intersect or a subtract, behind children of another kind (its sphere run then continues an accumulator) — so the pass runs in the
general kernels (FtSceneDev.cullPc).  Usage: python tools/fuzz_cull.py [first_seed] [count] [wide|nested]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
from fraytracer_amd import SdfForm, SdfObject, SdfMaterial, SdfScene, SdfLight, Camera, Lens
from oracle import binding as ob

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
nested = len(sys.argv) > 3 and sys.argv[3] == "nested"
dev = ft.Device(0)
from _opts import apply_env_options
applied = apply_env_options(dev)
if "math" in applied and applied["math"] != 0:
    ob.lib.orc_set_libm(1)
bad, culled, rays, lean = [], [], 0, 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(32, 401))
    spread = float(10.0 ** (rng.uniform(-1.0, 3.5) if wide else rng.uniform(-0.5, 1.0)))
    strength = float(10.0 ** rng.uniform(-1.5, 0.18)) * (spread if wide and rng.random() < 0.7 else 1.0)
    strength = min(strength, 100.0)
    rmax = float(spread * 10.0 ** rng.uniform(-1.5, -0.5))
    c = rng.normal(size=(n, 3)) * spread * 0.5
    forms = [SdfForm.Primitive.sphere(tuple(float(v) for v in c[i]), float(rng.uniform(0.2, 1.0) * rmax)) for i in range(n)]
    lights = [SdfLight.directional(tuple(float(v) for v in rng.normal(size=3)), (0.5, 0.5, 0.5))]
    if rng.random() < 0.4:
        lights.append(SdfLight.point(tuple(float(v) for v in rng.normal(size=3) * spread), (3.0, 2.0, 1.0)))
    root = SdfObject.create(SdfMaterial.createSolid((0.9, 0.6, 0.3)), SdfForm.unionSmooth(strength, forms))
    if nested:
        P, k = SdfForm.Primitive, int(rng.integers(0, 4))
        pt = lambda s=0.5: tuple(float(v) for v in rng.normal(size=3) * spread * s)
        m2 = SdfMaterial.createSolid((0.2, 0.5, 0.9))
        if k == 0: root = SdfObject.union([root, SdfObject.create(m2, P.torus(pt(), (0.0, 1.0, 0.0), spread * 0.4, spread * 0.05)), SdfObject.create(m2, P.capsule(pt(), pt(), spread * 0.08))])
        elif k == 1: root = SdfObject.intersect(root, [P.sphere(pt(0.1), spread * float(rng.uniform(0.5, 1.2)))])
        elif k == 2: root = SdfObject.subtract(root, P.sphere(pt(0.4), spread * float(rng.uniform(0.2, 0.6))))
        else: root = SdfObject.create(SdfMaterial.createSolid((0.9, 0.6, 0.3)), SdfForm.unionSmooth(strength, [P.capsule(pt(), pt(), spread * 0.05), P.torus(pt(), (1.0, 0.0, 0.0), spread * 0.3, spread * 0.04)] + forms))
    scene = SdfScene(root, syn.BACKGROUND, lights)
    dist = spread * float(10.0 ** rng.uniform(-0.7, 0.8))          # inside the cloud ... far outside
    pos = rng.normal(size=3); pos = pos / np.linalg.norm(pos) * dist
    cam = Camera.lookAt(Position=tuple(float(v) for v in pos), LookAt=tuple(float(v) for v in rng.normal(size=3) * spread * 0.2), Up=(0.0, 1.0, 0.0),
                        Lens=Lens.create(float(rng.uniform(20.0, 90.0))))
    W, H = int(rng.choice([1024, 2048, 4096])), int(rng.choice([8, 16, 24]))
    eps = float(10.0 ** rng.uniform(-3.0, -1.5)) * (max(1.0, spread / 3.0) if wide else 1.0)      # (a step count that stays bounded at large extents)
    length = float(spread * rng.uniform(2.0, 40.0))
    ds = dev.scene(scene)
    lean += ds.info()["fast_path"] == 1
    g, st = ds.render(eps, length, ft.ImageSize(W, H), cam)
    o, cnt = ob.Oracle().scene(scene).render(eps, length, W, H, cam.as_array(), nthreads=16)
    same = np.array_equal(g.view(np.uint32), o.view(np.uint32))
    keys = ("rays_primary", "rays_shadow", "hits_primary", "hits_shadow", "flags")
    if not same or any(st[k] != cnt[k] for k in keys):
        bad.append((seed, int((g.view(np.uint32) != o.view(np.uint32)).sum()), {k: (st[k], cnt[k]) for k in keys if st[k] != cnt[k]}))
    culled.append(st["culled_fraction"]); rays += st["rays_primary"] + st["rays_shadow"]
    if (seed - first + 1) % (5 if wide else 25) == 0:
        print(f"... {seed - first + 1} scenes, {len(bad)} mismatches, {time.time() - t0:.0f} s", flush=True)
cf = np.array(culled)
print(json.dumps({"options": applied, "build": ft.build_info()["src"], "first_seed": first, "scenes": count, "wide": wide, "lean_scenes": int(lean), "rays": int(rays),
                  "culled_fraction": {"mean": round(float(cf.mean()), 4), "median": round(float(np.median(cf)), 4), "max": round(float(cf.max()), 4),
                                      "scenes_above_10_percent": int((cf > 0.1).sum())},
                  "mismatching_scenes": len(bad), "mismatches": bad[:20], "seconds": round(time.time() - t0, 1)}))
