#!/usr/bin/env python3
"""Differential fuzzing: random scenes (fraytracer_amd.synthetic.fuzz_scene) rendered by the HIP path and by the
CPU oracle must agree float for float and ray for ray.  Usage: python tools/fuzz_parity.py [first_seed] [count] [big|edge]
(edge: fuzz_scene_edge — scenes up to 7000 from the origin, epsilon down to 1e-5, ray length up to 1000: the edge of the escape shortcut's drift bound)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
from oracle import binding as ob

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
big = len(sys.argv) > 3 and sys.argv[3] == "big"          # unions of 60-400 objects, larger images
edge = len(sys.argv) > 3 and sys.argv[3] == "edge"
dev = ft.Device(0)
from _opts import apply_env_options
applied = apply_env_options(dev)                    # FT_TAIL_K=64: every evaluation through the latency mode; FT_MATH=1|2: glibc arithmetic ...
if "math" in applied and applied["math"] != 0:
    ob.lib.orc_set_libm(1)                          # ... against the oracle calling this machine's expf / logf (FT_MATH must name the build libm resolves to)
    assert applied["math"] == ft.glibc_build_of_this_host(), "FT_MATH must be the glibc build of this host"
bad, skipped, flagged, rays, glassy = [], 0, 0, 0, 0
t0 = time.time()
for seed in range(first, first + count):
    length = syn.RAY_LENGTH
    if edge: scene, cam, size, eps, length, ext = syn.fuzz_scene_edge(seed)
    else: scene, cam, size, eps, ext = syn.fuzz_scene(seed, big)
    try:
        ds = dev.scene(scene)
    except ft.FrayTracerError as e:                  # e.g. a union cell without a candidate: rejected at build time
        try:
            ob.Oracle().scene(scene)
            bad.append((seed, "device rejects, oracle accepts: " + str(e)))
        except ob.OracleError:
            skipped += 1
        continue
    g, st = ds.render(eps, length, size, cam, **ext)
    o, cnt = ob.Oracle().scene(scene).render(eps, length, size.X, size.Y, cam.as_array(), nthreads=8, **ext)
    same = np.array_equal(g.view(np.uint32), o.view(np.uint32))
    keys = ("rays_primary", "rays_shadow", "rays_ext", "hits_primary", "hits_shadow", "flags")
    if not same or any(st[k] != cnt[k] for k in keys):
        bad.append((seed, int((g.view(np.uint32) != o.view(np.uint32)).sum()), {k: (st[k], cnt[k]) for k in keys if st[k] != cnt[k]}, ext))
    flagged += st["flags"] != 0
    glassy += st["rays_ext"] > 0
    rays += st["rays_primary"] + st["rays_shadow"] + st["rays_ext"]
    ds.close()
    if (seed - first) % 50 == 49:
        print(f"... {seed - first + 1} scenes, {len(bad)} mismatches, {time.time() - t0:.0f} s", flush=True)
print(json.dumps({"options": applied, "build": ft.build_info()["src"], "big": big, "edge": edge, "first_seed": first, "scenes": count, "rejected_by_both": skipped, "with_nan_or_cap_flags": int(flagged),
                  "with_extension_rays": int(glassy), "rays": int(rays), "mismatching_scenes": len(bad), "mismatches": bad[:20],
                  "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
