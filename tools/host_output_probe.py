#!/usr/bin/env python3
"""ft_render (host output) on C3 4096^2: wall time per frame for a pageable destination (pinned inside the call), a destination
registered once with ft_host_register, and the old path (FT_HOST_NO_PIN=1 FT_HOST_CHUNKS=1), next to the kernel-only time.
One JSON line per variant."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import fraytracer_amd as ft
    from fraytracer_amd import synthetic as syn
    tag, registered, cold = sys.argv[2], sys.argv[3] == "1", sys.argv[3] == "cold"
    size = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
    scene, _ = syn.config3(size=size)
    cam = syn.default_camera()
    dev = ft.Device(0)
    from _opts import apply_env_options
    applied = apply_env_options(dev)
    ds = dev.scene(scene)
    S = ft.ImageSize(size, size)
    out = np.zeros((size, size, 3), np.float32)             # touched pages, like Array2D.zeroCreate
    if registered:
        dev.host_register(out)
    ref, _ = ds.render(syn.EPSILON, syn.RAY_LENGTH, S, cam)
    ds.render(syn.EPSILON, syn.RAY_LENGTH, S, cam, out=out)
    ts, kms = [], []
    keep = []
    for _ in range(6):
        if cold:                                            # a fresh, untouched destination per frame (what a new FColor[,] is)
            keep.append(out)
            out = np.empty((size, size, 3), np.float32)
        t0 = time.perf_counter()
        _, st = ds.render(syn.EPSILON, syn.RAY_LENGTH, S, cam, out=out)
        ts.append((time.perf_counter() - t0) * 1e3); kms.append(st["kernel_ms"])
    rays = st["rays_primary"] + st["rays_shadow"]
    print(json.dumps({"variant": tag, "ms_per_frame_median": round(float(np.median(ts)), 2), "ms_min": round(min(ts), 2),
                      "Mrays_per_s": round(rays / np.median(ts) / 1e3, 1), "sum_of_kernel_ms": round(float(np.median(kms)), 2),
                      "identical_to_plain_render": bool(np.array_equal(out.view(np.uint32), ref.view(np.uint32))), "shader_mhz": round(st["shader_mhz"], 1)}), flush=True)
    if registered:
        dev.host_unregister(out)
    sys.exit(0)

size = sys.argv[1] if len(sys.argv) > 1 else "4096"
for tag, env, reg in (("old path: pageable, one copy after the kernel", {"FT_HOST_NO_PIN": "1", "FT_HOST_CHUNKS": "1"}, "0"),
                      ("pageable destination, pinned inside the call, 4 chunks", {}, "0"),
                      ("pageable destination, pinned inside the call, 1 chunk", {"FT_HOST_CHUNKS": "1"}, "0"),
                      ("pageable destination, pinned inside the call, 2 chunks", {"FT_HOST_CHUNKS": "2"}, "0"),
                      ("pageable destination, pinned inside the call, 8 chunks", {"FT_HOST_CHUNKS": "8"}, "0"),
                      ("fresh untouched destination per frame, old path", {"FT_HOST_NO_PIN": "1", "FT_HOST_CHUNKS": "1"}, "cold"),
                      ("fresh untouched destination per frame, pinned inside the call, 4 chunks", {}, "cold"),
                      ("fresh untouched destination per frame, not pinned, 4 chunks", {"FT_HOST_NO_PIN": "1"}, "cold"),
                      ("destination registered once (ft_host_register), 4 chunks", {}, "1"),
                      ("destination registered once, 8 chunks", {"FT_HOST_CHUNKS": "8"}, "1"),
                      ("destination registered once, 1 chunk", {"FT_HOST_CHUNKS": "1"}, "1")):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", tag, reg, size], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    print(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else f"FAILED {tag}: {r.stderr[-800:]}", flush=True)
