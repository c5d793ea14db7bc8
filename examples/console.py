#!/usr/bin/env python3
"""The reference's console program (src/FrayTracer.Console/Program.fs) on the MI355X path:
System.Random(19) scene of 1000 tori, 1000x1000, epsilon 0.01, ray length 30, timing line, result.bmp.

    python examples/console.py [--size 1000] [--tori 1000] [--out result.bmp] [--math glibc]

--math glibc: MathF.Pow of the tone map (and MathF.Exp / Log of any unionSmooth) as this host's C runtime computes them — what the reference's
CPU path returns on this machine under .NET (FT_OPT_MATH); the default is the library's fixed arithmetic.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
from fraytracer_amd.postprocess import saveBitmap

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1000)
ap.add_argument("--tori", type=int, default=1000)
ap.add_argument("--out", default="result.bmp")
ap.add_argument("--math", choices=("fixed", "glibc"), default="fixed")
args = ap.parse_args()
if args.math == "glibc":
    ft.Device.default(0).set_option("math", ft.glibc_build_of_this_host())

scene, _ = syn.console_scene(seed=19, n=args.tori, size=args.size)        # Program.fs:14-83
camera = syn.default_camera()                                              # Program.fs:16-22
imageSize = ft.ImageSize(args.size, args.size)
epsilon = 0.01                                                             # Program.fs:85

print("Rendering...")                                                      # Program.fs:87
t0 = time.perf_counter()
traced = ft.Image.render(epsilon, 30.0, imageSize, camera, ft.SdfScene.trace(scene))   # Program.fs:90-93
print(f"Time = {time.perf_counter() - t0:.2f} sec")                        # Program.fs:96 (includes scene upload)

# Program.fs:98-100: Image.toColors 2.2f rng |> Image.saveBitmap.  The tone map runs on the GPU (ft_tone_map_host); the reference draws its
# dithering noise from the scene's generator shared by a parallel map (racy): here it is a counter-based hash seeded with 19.
saveBitmap(args.out, ft.Image.toColors(2.2, 19, traced))
print("wrote", args.out)
