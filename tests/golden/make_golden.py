#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  The reference ships no fixtures and cannot run here (no .NET), so
these vectors are OUTPUTS OF THE CPU ORACLE (oracle/ft_oracle.cpp) on the synthetic configs: they pin
the oracle against regressions and give the GPU tests fixed data — they are NOT outputs of the F#
program ("parity unpinned").  Each file holds the float32 image [X, Y, 3], the exact ray /
hit counters and the render parameters.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from fraytracer_amd import synthetic as syn   # noqa: E402
from oracle import binding as ob               # noqa: E402

CASES = {
    "c1_sphere_64": (lambda: syn.config1()[0], 64, 64),
    "c2_union32_64": (lambda: syn.config2()[0], 64, 64),
    "c2_union32_boxes_64": (lambda: syn.config2(boxes=True)[0], 64, 64),
    "c3_smooth256_48": (lambda: syn.config3()[0], 48, 48),
    "console_like_300_64": (lambda: syn.console_like(n=300)[0], 64, 64),
    "mixed_nested_48x40": (lambda: syn.mixed_nested()[0], 48, 40),
    "combinator_zoo_56": (lambda: syn.combinator_zoo()[0], 56, 56),
    # EXTENSIONS (no reference counterpart at all): ambient occlusion + 4 spp; glass paths with wavelength bins
    "ext_c2_boxes_ao8_spp4_48": (lambda: syn.config2(boxes=True)[0], 48, 48, dict(spp=4, ao_samples=8, ao_radius=0.75)),
    "ext_c5_glass_spp4_bins4_56": (lambda: syn.config5(size=56)[0], 56, 56, dict(spp=4, spectral=4, max_bounces=4)),
}


def ext_params(name):
    return CASES[name][3] if len(CASES[name]) > 3 else {}


def render(name):
    make, W, H = CASES[name][:3]
    cam = syn.default_camera().as_array()
    img, cnt = ob.Oracle().scene(make()).render(syn.EPSILON, syn.RAY_LENGTH, W, H, cam, nthreads=4, **ext_params(name))
    keys = ("rays_primary", "rays_shadow", "hits_primary", "hits_shadow", "flags") + (("rays_ext",) if ext_params(name) else ())
    counts = np.array([cnt[k] for k in keys], np.int64)
    return img, counts, np.array([W, H], np.int32)


# Tone map fixture: the oracle's Image.toColors / toBitmap bytes of two golden images, plain and with noise
TONEMAP = {"tonemap_console_like_300_64": "console_like_300_64", "tonemap_c3_smooth256_48": "c3_smooth256_48"}


def tonemap(name):
    img = np.load(os.path.join(HERE, TONEMAP[name] + ".npz"), allow_pickle=False)["image"]
    plain, mx = ob.tone_map(img, gamma=2.2)
    noisy_bmp, _ = ob.tone_map(img, gamma=2.2, seed=19, bmp_order=True)
    return plain, noisy_bmp, np.float32(mx)


# FT_OPT_MATH = glibc: the oracle calling the C runtime's expf / logf / powf.  These vectors are what glibc 2.35's FMA build
# of those functions gives — generated on a host whose libm resolves to it (every x86-64 CPU with FMA and AVX2) — and the GPU's restatement must
# reproduce them on ANY host, because it does not use the host's libm at all.
GLIBC_CASES = {"glibc_fma_c3_smooth256_48": (lambda: syn.config3()[0], 48, 48), "glibc_fma_mixed_nested_48x40": (lambda: syn.mixed_nested()[0], 48, 40)}


def render_glibc(name):
    make, W, H = GLIBC_CASES[name]
    cam = syn.default_camera().as_array()
    ob.lib.orc_set_libm(1)
    try:
        img, cnt = ob.Oracle().scene(make()).render(syn.EPSILON, syn.RAY_LENGTH, W, H, cam, nthreads=4)
        tm, mx = ob.tone_map(img, gamma=2.2)
    finally:
        ob.lib.orc_set_libm(0)
    counts = np.array([cnt[k] for k in ("rays_primary", "rays_shadow", "hits_primary", "hits_shadow", "flags")], np.int64)
    return img, counts, np.array([W, H], np.int32), tm, np.float32(mx)


if __name__ == "__main__":
    import fraytracer_amd as _ft
    if "glibc" in sys.argv[1:]:
        assert _ft.glibc_build_of_this_host() == 1, "generate the glibc fixtures on a host whose libm resolves to the FMA build"
        for name in GLIBC_CASES:
            img, counts, size, tm, mx = render_glibc(name)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img, counts=counts, size=size, tonemap=tm, max=mx)
            print(name, img.shape, counts.tolist())
        sys.exit(0)
    for name in (sys.argv[1:] or list(CASES) + list(TONEMAP)):
        if name in TONEMAP:
            plain, noisy_bmp, mx = tonemap(name)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), plain=plain, noisy_bmp=noisy_bmp, max=mx)
            print(name, plain.shape, noisy_bmp.shape, float(mx))
            continue
        img, counts, size = render(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img, counts=counts, size=size,
                            epsilon=np.float32(syn.EPSILON), length=np.float32(syn.RAY_LENGTH))
        print(name, img.shape, counts.tolist())
