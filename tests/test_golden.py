"""Committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py from the CPU
oracle — the reference has no fixtures of its own).  CPU: the oracle still reproduces them bit for bit.
GPU: the HIP path, through the C ABI, reproduces them bit for bit."""
import os
import sys

import numpy as np
import pytest

import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
from helpers import assert_bit_equal

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402

GOLDEN = os.path.dirname(os.path.abspath(make_golden.__file__))
NAMES = sorted(make_golden.CASES)


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return z["image"], z["counts"], tuple(int(v) for v in z["size"])


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    want, counts, (W, H) = load(name)
    img, got_counts, _ = make_golden.render(name)
    assert_bit_equal(img, want, name)
    assert got_counts.tolist() == counts.tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_reproduces_golden(gpu, name):
    want, counts, (W, H) = load(name)
    scene = make_golden.CASES[name][0]()
    ext = make_golden.ext_params(name)
    img, st = gpu.scene(scene).render(syn.EPSILON, syn.RAY_LENGTH, ft.ImageSize(W, H), syn.default_camera(), **ext)
    assert_bit_equal(img, want, name)
    got = [st["rays_primary"], st["rays_shadow"], st["hits_primary"], st["hits_shadow"], st["flags"]] + ([st["rays_ext"]] if ext else [])
    assert got == counts.tolist()


@pytest.mark.parametrize("name", sorted(make_golden.TONEMAP))
def test_oracle_reproduces_golden_tone_map(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    plain, noisy_bmp, mx = make_golden.tonemap(name)
    assert np.array_equal(plain, z["plain"]) and np.array_equal(noisy_bmp, z["noisy_bmp"]) and mx == z["max"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(make_golden.TONEMAP))
def test_hip_reproduces_golden_tone_map(gpu, name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    img = np.load(os.path.join(GOLDEN, make_golden.TONEMAP[name] + ".npz"), allow_pickle=False)["image"]
    assert np.array_equal(ft.Image.toColors(2.2, None, img, gpu), z["plain"])
    assert np.array_equal(ft.Image.toColors(2.2, 19, img, gpu, bmp_order=True), z["noisy_bmp"])


def _glibc_235_fma_host():
    import platform
    return ft.glibc_build_of_this_host() == 1 and platform.libc_ver() == ("glibc", "2.35")


@pytest.mark.parametrize("name", sorted(make_golden.GLIBC_CASES))
def test_oracle_with_this_hosts_libm_reproduces_the_glibc_golden(name):
    """the fixtures are glibc 2.35's FMA build: on such a host the oracle calling the real expf / logf / powf gives them again"""
    if not _glibc_235_fma_host():
        pytest.skip("the glibc fixtures were generated with glibc 2.35's FMA build of expf / logf / powf")
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    img, counts, _, tm, mx = make_golden.render_glibc(name)
    assert_bit_equal(img, z["image"], name)
    assert counts.tolist() == z["counts"].tolist() and np.array_equal(tm, z["tonemap"]) and mx == z["max"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(make_golden.GLIBC_CASES))
def test_hip_glibc_mode_reproduces_the_glibc_golden(gpu, name):
    """FT_MATH_GLIBC_FMA on the GPU reproduces glibc 2.35's FMA-build results whatever libm the host has: the restatement runs on the device"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    W, H = (int(v) for v in z["size"])
    gpu.set_option("math", 1)
    try:
        img, st = gpu.scene(make_golden.GLIBC_CASES[name][0]()).render(syn.EPSILON, syn.RAY_LENGTH, ft.ImageSize(W, H), syn.default_camera())
        tm = ft.Image.toColors(2.2, None, z["image"], gpu)
    finally:
        gpu.set_option("math", 0)
    assert_bit_equal(img, z["image"], name)
    assert [st["rays_primary"], st["rays_shadow"], st["hits_primary"], st["hits_shadow"], st["flags"]] == z["counts"].tolist()
    assert np.array_equal(tm, z["tonemap"])
