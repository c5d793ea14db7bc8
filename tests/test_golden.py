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
