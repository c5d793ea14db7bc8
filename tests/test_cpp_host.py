"""The compiled-language host mirror (host/cpp/FrayTracer.hpp + console.cpp, the C++ twin of
src/FrayTracer.Console/Program.fs) over the C ABI."""
import os
import subprocess

import numpy as np
import pytest

import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
from helpers import assert_bit_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "host", "cpp")


@pytest.fixture(scope="module")
def console():
    subprocess.check_call(["make", "-s", "-C", HOST])
    return os.path.join(HOST, "console")


def test_console_builds_scene_without_gpu(console):
    out = subprocess.check_output([console, "--device", "-1", "--tori", "300"], text=True)
    assert "300 tori" in out


@pytest.mark.gpu
def test_console_matches_python_host(console, gpu, tmp_path):
    raw, bmp = tmp_path / "img.f32", tmp_path / "result.bmp"
    out = subprocess.check_output([console, "--size", "64", "--tori", "120", "--raw", str(raw), "--out", str(bmp)], text=True)
    assert "Rendering..." in out and "Time =" in out
    got = np.fromfile(raw, np.float32).reshape(64, 64, 3)
    scene, _ = syn.console_scene(seed=19, n=120, size=64)            # same System.Random(19) draws in Python
    want, _ = gpu.scene(scene).render(syn.EPSILON, syn.RAY_LENGTH, ft.ImageSize(64, 64), syn.default_camera())
    assert_bit_equal(got, want, "C++ console vs Python host")
    assert bmp.read_bytes()[:2] == b"BM"
