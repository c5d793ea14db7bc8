"""`Device.default()` — the context behind every call that names no device — must be constructible (it used to take a
non-re-entrant lock twice and hang) and be one object per index.  Host-only context here; the GPU twin is in test_gpu_parity.py."""
import threading

import fraytracer_amd as ft


def test_default_device_is_constructed_once_without_deadlock():
    got = []
    t = threading.Thread(target=lambda: got.append(ft.Device.default(-1)))
    t.start(); t.join(20)
    assert not t.is_alive(), "Device.default() hangs"
    assert got and got[0] is ft.Device.default(-1) and got[0].index == -1
