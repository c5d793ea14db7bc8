#!/usr/bin/env python3
"""How long does page-locking a 201 MB host buffer take (hipHostRegister / hipHostUnregister), for touched and for untouched
(freshly mapped) memory, and how fast are device-to-host copies into pageable / page-locked memory?"""
import ctypes as C, time, mmap
import numpy as np
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]; hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
n = 4096 * 4096 * 12
d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), n) == 0
hip.hipDeviceSynchronize()
def t(f):
    t0 = time.perf_counter(); r = f(); return (time.perf_counter() - t0) * 1e3, r
for kind in ("touched", "cold", "cold", "touched"):
    a = np.empty(n, np.uint8)
    if kind == "touched": a[:] = 1
    p = a.ctypes.data
    tr, rc = t(lambda: hip.hipHostRegister(p, n, 0))
    tc, _ = t(lambda: hip.hipMemcpy(p, d, n, 2))
    tc2, _ = t(lambda: hip.hipMemcpy(p, d, n, 2))
    tu, rc2 = t(lambda: hip.hipHostUnregister(p))
    print(f"{kind:8s}: register {tr:7.2f} ms (rc {rc}), D2H into page-locked {tc:6.2f} ms then {tc2:6.2f} ms ({n / tc2 / 1e6:.1f} GB/s), unregister {tu:6.2f} ms (rc {rc2})")
    del a
for kind in ("touched", "cold"):
    a = np.empty(n, np.uint8)
    if kind == "touched": a[:] = 1
    tc, _ = t(lambda: hip.hipMemcpy(a.ctypes.data, d, n, 2))
    tc2, _ = t(lambda: hip.hipMemcpy(a.ctypes.data, d, n, 2))
    print(f"{kind:8s}: D2H into pageable {tc:6.2f} ms, again {tc2:6.2f} ms ({n / tc2 / 1e6:.1f} GB/s)")
    del a
