#!/usr/bin/env python3
"""One-off: the device math primitives against the oracle's for EVERY float32 bit pattern
(exp, log, sqrt: all 2^32 inputs each).  GPU side through ft_math_eval, CPU side through the oracle's array
entry points on a thread pool (ctypes releases the GIL)."""
import json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fraytracer_amd as ft
from oracle import binding as ob

CHUNK = 1 << 26
dev = ft.Device(0)
threads = 16
res = {}
t0 = time.perf_counter()
for name, op, cpu in (("exp", 0, ob.expf), ("log", 1, ob.logf), ("sqrt", 2, ob.sqrtf)):
    bad = 0; first = None
    for base in range(0, 1 << 32, CHUNK):
        x = np.arange(base, base + CHUNK, dtype=np.uint64).astype(np.uint32).view(np.float32)
        g = dev.math_eval(op, x)
        parts = np.array_split(x, threads)
        with ThreadPoolExecutor(threads) as ex:
            c = np.concatenate(list(ex.map(cpu, parts)))
        gb, cb = g.view(np.uint32), c.view(np.uint32)
        diff = (gb != cb) & ~(np.isnan(g) & np.isnan(c))
        n = int(diff.sum())
        if n and first is None:
            i = int(np.flatnonzero(diff)[0]); first = (hex(base + i), float(g[i]), float(c[i]))
        bad += n
    res[name] = {"inputs": 1 << 32, "mismatches": bad, "first": first}
    print(name, res[name], f"{time.perf_counter() - t0:.0f}s", flush=True)
print(json.dumps(res))
