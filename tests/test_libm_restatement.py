"""FT_OPT_MATH = glibc: the product's restatement of glibc 2.35's expf / logf / powf (fraytracer_amd/csrc/ft_libm.h — what MathF.Exp /
MathF.Log / MathF.Pow of SdfForm.fs:80,82 and FColor.fs:50-55 reach under .NET on Linux x86-64) against the C runtime of THIS machine,
over ALL 2^32 float inputs, for both of glibc's builds:

  * the FMA build (`__expf_fma` ...) is what the running libm uses on a CPU with FMA + AVX2 — every host of an MI355X;
  * the SSE2 build is reached by running the same check in a child process under GLIBC_TUNABLES=glibc.cpu.hwcaps=-FMA,-AVX2, which makes
    glibc's ifunc resolvers pick `__expf_sse2` ...

The header is compiled for the host by oracle/libm_check.cpp exactly as kernels.hip compiles it for the device (same source, -ffp-contract=off,
explicit fma() only); the device side of the proof is tests/test_gpu_parity.py::test_glibc_restatement_on_the_device (checksums over every
float against this machine's libm).  Parity with the F# program itself stays unpinned (no .NET here): this pins the one third-party
arithmetic the reference path depends on, for the platform it would run on beside this GPU."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libft_libm_check.so")
Y_GAMMA = float(np.float32(1.0) / np.float32(2.2))                   # Image.fs:38 gammaInv for Program.fs:98's 2.2f


def checker():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = C.CDLL(LIB)
    lib.chk_compare.restype = C.c_uint64
    lib.chk_compare.argtypes = [C.c_int, C.c_int, C.c_float, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(C.c_uint32)]
    lib.chk_compare_pow_pairs.restype = C.c_uint64
    lib.chk_compare_pow_pairs.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    lib.chk_eval.restype = C.c_float
    lib.chk_eval.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float]
    return lib


def running_variant():
    """which build glibc's ifunc resolvers pick here: 1 (FMA) iff the CPU has FMA and AVX2 and no tunable masks them (sysdeps/x86_64/fpu/multiarch/ifunc-fma.h)"""
    flags = open("/proc/cpuinfo").read().split("flags", 1)[1].split("\n", 1)[0].split()
    masked = "-FMA" in os.environ.get("GLIBC_TUNABLES", "") or "-AVX2" in os.environ.get("GLIBC_TUNABLES", "")
    return 1 if ("fma" in flags and "avx2" in flags and not masked) else 2


def threads():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def exhaustive(lib, variant):
    out = {}
    fb = C.c_uint32()
    for name, op, y in (("expf", 0, 0.0), ("logf", 1, 0.0), ("powf(x, 1/2.2f)", 2, Y_GAMMA)):
        bad = lib.chk_compare(op, variant, y, 0, 1 << 32, threads(), C.byref(fb))
        out[name] = (int(bad), hex(fb.value))
    return out


def test_restatement_equals_the_running_libm_on_every_float():
    lib = checker()
    v = running_variant()
    res = exhaustive(lib, v)
    print(f"glibc build in use: {'FMA' if v == 1 else 'SSE2'}; mismatches over 2^32 inputs: {res}")
    assert all(bad == 0 for bad, _ in res.values()), res


def test_the_other_glibc_build_in_a_child_process():
    """GLIBC_TUNABLES masks FMA / AVX2 for the child: its libm resolves expf / logf / powf to the SSE2 build, and the restatement's
    SSE2 variant equals it on every float; the two builds really differ (a handful of inputs), so the variants are not interchangeable"""
    if running_variant() != 1:
        pytest.skip("this CPU has no FMA / AVX2: the SSE2 build is the one the first test already proved")
    code = ("import sys; sys.path.insert(0, %r); import test_libm_restatement as t; lib = t.checker(); "
            "assert t.running_variant() == 2; r2 = t.exhaustive(lib, 2); "
            "import ctypes as C; fb = C.c_uint32(); d = lib.chk_compare(0, 1, 0.0, 0, 1 << 32, t.threads(), C.byref(fb)); "
            "print(r2, d)") % os.path.join(ROOT, "tests")
    env = dict(os.environ, GLIBC_TUNABLES="glibc.cpu.hwcaps=-FMA,-AVX2")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr
    res, differing = out.stdout.strip().rsplit("} ", 1)
    res = eval(res + "}")
    assert all(bad == 0 for bad, _ in res.values()), res
    assert 0 < int(differing) < 100            # expf: the FMA variant is NOT the SSE2 build (2 inputs at the time of writing)


def test_powf_on_random_and_special_operand_pairs():
    lib = checker()
    v = running_variant()
    rng = np.random.default_rng(5)
    n = 1 << 21
    x = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32).copy()
    y = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32).copy()
    # moderate operands (where the result is finite and non-trivial), integer exponents for negative bases, and every special value squared
    m = n // 2
    x[:m] = np.exp(rng.normal(0, 3, m)).astype(np.float32) * np.where(rng.random(m) < 0.2, -1, 1).astype(np.float32)
    y[:m] = rng.normal(0, 4, m).astype(np.float32)
    y[: m // 4] = np.rint(y[: m // 4])
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 0.5, 2.0, -2.0, 3.0, 1e-45, -1e-45, 1e-38, 3.4e38, 0.45454547, 2.2, 1e10, -1e10, 8388608.0, 16777216.0, -3.0, 0.99999994, 1.0000001], np.float32)
    gx, gy = np.meshgrid(sp, sp)
    x = np.concatenate([x, gx.ravel()]); y = np.concatenate([y, gy.ravel()])
    first = C.c_int64()
    bad = lib.chk_compare_pow_pairs(v, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.size, C.byref(first))
    assert bad == 0, (bad, x[first.value], y[first.value])


def test_known_values_of_the_restatement():
    lib = checker()
    for v in (1, 2):
        assert lib.chk_eval(0, v, 0.0, 0.0) == 1.0 and lib.chk_eval(1, v, 1.0, 0.0) == 0.0
        assert lib.chk_eval(0, v, -200.0, 0.0) == 0.0 and lib.chk_eval(0, v, 100.0, 0.0) == float("inf")
        assert lib.chk_eval(0, v, float("-inf"), 0.0) == 0.0 and np.isnan(lib.chk_eval(0, v, float("nan"), 0.0))
        assert lib.chk_eval(1, v, 0.0, 0.0) == float("-inf") and np.isnan(lib.chk_eval(1, v, -1.0, 0.0))
        assert lib.chk_eval(2, v, 4.0, 0.5) == 2.0 and lib.chk_eval(2, v, 0.0, 0.45) == 0.0 and lib.chk_eval(2, v, 1.0, 7.3) == 1.0
        assert np.isnan(lib.chk_eval(2, v, -0.5, 0.45)) and lib.chk_eval(2, v, -2.0, 3.0) == -8.0
