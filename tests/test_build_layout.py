"""The build's machine-code layout pass (fraytracer_amd/csrc/loop_layout.py): every 4-children-per-trip
sphere loop of every trace kernel in the built library must sit in the fast 8-byte phase (its run of 64-bit encoded VALU
instructions starting at 4 mod 8), so the C3 frame time does not depend on a compile-time lottery.  The same pass checks the
stretches of code that run with output modifiers enabled (MODE.IEEE off, f32 denormals flushed: the NEAR sphere loop's
four-instruction root): nothing but the loop's own float forms and mode-blind instructions may stand inside."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_sphere_loop_of_the_built_library_is_in_the_fast_phase():
    lib = os.path.join(ROOT, "fraytracer_amd", "libfraytracer_hip.so")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "fraytracer_amd", "csrc", "loop_layout.py"), "check", lib], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("ft_trace_kernel") and " loop at " in l]
    assert len(lines) >= 12 and all("fast phase" in l for l in lines)
    regions = [l for l in r.stdout.splitlines() if "output-modifier region" in l]
    assert len(regions) >= 7 and all("only the sphere loop inside" in l for l in regions)
    assert {l.split(":")[0] for l in regions} >= {"ft_trace_kernel", "ft_trace_kernel_smooth_spheres", "ft_selftest_kernel"}
    kernels = {l.split(":")[0] for l in lines}
    assert {"ft_trace_kernel", "ft_trace_kernel_smooth_spheres", "ft_trace_kernel_ext", "ft_trace_kernel_smooth_spheres_ext",
            "ft_trace_kernel_calls", "ft_trace_kernel_calls_ext"} <= kernels
