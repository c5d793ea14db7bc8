#!/usr/bin/env python3
"""Runs tools/pad_probe.py once per library in tools/_padsweep/ (built by tools/pad_sweep_build.sh), each in its own process,
twice round-robin so that slow drifts (clock, temperature) show up as differences between the two passes."""
import glob, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sorted(glob.glob(os.path.join(root, "tools", "_padsweep", "*.so")), key=lambda p: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(p))])
frames = sys.argv[1] if len(sys.argv) > 1 else "6"
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, FRAYTRACER_HIP_LIB=lib)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "pad_probe.py"), frames], env=env, capture_output=True, text=True, timeout=300)
        print(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else f"FAILED {lib}: {r.stderr[-500:]}", flush=True)
