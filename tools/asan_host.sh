#!/bin/bash
# AddressSanitizer over the HOST side of the library (builder, flattener, grid build, C ABI) — no GPU needed; GPU-side
# sanitizers are not available on the test pool.  Builds a separate library under /tmp and runs the host tests and a
# flatten-only fuzz through it.  Usage: bash tools/asan_host.sh
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO/fraytracer_amd/csrc"
FLAGS="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address -fno-omit-frame-pointer -Wno-unused-function -Wno-option-ignored"
make >/dev/null
for f in scene capi multi; do /opt/rocm/bin/hipcc $FLAGS -c $f.cpp -o /tmp/ft_${f}_asan.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=address -o /tmp/libfraytracer_hip_asan.so kernels.o /tmp/ft_scene_asan.o /tmp/ft_capi_asan.o /tmp/ft_multi_asan.o -ldl -Wl,-rpath,/opt/rocm/lib
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd "$REPO"
export ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$RT FRAYTRACER_HIP_LIB=/tmp/libfraytracer_hip_asan.so
python3 -m pytest tests/test_host_flatten.py tests/test_oracle_glass_ext.py -x -q -p no:cacheprovider
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
host = ft.Device(-1)
ok = rej = 0
for big, seeds in ((False, range(0, 400)), (True, range(0, 60))):
    for seed in seeds:
        try:
            ds = host.scene(syn.fuzz_scene(seed, big)[0]); ds.info(); ds.close(); ok += 1
        except ft.FrayTracerError:
            rej += 1
for sc in (syn.combinator_crowd()[0], syn.console_scene()[0], syn.config5()[0], syn.combinator_zoo()[0]):
    ds = host.scene(sc); ds.grid(0); ds.close()
host.close()
print("ASAN: flattened", ok, "fuzz scenes,", rej, "rejected, no findings")
PY
