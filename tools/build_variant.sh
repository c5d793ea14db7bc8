#!/bin/bash
# Experiment builds: tools/build_variant.sh <name> "<extra -D flags>" -> tools/_padsweep/libft_<name>.so (kernels.hip compiled
# directly, without the layout pass); load with FRAYTRACER_HIP_LIB.
set -e
NAME=$1; EXTRA=${2:-}
cd "$(dirname "$0")/../fraytracer_amd/csrc"
mkdir -p ../../tools/_padsweep
COMMON="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $COMMON $EXTRA -c kernels.hip -o ../../tools/_padsweep/k_$NAME.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/_padsweep/libft_$NAME.so ../../tools/_padsweep/k_$NAME.o scene.o capi.o multi.o -ldl -Wl,-rpath,/opt/rocm/lib
rm -f ../../tools/_padsweep/k_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize $EXTRA --cuda-device-only -S kernels.hip -o /tmp/k_$NAME.s 2>/dev/null
echo "$NAME: $(grep -E '^\s+\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size):|^\s+\.name:\s+ft_trace_kernel$' /tmp/k_$NAME.s | paste - - - - | sed 's/\s\+/ /g' | grep 'ft_trace_kernel ')"
