#!/bin/bash
# Builds one copy of the library per placement of the sphere loops (FT_LOOP_PAD = k s_nops after a 64-byte
# boundary) into tools/_padsweep/ (not tracked; travels to the GPU box).  Usage: tools/pad_sweep_build.sh "0 1 2 ... 15" [extra -D flags]
set -e
cd "$(dirname "$0")/../fraytracer_amd/csrc"
PADS=${1:-"0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15"}
EXTRA=${2:-}
COMMON="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function"
mkdir -p ../../tools/_padsweep
for k in $PADS; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 $COMMON -DFT_LOOP_PAD=$k $EXTRA -c kernels.hip -o ../../tools/_padsweep/kernels_$k.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/_padsweep/libft_pad$k.so ../../tools/_padsweep/kernels_$k.o scene.o capi.o multi.o -ldl -Wl,-rpath,/opt/rocm/lib &&
    rm -f ../../tools/_padsweep/kernels_$k.o ) &
  while [ "$(jobs -r | wc -l)" -ge 6 ]; do sleep 0.2; done
done
wait
ls ../../tools/_padsweep
