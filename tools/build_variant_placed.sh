#!/bin/bash
# A whole-library variant built THROUGH the layout pass (same steps as fraytracer_amd/csrc/Makefile) with extra -D flags:
#   tools/build_variant_placed.sh <name> "<-D flags>"   ->  tools/_padsweep/libft_<name>.so   (load it with FRAYTRACER_HIP_LIB)
set -e
NAME=$1; EXTRA=${2:-}
cd "$(dirname "$0")/../fraytracer_amd/csrc"
L=/opt/rocm/lib/llvm/bin
COMMON="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function"
T=$(mktemp -d)
mkdir -p ../../tools/_padsweep
/opt/rocm/bin/hipcc --offload-arch=gfx950 $COMMON $EXTRA --cuda-device-only -S kernels.hip -o $T/k.s 2>/dev/null
python3 loop_layout.py fix $T/k.s $T/k.placed.s
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/k.placed.s -o $T/k.dev.o
$L/ld.lld -shared $T/k.dev.o -o $T/k.co
$L/clang-offload-bundler --type=o --bundle-align=4096 --targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 --input=/dev/null --input=$T/k.co --output=$T/k.hipfb
/opt/rocm/bin/hipcc --offload-arch=gfx950 $COMMON $EXTRA --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/k.hipfb -c kernels.hip -o $T/k.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/_padsweep/libft_$NAME.so $T/k.o scene.o capi.o multi.o -ldl -Wl,-rpath,/opt/rocm/lib 2>/dev/null
python3 loop_layout.py check ../../tools/_padsweep/libft_$NAME.so | grep -c "fast phase"
rm -rf $T
