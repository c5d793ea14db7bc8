#!/bin/bash
# VGPRs / SGPRs / LDS / scratch (spills) of every kernel in the built library (from the code object's metadata)
LIB=${1:-fraytracer_amd/libfraytracer_hip.so}
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=<(objcopy -O binary --only-section=.hip_fatbin $LIB /dev/stdout) --output=$T/co --unbundle 2>/dev/null || {
  objcopy -O binary --only-section=.hip_fatbin $LIB $T/fatbin; $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fatbin --output=$T/co --unbundle; }
$LLVM/llvm-readelf --notes $T/co | awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.sgpr_count:/{s=$2} /\.private_segment_fixed_size:/{p=$2} /\.vgpr_spill_count:/{sp=$2} /\.agpr_count:/{a=$2} /\.wavefront_size:/{printf "%-48s vgpr %3s agpr %3s sgpr %3s scratch %5s spill %3s\n", n, v, a, s, p, sp}'
rm -rf $T
