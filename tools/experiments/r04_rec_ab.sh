#!/bin/bash
# round 4: lighter record loads in the carved walk (FT_CARVE_REC: 0 = two 16-byte loads per candidate, 1 = 16 + 12, 2 = 16 + 8 and the material read on a
# new minimum only); variant libraries tools/_padsweep/libft_<v>.so (tools/build_variant.sh: no layout pass, so rec0 is the like-for-like baseline)
TAG=${1:-r04rec}; shift
OUT=gpurun_out
run() {   # label, env...
  local name=$1; shift
  env "$@" FT_KERNEL_ONLY=1 python tools/bench_scenes.py "Program.fs" "console-like" "C2" 2>/dev/null | python -c "
import sys, json
r = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('%-12s' % '$name', '  '.join('%s %.3f' % (d['scene'][:22], d['kernel_ms']) for d in r), flush=True)"
}
for pass in 1 2 3; do
  run product X=1
  for v in "$@"; do run $v FRAYTRACER_HIP_LIB=$PWD/tools/_padsweep/libft_$v.so; done
done
