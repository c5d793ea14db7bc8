#!/bin/bash
# round 4: micro-variants of the carved walk (branch-free NaN minimum, byte offset from the record, both tests computed eagerly); product = none of them
run() { local name=$1; shift; env "$@" FT_KERNEL_ONLY=1 python tools/bench_scenes.py "Program.fs" "console-like" "C2 union32 4096" 2>/dev/null | python -c "
import sys, json
r = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('%-10s' % '$name', '  '.join('%s %.3f' % (d['scene'].replace(' scene', '').replace(' union32', '').replace(' 1000 tori',''), d['kernel_ms']) for d in r), flush=True)"; }
for pass in 1 2 3; do
  run product X=1
  for v in minsel byteoff eager all3 ms_bo; do run $v FRAYTRACER_HIP_LIB=$PWD/tools/_padsweep/libft_cv_$v.so; done
done
