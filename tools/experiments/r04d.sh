#!/bin/bash
run() {   # label, env...
  local name=$1; shift
  env "$@" FT_TAIL_K=2 FT_KERNEL_ONLY=1 python tools/bench_scenes.py "Program.fs" "C2 union32 4096" "C2 union32 1024" 2>/dev/null | python -c "
import sys, json
r = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('%-22s' % '$name', '  '.join('%s %.3f' % (d['scene'].replace(' scene', '').replace(' union32', ''), d['kernel_ms']) for d in r), flush=True)"
}
for pass in 1 2; do
for v in cw7k2 cw6k2 cw5k2 cw5k3 cw4k3; do
  for mb in 0 4 3; do run "$v mb=$mb" FRAYTRACER_HIP_LIB=$PWD/tools/_padsweep/libft_$v.so FT_MAX_BLOCKS_PER_CU=$mb; done
done
done
