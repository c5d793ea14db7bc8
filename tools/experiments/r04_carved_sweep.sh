#!/bin/bash
# round 4: latency mode / resident workgroups / chunk size for the carved-union kernels on the Program.fs scene; kernel ms -> gpurun_out/$1_sweep.txt
TAG=${1:-r04b}
OUT=gpurun_out
run() {   # label, env...
  local name=$1; shift
  env "$@" FT_KERNEL_ONLY=1 python tools/bench_scenes.py "Program.fs" 2>/dev/null | python -c "
import sys, json
r = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('%-34s' % '$name', '  '.join('%s %.3f' % (d['scene'].split()[-1], d['kernel_ms']) for d in r), flush=True)"
}
for pass in 1 2; do
  for v in product cw8k0 cw7k0 cw7k2 cw6k2; do
    L=""; [ $v != product ] && L="FRAYTRACER_HIP_LIB=$PWD/tools/_padsweep/libft_$v.so"
    for tk in 0 2; do run "$v tail_k=$tk" $L FT_TAIL_K=$tk; done
  done
  for mb in 4 5 6 7; do run "product tail_k=2 max_blocks=$mb" FT_TAIL_K=2 FT_MAX_BLOCKS_PER_CU=$mb; done
  for mb in 4 5 6; do run "cw7k2 tail_k=2 max_blocks=$mb" FRAYTRACER_HIP_LIB=$PWD/tools/_padsweep/libft_cw7k2.so FT_TAIL_K=2 FT_MAX_BLOCKS_PER_CU=$mb; done
  run "product tail_k=2 chunk=32" FT_TAIL_K=2 FT_CHUNK=32
  run "product tail_k=4" FT_TAIL_K=4
  run "product tail_k=8" FT_TAIL_K=8
done
