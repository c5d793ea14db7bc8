#!/bin/bash
python tools/experiments/r04_blocks_sweep.py > gpurun_out/r04c_blocks_product.jsonl 2>gpurun_out/r04c_blocks.err
FRAYTRACER_HIP_LIB=$PWD/tools/_padsweep/libft_cw7k2.so python tools/experiments/r04_blocks_sweep.py > gpurun_out/r04c_blocks_cw7k2.jsonl 2>>gpurun_out/r04c_blocks.err
for v in cw7k3 cw6k3 cw7k2; do
  for tk in 2; do
    FRAYTRACER_HIP_LIB=$PWD/tools/_padsweep/libft_$v.so FT_TAIL_K=$tk FT_KERNEL_ONLY=1 python tools/bench_scenes.py "Program.fs" 2>/dev/null | python -c "
import sys, json
r = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('%-20s' % '$v tk=$tk', '  '.join('%s %.3f' % (d['scene'].split()[-1], d['kernel_ms']) for d in r), flush=True)"
  done
done
