#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/r04g_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04g_tests.log; tail -4 gpurun_out/r04g_tests.log
FT_KERNEL_ONLY=1 python tools/bench_scenes.py "Program.fs" "console-like" "C2 union32 4096" "mixed" "crowd" "C3" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-45s %8.3f ms  evals/ray %.2f' % (d['scene'], d['kernel_ms'], d['evals_per_ray']), flush=True)"
for mb in 3 4 5; do FT_MAX_BLOCKS_PER_CU=$mb FT_KERNEL_ONLY=1 python tools/bench_scenes.py "C3" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('C3 max_blocks=$mb %8.3f ms' % d['kernel_ms'], flush=True)"; done
timeout -k 10 400 python tools/fuzz_parity.py 0 150 edge > gpurun_out/r04g_fuzz_edge.txt 2>&1; tail -2 gpurun_out/r04g_fuzz_edge.txt
timeout -k 10 300 python tools/fuzz_parity.py 900000 400 > gpurun_out/r04g_fuzz.txt 2>&1; tail -1 gpurun_out/r04g_fuzz.txt
