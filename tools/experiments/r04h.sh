#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/r04h_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04h_tests.log; tail -4 gpurun_out/r04h_tests.log
FT_KERNEL_ONLY=1 python tools/bench_scenes.py 2>/dev/null | tee gpurun_out/r04h_scenes.jsonl | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-45s %8.3f ms  evals/ray %.2f  culled %.3f' % (d['scene'], d['kernel_ms'], d['evals_per_ray'], d['culled_fraction']), flush=True)"
python tools/bench_ext.py 2>/dev/null | tee gpurun_out/r04h_ext.jsonl | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-90s %8.3f ms  culled %.3f site %s' % (d['scene'][:90], d['kernel_ms'], d['culled_fraction'], d['cull_site']), flush=True)"
