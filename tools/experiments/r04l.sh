#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/r04l_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04l_tests.log; tail -3 gpurun_out/r04l_tests.log
run() { local name=$1; shift; env "$@" FT_KERNEL_ONLY=1 python tools/bench_scenes.py "Program.fs" "C2 union32 4096" "C3" "mixed" "crowd" "C5" 2>/dev/null | python -c "
import sys, json
r = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('%-10s' % '$name', '  '.join('%s %.3f (%.2f)' % (d['scene'].replace(' scene', '').replace(' union32', '').replace(' smooth256','').replace('crowd of 300 combinator objects','crowd').replace(' glass','').replace(' nested',''), d['kernel_ms'], d['evals_per_ray']) for d in r), flush=True)"; }
for pass in 1 2; do run reuse=1 FT_REUSE=1; run reuse=0 FT_REUSE=0; done
python tools/bench_ext.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-80s %8.3f ms %8.1f Mrays/s' % (d['scene'][:80], d['kernel_ms'], d['Mrays/s']), flush=True)"
