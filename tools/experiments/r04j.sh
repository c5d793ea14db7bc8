#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/r04j_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04j_tests.log; tail -3 gpurun_out/r04j_tests.log
run() { local name=$1; shift; env "$@" FT_KERNEL_ONLY=1 python tools/bench_scenes.py "Program.fs" "C2 union32 4096" 2>/dev/null | python -c "
import sys, json
r = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('%-26s' % '$name', '  '.join('%s %.3f' % (d['scene'].replace(' scene', '').replace(' union32', ''), d['kernel_ms']) for d in r), flush=True)"; }
for pass in 1 2; do run default X=1; run refill_min=56 FT_REFILL_MIN=56; run refill_min=48 FT_REFILL_MIN=48; run tail_k=4 FT_TAIL_K=4; run tail_k=1 FT_TAIL_K=1; run tail_k=0 FT_TAIL_K=0; done
python tools/tail_probe.py 4096 > gpurun_out/r04j_tail_probe_4096.json 2>/dev/null; cat gpurun_out/r04j_tail_probe_4096.json
python tools/tail_probe.py 8192 > gpurun_out/r04j_tail_probe_8192.json 2>/dev/null; cat gpurun_out/r04j_tail_probe_8192.json
