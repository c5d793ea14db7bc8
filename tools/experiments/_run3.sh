set -u
mkdir -p gpurun_out/esc
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/esc/gputests.log 2>&1; rc=$?; tail -5 gpurun_out/esc/gputests.log
if [ $rc -ne 0 ]; then grep -n "^E " gpurun_out/esc/gputests.log | head -20; exit $rc; fi
for e in 1 0; do
  FT_KERNEL_ONLY=1 FT_ESCAPE=$e timeout -k 10 300 python3 tools/bench_scenes.py > gpurun_out/esc/scenes_esc$e.jsonl 2> gpurun_out/esc/scenes_esc$e.err; rc=$?
  if [ $rc -ge 124 ]; then exit $rc; fi
done
python3 - <<'PY'
import json
a={json.loads(l)['scene']:json.loads(l) for l in open('gpurun_out/esc/scenes_esc1.jsonl') if l.startswith('{')}
b={json.loads(l)['scene']:json.loads(l) for l in open('gpurun_out/esc/scenes_esc0.jsonl') if l.startswith('{')}
for k in a: print('%-45s escape on %8.3f ms (%.2f evals/ray)   off %8.3f ms (%.2f)' % (k, a[k]['kernel_ms'], a[k]['evals_per_ray'], b[k]['kernel_ms'], b[k]['evals_per_ray']))
PY
timeout -k 10 300 python3 tools/fuzz_parity.py 900000 6000 > gpurun_out/esc/fuzz_small.log 2>&1; rc=$?; tail -2 gpurun_out/esc/fuzz_small.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python3 tools/fuzz_parity.py 910000 150 big > gpurun_out/esc/fuzz_big.log 2>&1; rc=$?; tail -2 gpurun_out/esc/fuzz_big.log
