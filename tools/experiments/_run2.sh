set -u
mkdir -p gpurun_out/cull
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/cull/gputests.log 2>&1; rc=$?; tail -5 gpurun_out/cull/gputests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for c in 1 0; do
  FT_KERNEL_ONLY=1 FT_CULL=$c timeout -k 10 200 python3 tools/bench_scenes.py "C3" > gpurun_out/cull/scenes_cull$c.jsonl 2> gpurun_out/cull/scenes_cull$c.err; rc=$?
  if [ $rc -ge 124 ]; then exit $rc; fi
  cut -c1-400 gpurun_out/cull/scenes_cull$c.jsonl
done
timeout -k 10 300 python bench.py > gpurun_out/cull/bench.json 2> gpurun_out/cull/bench.err; echo bench rc=$?
python3 -c "
import json
d=json.loads(open('gpurun_out/cull/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['shader_mhz'], d['config'].get('max_abs_delta_vs_oracle'), d['config']['glibc_math_mode']['value'], d['config']['north_star_target']['value'])
"
