# One gpurun call of the round-end evidence: the bench line, every scene and extension, the tail probe, the RCCL rehearsal and the whole-frame checks.
# Run after the profile summaries of the same build are in profiles/ (bench.py quotes PMC figures only from a summary with the running build's stamp).
set -u
mkdir -p gpurun_out/ev6
timeout -k 10 400 python bench.py > gpurun_out/ev6/bench.json 2> gpurun_out/ev6/bench.err; echo bench rc=$?
timeout -k 10 300 python3 tools/bench_scenes.py > gpurun_out/ev6/bench_scenes.jsonl 2> gpurun_out/ev6/bench_scenes.err; echo scenes rc=$?
timeout -k 10 300 python3 tools/bench_ext.py > gpurun_out/ev6/bench_ext.jsonl 2> gpurun_out/ev6/bench_ext.err; echo ext rc=$?
timeout -k 10 120 python3 tools/tail_probe.py > gpurun_out/ev6/tail_probe.json 2>/dev/null; cat gpurun_out/ev6/tail_probe.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29545 timeout -k 10 300 python bench.py --force-dist --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ev6/bench_force_dist.json 2> gpurun_out/ev6/bench_force_dist.err; echo dist rc=$?
timeout -k 10 600 python tools/full_frame_check.py > gpurun_out/ev6/full_frame.jsonl 2> gpurun_out/ev6/full_frame.err; echo full rc=$?
python3 -c "
import json
d=json.loads(open('gpurun_out/ev6/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; c=d['config']
print('value',d['value'],'ms',d['ms_per_step'],'frac',r['frac'],'frac_executed',r['frac_executed'],'valu_busy',r['valu_busy_pmc'],'traffic',r['traffic'],'mhz',c['shader_mhz'],'delta',c.get('max_abs_delta_vs_oracle'))
p=c['program_fs_scene']; print('program.fs',p['kernel_ms'],p['kernel_ms_with_every_ray_marched_to_its_end'],p['roofline']['frac_reference_work'],p['roofline']['frac_executed'],p['roofline']['valu_busy_pmc'],p['max_abs_delta_vs_oracle'])
print('glibc',c['glibc_math_mode']['value'],c['glibc_math_mode']['max_abs_delta_vs_oracle_with_libm'])
"
