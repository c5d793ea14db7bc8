#!/bin/bash
# One gpurun call that produces a build's whole evidence set (when GPU minutes are short): GPU suite, short fuzz, rocprof summaries (installed into
# profiles/ ON THE BOX so that the bench line of the same call can quote them), bench lines, scene benches, tail probe, RCCL rehearsal, whole-frame checks.
# Usage (GPU box, repo root): bash tools/evidence/all_in_one.sh ; then copy gpurun_out/ev_all/* into profiles/ under the round's names.
set -u
O=gpurun_out/ev_all; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; rc=$?; tail -2 $O/gputests.log
if [ $rc -ne 0 ]; then grep -n "^E " $O/gputests.log | head; exit $rc; fi
FT_KERNEL_ONLY=1 timeout -k 10 120 python3 tools/bench_scenes.py "C3 smooth256" 2>/dev/null | cut -c1-120
timeout -k 10 300 python3 tools/fuzz_cull.py 80000 1500 > $O/fuzz_cull.log 2>&1; rc=$?; tail -1 $O/fuzz_cull.log | cut -c1-400; if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 200 python3 tools/fuzz_parity.py 6000000 10000 > $O/fuzz_default.log 2>&1; rc=$?; tail -1 $O/fuzz_default.log | cut -c1-330; if [ $rc -ge 124 ]; then exit $rc; fi
grep -q '"mismatching_scenes": 0' $O/fuzz_cull.log && grep -q '"mismatching_scenes": 0' $O/fuzz_default.log || { echo "FUZZ MISMATCH"; exit 1; }
bash tools/profile.sh evall > $O/profile.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/profile_scene.sh evall_pfs1000 "Program.fs scene 1000^2" > $O/pfs1000.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/profile_scene.sh evall_pfs4000 "Program.fs scene 4000^2" > $O/pfs4000.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
cp gpurun_out/prof_evall/summary.txt $O/final_summary.txt; cp gpurun_out/prof_evall_pfs1000/summary.txt $O/program_fs_1000_summary.txt; cp gpurun_out/prof_evall_pfs4000/summary.txt $O/program_fs_4000_summary.txt
cp $O/final_summary.txt profiles/r03_final_summary.txt; cp $O/program_fs_1000_summary.txt profiles/r03_program_fs_1000_summary.txt; cp $O/program_fs_4000_summary.txt profiles/r03_program_fs_4000_summary.txt
echo profiles done
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo bench rc=$?
timeout -k 10 300 python3 tools/bench_scenes.py > $O/bench_scenes.jsonl 2> $O/bench_scenes.err; echo scenes rc=$?
timeout -k 10 300 python3 tools/bench_ext.py > $O/bench_ext.jsonl 2> $O/bench_ext.err; echo ext rc=$?
timeout -k 10 120 python3 tools/tail_probe.py > $O/tail_probe.json 2>/dev/null; cat $O/tail_probe.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29546 timeout -k 10 300 python bench.py --force-dist --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_force_dist.json 2> $O/bench_force_dist.err; echo dist rc=$?
timeout -k 10 600 python tools/full_frame_check.py > $O/full_frame.jsonl 2> $O/full_frame.err; echo full rc=$?
python3 -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; c=d['config']; p=c['program_fs_scene']
print('value',d['value'],'ms',d['ms_per_step'],'frac',r['frac'],'frac_executed',r['frac_executed'],'valu_busy',r['valu_busy_pmc'],'traffic',r['traffic'],'mhz',c['shader_mhz'],'delta',c.get('max_abs_delta_vs_oracle'))
print('program.fs',p['kernel_ms'],p['roofline']['frac_reference_work'],p['roofline']['frac_executed'],p['roofline']['valu_busy_pmc'],p['max_abs_delta_vs_oracle'])
"
