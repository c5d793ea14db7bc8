#!/bin/bash
# final build of round 4: GPU suite, stamped rocprof summaries (C3 fixed + glibc, Program.fs per size)
python -m pytest tests -m gpu -x -q > gpurun_out/r04z_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04z_tests.log; tail -3 gpurun_out/r04z_tests.log
bash tools/profile.sh r04z_c3 > /dev/null 2>&1; echo "== C3 fixed"; cat gpurun_out/prof_r04z_c3/summary.txt
bash tools/profile.sh r04z_c3_glibc --steps 3 --warmup 1 --no-cpu-baseline --no-side --no-reference-launch --math glibc > /dev/null 2>&1; echo "== C3 glibc"; cat gpurun_out/prof_r04z_c3_glibc/summary.txt
bash tools/profile_scene.sh r04z_pf1000 "Program.fs scene 1000" > /dev/null 2>&1; bash tools/profile_scene.sh r04z_pf4000 "Program.fs scene 4000" > /dev/null 2>&1
echo "== Program.fs"; cat gpurun_out/prof_r04z_pf1000/summary.txt gpurun_out/prof_r04z_pf4000/summary.txt
