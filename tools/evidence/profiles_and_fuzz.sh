# One gpurun call of the round-end evidence: rocprof summaries (lean kernel via bench.py, Program.fs scene per size) and the shorter fuzz configurations.
# Usage (GPU box, repo root): bash tools/evidence/profiles_and_fuzz.sh ; then copy gpurun_out/prof_*/summary.txt to profiles/ (README there).
set -u
mkdir -p gpurun_out/ev5
bash tools/profile.sh r03final2 > gpurun_out/ev5/profile.log 2>&1; rc=$?; tail -2 gpurun_out/ev5/profile.log; if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/profile_scene.sh r03b_pfs1000 "Program.fs scene 1000^2" > gpurun_out/ev5/pfs1000.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/profile_scene.sh r03b_pfs4000 "Program.fs scene 4000^2" > gpurun_out/ev5/pfs4000.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
echo profiles done
run() { local name=$1; shift; timeout -k 10 300 env "$@" > gpurun_out/ev5/fuzz_$name.log 2>&1; tail -1 gpurun_out/ev5/fuzz_$name.log | cut -c1-300; }
run math1 FT_MATH=1 python3 tools/fuzz_parity.py 3100000 6000
run k64 FT_TAIL_K=64 python3 tools/fuzz_parity.py 3200000 2000
run noshortcuts FT_CULL=0 FT_ESCAPE=0 FT_LAZY_UNION=0 python3 tools/fuzz_parity.py 3000000 4000
run cull python3 tools/fuzz_cull.py 70000 1500
