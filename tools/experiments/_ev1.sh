set -u
mkdir -p gpurun_out/ev
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/ev/gputests.log 2>&1; rc=$?; tail -2 gpurun_out/ev/gputests.log
if [ $rc -ne 0 ]; then grep -n "^E " gpurun_out/ev/gputests.log | head; exit $rc; fi
bash tools/profile.sh r03final > gpurun_out/ev/profile.log 2>&1; rc=$?; tail -3 gpurun_out/ev/profile.log; if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/profile_scene.sh r03_pfs1000 "Program.fs scene 1000" > gpurun_out/ev/pfs1000.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/profile_scene.sh r03_pfs4000 "Program.fs scene 4000" > gpurun_out/ev/pfs4000.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
echo profiles done
