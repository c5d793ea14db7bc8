#!/bin/bash
# rocprofv3 kernel trace + SQ counters for one tools/bench_scenes.py case.  Usage: bash tools/profile_scene.sh <tag> <case substring>
set -u
TAG=$1; CASE=$2
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export FT_KERNEL_ONLY=1      # bench_scenes.py: no ft_render (host output) calls, so that every ft_trace_kernel launch in the trace is one whole frame
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/tools/bench_scenes.py" "$CASE" > "$OUT/trace.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/tools/bench_scenes.py" "$CASE" > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- python3 "$REPO/tools/bench_scenes.py" "$CASE" > "$OUT/pmc_sq2.log" 2>&1
cd "$REPO"; { echo "# case: $CASE (tools/bench_scenes.py; every launch of the profiled process is this one scene at this one size)"; python3 tools/summarize_prof.py "$OUT"; } > "$OUT/summary.txt" 2>&1; cat "$OUT/summary.txt"
