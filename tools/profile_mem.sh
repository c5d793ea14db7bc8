#!/bin/bash
# texture-addresser / L1 counters for one tools/bench_scenes.py case, few counters per pass (more than two TA or TCP
# counters in one pass exceed what the hardware collects and rocprofv3 aborts).  Usage: bash tools/profile_mem.sh <tag> <case substring>
set -u
TAG=$1; CASE=$2
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 --pmc TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_ta" -- python3 "$REPO/tools/bench_scenes.py" "$CASE" > "$OUT/pmc_ta.log" 2>&1 || echo "TA pass failed"
timeout -k 5 120 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_tcp" -- python3 "$REPO/tools/bench_scenes.py" "$CASE" > "$OUT/pmc_tcp.log" 2>&1 || echo "TCP pass failed"
timeout -k 5 120 rocprofv3 --pmc TCP_TA_TCP_STATE_READ_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_tcp2" -- python3 "$REPO/tools/bench_scenes.py" "$CASE" > "$OUT/pmc_tcp2.log" 2>&1 || echo "TCP2 pass failed"
cd "$REPO"; python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1; cat "$OUT/summary.txt"
