#!/bin/bash
# rocprofv3 vector-memory-pipe counters (TA / TCP / SQ levels) for one tools/bench_scenes.py case.  Usage: bash tools/profile_scene_mem.sh <tag> <case substring>
set -u
TAG=$1; CASE=$2
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export FT_KERNEL_ONLY=1
pass() { local name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/tools/bench_scenes.py" "$CASE" > "$OUT/$name.log" 2>&1; }
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/tools/bench_scenes.py" "$CASE" > "$OUT/trace.log" 2>&1
pass pmc_ta TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass pmc_ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pass pmc_tcp TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
pass pmc_tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE
pass pmc_sq3 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
cd "$REPO"; { echo "# case: $CASE (tools/bench_scenes.py; every launch of the profiled process is this one scene at this one size)"; python3 tools/summarize_prof.py "$OUT"; } > "$OUT/summary.txt" 2>&1; cat "$OUT/summary.txt"
for n in pmc_ta pmc_ta2 pmc_tcp pmc_tcp2 pmc_sq3; do tail -3 "$OUT/$n.log" | cut -c1-300; done
