#!/bin/bash
# rocprofv3 passes for the bench workload.  Usage (on the GPU box, from the repo root):
#   bash tools/profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{trace,pmc_sq,pmc_sq2,pmc_fetch,pmc_write}; copy the summaries to profiles/.
set -u
TAG=${1:-r01}; shift || true
ARGS=${*:---steps 3 --warmup 1 --no-cpu-baseline --no-side --no-reference-launch}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || echo "trace pass failed"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1 || echo "pmc_sq pass failed"
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq2" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_sq2.log" 2>&1 || echo "pmc_sq2 pass failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1 || echo "fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1 || echo "write pass failed"
cd "$REPO"
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
