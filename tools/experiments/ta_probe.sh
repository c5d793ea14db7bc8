#!/bin/bash
# Is the grid-union walk bound by the rate of its vector loads?  (1) tools/experiments/ta_rate.hip: what one CU sustains per load instruction;
# (2) texture-addresser / vector-L1 counters of the Program.fs 4000^2 frame.  Usage (GPU box): bash tools/experiments/ta_probe.sh
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/ta_probe; mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o "$OUT/ta_rate" tools/experiments/ta_rate.hip > "$OUT/hipcc.log" 2>&1 || { echo "hipcc failed"; cat "$OUT/hipcc.log"; exit 1; }
timeout -k 10 240 "$OUT/ta_rate" > "$OUT/ta_rate.json" 2> "$OUT/ta_rate.err"; rc=$?; echo "ta_rate rc=$rc"; cat "$OUT/ta_rate.err"
if [ $rc -ge 124 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters.txt" 2>&1
export FT_KERNEL_ONLY=1
CASE="Program.fs scene 4000"
run_pass() {   # tag, counters...
    local tag=$1; shift
    timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$tag" -- python3 "$REPO/tools/bench_scenes.py" "$CASE" > "$OUT/$tag.log" 2>&1
    local rc=$?; echo "$tag rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
}
run_pass ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run_pass ta2 TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
run_pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
run_pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE
cd "$REPO"
for t in ta1 ta2 tcp1 tcp2; do
    f=$(find "$OUT/$t" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 - "$f" "$t" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Kernel_Name"].startswith("ft_trace_kernel"): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
# one row per dispatch and counter (possibly per dimension: sum them per dispatch is not possible without ids; report mean x count)
for k, v in sorted(acc.items()): print(sys.argv[2], k, "mean", sum(v) / len(v), "n", len(v))
PY
done > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
