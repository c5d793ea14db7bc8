#!/bin/bash
# round 4: is the carved tori kernel bound by the texture addresser at more resident waves?  Variant libraries (tools/build_variant.sh) with 6 / 7 / 8 waves
# per SIMD asked of the register allocator: kernel ms and TA_TA_BUSY of the Program.fs scene at 4000^2
REPO=$(pwd); OUT=$REPO/gpurun_out/r04ta; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export FT_KERNEL_ONLY=1
for v in rec0 tw7 tw8; do
  export FRAYTRACER_HIP_LIB=$REPO/tools/_padsweep/libft_$v.so
  python3 $REPO/tools/bench_scenes.py "Program.fs scene 4000" 2>/dev/null | grep '^{' | cut -c1-200 > $OUT/ms_$v.txt
  rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_$v -- python3 $REPO/tools/bench_scenes.py "Program.fs scene 4000" > $OUT/pmc_$v.log 2>&1
  echo "== $v"; cat $OUT/ms_$v.txt
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "carved" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print({k: "%.4g" % x for k, x in m.items()})
if m: print("TA busy per CU %.3f  VALU busy %.3f  waves/SIMD %.2f" % (m["TA_TA_BUSY_sum"] / 256 / (m["GRBM_GUI_ACTIVE"] / 8), m["SQ_ACTIVE_INST_VALU"] * 2 / (1024 * m["GRBM_GUI_ACTIVE"] / 8), m["SQ_WAVE_CYCLES"] * 4 / (1024 * m["GRBM_GUI_ACTIVE"] / 8)))
PY
done
