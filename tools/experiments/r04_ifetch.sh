#!/bin/bash
# round 4: instruction-fetch and issue-cycle counters of the carved tori kernel (Program.fs scene 4000^2) and, for comparison, of the lean kernel (C3 4096^2)
REPO=$(pwd); OUT=$REPO/gpurun_out/r04if; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export FT_KERNEL_ONLY=1
for c in "Program.fs scene 4000" "C3 smooth256"; do
  t=$(echo "$c" | cut -c1-2)
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $OUT/a_$t -- python3 $REPO/tools/bench_scenes.py "$c" > $OUT/a_$t.log 2>&1
  rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/b_$t -- python3 $REPO/tools/bench_scenes.py "$c" > $OUT/b_$t.log 2>&1
  rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_BUSY_CYCLES SQC_TC_STALL GRBM_GUI_ACTIVE --output-format csv -d $OUT/c_$t -- python3 $REPO/tools/bench_scenes.py "$c" > $OUT/c_$t.log 2>&1
  echo "== $c"
  python3 - <<PY
import csv, glob, collections
for p in "abc":
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s_$t/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            if "ft_trace_kernel" in r["Kernel_Name"]: acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k in sorted(acc): print("%-40s %-30s %.5g (n=%d)" % (k[0], k[1], sum(acc[k]) / len(acc[k]), len(acc[k])))
PY
  tail -2 $OUT/a_$t.log | cut -c1-200
done
