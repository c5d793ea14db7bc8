"""Condense rocprofv3 CSV output of tools/profile.sh into a short text summary (for profiles/)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]

# stamp: which sources the profiled library was built from (bench.py quotes PMC figures only from a summary whose stamp equals the
# hash of the library it runs on; fraytracer_amd/csrc/source_hash.py --rev <commit> maps a hash to a commit)
try:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import fraytracer_amd as _ft
    _info = _ft.build_info()
    print(f"# build src={_info['src']} kind={_info['kind']}")
except Exception as e:                                             # noqa: BLE001 - the summary is still useful without the stamp
    print(f"# build unknown ({e})")


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield f, r


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f, r in rows("trace/**/*kernel_stats.csv"):
    print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
print("== per-dispatch durations of ft_trace_kernel (ns) ==")
for f, r in rows("trace/**/*kernel_trace.csv"):
    if "ft_trace_kernel" in r.get("Kernel_Name", ""):
        print(r.get("Kernel_Name"), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "VGPR", r.get("VGPR_Count"), "SGPR", r.get("SGPR_Count"),
              "LDS", r.get("LDS_Block_Size"), "grid", r.get("Grid_Size"), "wg", r.get("Workgroup_Size"))
for name in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_ta", "pmc_tcp", "pmc_tcp2"):
    acc = defaultdict(list)
    for f, r in rows(f"{name}/**/*counter_collection.csv"):
        if "ft_trace_kernel" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if acc:
        print(f"== {name}: mean per ft_trace_kernel dispatch ==")
        for k, v in sorted(acc.items()):
            print(f"{k:28s} {sum(v) / len(v):.6g}   (n={len(v)})")
