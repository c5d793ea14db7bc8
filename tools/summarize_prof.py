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
# registers per kernel from the CODE OBJECT of the library that ran (tools/kernel_resources.sh): the trace's VGPR_Count column is the
# allocation in units of two registers (80 -> 40, 81 -> 88 allocated -> 44), which round 3's summaries printed as if it were the count
resources = {}
try:
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    for line in subprocess.run(["bash", os.path.join(here, "kernel_resources.sh"), _ft._lib.LIB_PATH], capture_output=True, text=True, timeout=120).stdout.splitlines():
        w = line.split()
        if len(w) >= 11 and w[1] == "vgpr":
            resources[w[0]] = f"vgpr {w[2]} sgpr {w[6]} scratch {w[8]} B spilled vgprs {w[10]}"
except Exception as e:                                             # noqa: BLE001
    print(f"# code-object resources unavailable ({e})")
print("== per-dispatch durations of ft_trace_kernel (ns) ==")
seen = set()
for f, r in rows("trace/**/*kernel_trace.csv"):
    name = r.get("Kernel_Name", "")
    if "ft_trace_kernel" in name:
        print(name, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "LDS", r.get("LDS_Block_Size"), "grid", r.get("Grid_Size_X", r.get("Grid_Size")),
              "wg", r.get("Workgroup_Size_X", r.get("Workgroup_Size")), "scratch", r.get("Scratch_Size"))
        seen.add(name.split("(")[0].strip())
for name in sorted(seen):
    print(f"code object: {name}: {resources.get(name, 'unknown')}")
for name in ("pmc_sq", "pmc_sq2", "pmc_sq3", "pmc_fetch", "pmc_write", "pmc_ta", "pmc_ta2", "pmc_tcp", "pmc_tcp2"):
    acc = defaultdict(list)
    for f, r in rows(f"{name}/**/*counter_collection.csv"):
        if "ft_trace_kernel" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if acc:
        print(f"== {name}: mean per ft_trace_kernel dispatch ==")
        for k, v in sorted(acc.items()):
            print(f"{k:28s} {sum(v) / len(v):.6g}   (n={len(v)})")
