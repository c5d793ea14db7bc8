#!/bin/bash
# final build of round 4: bench lines (default, glibc, 1-rank RCCL rehearsal incl. the C4 block), every scene, extensions
python bench.py --steps 10 --warmup 3 > gpurun_out/r04z_bench.json 2>gpurun_out/r04z_bench.err
python bench.py --steps 5 --warmup 2 --math glibc --no-side > gpurun_out/r04z_bench_glibc.json 2>>gpurun_out/r04z_bench.err
python bench.py --force-dist --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r04z_bench_forcedist.json 2>>gpurun_out/r04z_bench.err
FT_KERNEL_ONLY=1 python tools/bench_scenes.py > gpurun_out/r04z_scenes.jsonl 2>/dev/null
python tools/bench_ext.py > gpurun_out/r04z_ext.jsonl 2>/dev/null
python tools/experiments/r04_blocks_sweep.py > gpurun_out/r04z_blocks.jsonl 2>/dev/null
python - <<PY
import json
for f in ("r04z_bench", "r04z_bench_glibc", "r04z_bench_forcedist"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], d["config"].get("arithmetic"), "frac", r["frac"], "ref-work", r["frac_reference_work"], "busy", r["valu_busy_pmc"], "traffic", r["traffic"])
        for k in ("program_fs_scene", "glibc_math_mode", "c4_8192"):
            if k in d["config"]: print("   ", k, json.dumps(d["config"][k])[:600])
        print("   delta", d["config"].get("max_abs_delta_vs_oracle"), "cpu", d.get("cpu_baseline", {}).get("value"))
    except Exception as e: print(f, "ERR", e)
for f in ("r04z_scenes", "r04z_ext"):
    for l in open(f"gpurun_out/{f}.jsonl"):
        if l.startswith("{"):
            d = json.loads(l); print("%-90s %8.3f ms %8.1f Mrays/s culled %.3f" % (d["scene"][:90], d["kernel_ms"], d["Mrays/s"], d.get("culled_fraction", 0)))
PY
