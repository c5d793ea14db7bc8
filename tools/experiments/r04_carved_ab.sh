#!/bin/bash
# A/B of the carved-union kernels (round 4): product library with FT_CARVED on / off, then the experiment builds
# tools/_padsweep/libft_cw<waves>k<walk>.so (tools/build_variant.sh), two passes each; kernel ms per scene -> gpurun_out/$1_*.jsonl
TAG=${1:-r04a}
OUT=gpurun_out
SCENES='Program.fs console-like C2'
run() {   # name, env...
  local name=$1; shift
  env "$@" FT_KERNEL_ONLY=1 python tools/bench_scenes.py $SCENES > $OUT/${TAG}_scenes_${name}.jsonl 2>$OUT/${TAG}_scenes_${name}.err || echo "FAILED $name"
}
for pass in 1 2; do
  run product_p$pass X=1
  run general_p$pass FT_CARVED=0
  for v in cw8k0 cw8k2 cw7k0 cw7k2 cw6k0 cw6k2; do
    [ -f tools/_padsweep/libft_$v.so ] && run ${v}_p$pass FRAYTRACER_HIP_LIB=$PWD/tools/_padsweep/libft_$v.so
  done
done
python - <<'PY'
import json, glob, os, collections
tag = os.environ.get("TAG", "r04a")
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(f"gpurun_out/{tag}_scenes_*.jsonl")):
    name = os.path.basename(f)[len(tag) + 8:-6]
    for line in open(f):
        if line.startswith("{"):
            d = json.loads(line); rows[d["scene"]][name] = d["kernel_ms"]
for scene, r in rows.items():
    print(scene)
    for k in sorted(r): print(f"   {k:18s} {r[k]:8.3f} ms")
PY
