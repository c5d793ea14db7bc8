#!/bin/bash
# burst-refill threshold sweep (kernels.hip "Burst refill"): kernel ms per scene for FT_REFILL_MIN = 1 ... 64
for m in ${1:-1 16 32 40 48 56 64}; do
  echo "refillMin $m"
  FT_REFILL_MIN=$m python tools/bench_scenes.py "console-like 1000 tori 4000^2" "console-like 1000 tori 1000^2" "C2 union32 4096^2" "C2 union32 1024^2" "mixed" "crowd" "C5" "C3 smooth256 4096" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-42s %8.3f ms  %8.1f Mrays/s  lane_util %.3f' % (d['scene'], d['kernel_ms'], d['Mrays/s'], d['lane_util']))"
done
