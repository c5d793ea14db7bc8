#!/bin/bash
# more fuzz of the final build on fresh seeds (no source change): sized to ~14 minutes
o=gpurun_out
timeout -k 10 330 python tools/fuzz_parity.py 5000000 40000 > $o/r04v_fuzz.txt 2>&1; tail -1 $o/r04v_fuzz.txt | cut -c1-400
timeout -k 10 150 python tools/fuzz_parity.py 50000 700 big > $o/r04v_fuzz_big.txt 2>&1; tail -1 $o/r04v_fuzz_big.txt | cut -c1-400
timeout -k 10 120 python tools/fuzz_cull.py 40000 1000 > $o/r04v_fuzz_cull.txt 2>&1; tail -1 $o/r04v_fuzz_cull.txt | cut -c1-400
timeout -k 10 140 python tools/fuzz_cull.py 40000 1000 nested > $o/r04v_fuzz_cull_nested.txt 2>&1; tail -1 $o/r04v_fuzz_cull_nested.txt | cut -c1-400
FT_MATH=1 timeout -k 10 80 python tools/fuzz_parity.py 5200000 10000 > $o/r04v_fuzz_glibc.txt 2>&1; tail -1 $o/r04v_fuzz_glibc.txt | cut -c1-400
FT_CARVED=0 timeout -k 10 50 python tools/fuzz_parity.py 5300000 5000 > $o/r04v_fuzz_nocarved.txt 2>&1; tail -1 $o/r04v_fuzz_nocarved.txt | cut -c1-400
