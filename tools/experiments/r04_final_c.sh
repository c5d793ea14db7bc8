#!/bin/bash
# final build of round 4: differential fuzz (HIP path against the CPU oracle)
o=gpurun_out
timeout -k 10 420 python tools/fuzz_parity.py 1000000 60000 > $o/r04z_fuzz.txt 2>&1; tail -1 $o/r04z_fuzz.txt
timeout -k 10 240 python tools/fuzz_parity.py 30000 600 big > $o/r04z_fuzz_big.txt 2>&1; tail -1 $o/r04z_fuzz_big.txt
timeout -k 10 200 python tools/fuzz_cull.py 20000 1500 > $o/r04z_fuzz_cull.txt 2>&1; tail -1 $o/r04z_fuzz_cull.txt
timeout -k 10 200 python tools/fuzz_cull.py 20000 1500 nested > $o/r04z_fuzz_cull_nested.txt 2>&1; tail -1 $o/r04z_fuzz_cull_nested.txt
timeout -k 10 120 python tools/fuzz_cull.py 20000 60 wide > $o/r04z_fuzz_cull_wide.txt 2>&1; tail -1 $o/r04z_fuzz_cull_wide.txt
FT_TAIL_K=64 timeout -k 10 120 python tools/fuzz_parity.py 1100000 8000 > $o/r04z_fuzz_tail64.txt 2>&1; tail -1 $o/r04z_fuzz_tail64.txt
FT_MATH=1 timeout -k 10 120 python tools/fuzz_parity.py 1200000 8000 > $o/r04z_fuzz_glibc.txt 2>&1; tail -1 $o/r04z_fuzz_glibc.txt
FT_CARVED=0 timeout -k 10 120 python tools/fuzz_parity.py 1300000 8000 > $o/r04z_fuzz_nocarved.txt 2>&1; tail -1 $o/r04z_fuzz_nocarved.txt
timeout -k 10 420 python tools/fuzz_parity.py 500 160 edge > $o/r04z_fuzz_edge.txt 2>&1; tail -1 $o/r04z_fuzz_edge.txt
