#!/bin/bash
# fuzz of the final build (after FT_OPT_REUSE): sized to ~10 minutes
o=gpurun_out
timeout -k 10 170 python tools/fuzz_parity.py 3000000 20000 > $o/r04w_fuzz.txt 2>&1; tail -1 $o/r04w_fuzz.txt | cut -c1-400
timeout -k 10 110 python tools/fuzz_parity.py 40000 250 big > $o/r04w_fuzz_big.txt 2>&1; tail -1 $o/r04w_fuzz_big.txt | cut -c1-400
timeout -k 10 70 python tools/fuzz_cull.py 30000 500 > $o/r04w_fuzz_cull.txt 2>&1; tail -1 $o/r04w_fuzz_cull.txt | cut -c1-400
timeout -k 10 70 python tools/fuzz_cull.py 30000 500 nested > $o/r04w_fuzz_cull_nested.txt 2>&1; tail -1 $o/r04w_fuzz_cull_nested.txt | cut -c1-400
FT_TAIL_K=64 timeout -k 10 60 python tools/fuzz_parity.py 3100000 1500 > $o/r04w_fuzz_tail64.txt 2>&1; tail -1 $o/r04w_fuzz_tail64.txt | cut -c1-400
FT_MATH=1 timeout -k 10 50 python tools/fuzz_parity.py 3200000 5000 > $o/r04w_fuzz_glibc.txt 2>&1; tail -1 $o/r04w_fuzz_glibc.txt | cut -c1-400
FT_REUSE=0 timeout -k 10 50 python tools/fuzz_parity.py 3300000 5000 > $o/r04w_fuzz_noreuse.txt 2>&1; tail -1 $o/r04w_fuzz_noreuse.txt | cut -c1-400
timeout -k 10 260 python tools/fuzz_parity.py 0 70 edge > $o/r04w_fuzz_edge.txt 2>&1; tail -1 $o/r04w_fuzz_edge.txt | cut -c1-400
