#!/bin/bash
# fuzz of the final build on INDEPENDENT scene streams (synthetic._fuzz_stream: seeds >= 10^7): sized to ~13 minutes
o=gpurun_out
timeout -k 10 330 python tools/fuzz_parity.py 10000000 40000 > $o/r04u_fuzz.txt 2>&1; tail -1 $o/r04u_fuzz.txt | cut -c1-400
timeout -k 10 200 python tools/fuzz_parity.py 20000000 1500 big > $o/r04u_fuzz_big.txt 2>&1; tail -1 $o/r04u_fuzz_big.txt | cut -c1-400
FT_MATH=1 timeout -k 10 60 python tools/fuzz_parity.py 30000000 7000 > $o/r04u_fuzz_glibc.txt 2>&1; tail -1 $o/r04u_fuzz_glibc.txt | cut -c1-400
FT_TAIL_K=64 timeout -k 10 70 python tools/fuzz_parity.py 40000000 1500 > $o/r04u_fuzz_tail64.txt 2>&1; tail -1 $o/r04u_fuzz_tail64.txt | cut -c1-400
timeout -k 10 120 python tools/fuzz_parity.py 50000000 30 edge > $o/r04u_fuzz_edge.txt 2>&1; tail -1 $o/r04u_fuzz_edge.txt | cut -c1-400
