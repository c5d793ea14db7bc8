#!/bin/bash
o=gpurun_out
timeout -k 10 200 python tools/fuzz_parity.py 2000000 20000 > $o/r04z_fuzz2.txt 2>&1; tail -1 $o/r04z_fuzz2.txt | cut -c1-400
FT_TAIL_K=64 timeout -k 10 100 python tools/fuzz_parity.py 1100000 2000 > $o/r04z_fuzz_tail64.txt 2>&1; tail -1 $o/r04z_fuzz_tail64.txt | cut -c1-400
FT_CARVED=0 timeout -k 10 100 python tools/fuzz_parity.py 1300000 6000 > $o/r04z_fuzz_nocarved.txt 2>&1; tail -1 $o/r04z_fuzz_nocarved.txt | cut -c1-400
timeout -k 10 330 python tools/fuzz_parity.py 500 100 edge > $o/r04z_fuzz_edge.txt 2>&1; tail -1 $o/r04z_fuzz_edge.txt | cut -c1-400
