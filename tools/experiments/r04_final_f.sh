#!/bin/bash
# edge and big fuzz of the final build on independent scene streams: sized to ~11 minutes
o=gpurun_out
timeout -k 10 420 python tools/fuzz_parity.py 50000000 100 edge > $o/r04t_fuzz_edge.txt 2>&1; tail -1 $o/r04t_fuzz_edge.txt | cut -c1-400
timeout -k 10 200 python tools/fuzz_parity.py 60000000 1500 big > $o/r04t_fuzz_big.txt 2>&1; tail -1 $o/r04t_fuzz_big.txt | cut -c1-400
FT_ESCAPE=0 timeout -k 10 50 python tools/fuzz_parity.py 70000000 5000 > $o/r04t_fuzz_noescape.txt 2>&1; tail -1 $o/r04t_fuzz_noescape.txt | cut -c1-400
