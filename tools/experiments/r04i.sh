#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/r04i_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04i_tests.log; tail -4 gpurun_out/r04i_tests.log
timeout -k 10 300 python tools/fuzz_cull.py 0 150 nested > gpurun_out/r04i_fuzz_cull_nested.txt 2>&1; tail -1 gpurun_out/r04i_fuzz_cull_nested.txt
timeout -k 10 200 python tools/fuzz_cull.py 5000 100 > gpurun_out/r04i_fuzz_cull.txt 2>&1; tail -1 gpurun_out/r04i_fuzz_cull.txt
timeout -k 10 200 python tools/fuzz_parity.py 910000 3000 > gpurun_out/r04i_fuzz.txt 2>&1; tail -1 gpurun_out/r04i_fuzz.txt
timeout -k 10 200 python tools/fuzz_parity.py 20000 60 big > gpurun_out/r04i_fuzz_big.txt 2>&1; tail -1 gpurun_out/r04i_fuzz_big.txt
