set -u
mkdir -p gpurun_out/ev8
timeout -k 10 900 python3 tools/fuzz_parity.py 5000000 110000 > gpurun_out/ev8/fuzz_default.log 2>&1; tail -1 gpurun_out/ev8/fuzz_default.log | cut -c1-330
timeout -k 10 230 python3 tools/fuzz_parity.py 21000 500 big > gpurun_out/ev8/fuzz_big.log 2>&1; tail -1 gpurun_out/ev8/fuzz_big.log | cut -c1-330
