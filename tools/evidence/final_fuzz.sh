set -u
O=gpurun_out/ev10; mkdir -p $O
timeout -k 10 420 python3 tools/fuzz_parity.py 7000000 50000 > $O/fuzz_default.log 2>&1; tail -1 $O/fuzz_default.log | cut -c1-330
timeout -k 10 300 python3 tools/fuzz_cull.py 100000 2500 > $O/fuzz_cull.log 2>&1; tail -1 $O/fuzz_cull.log | cut -c1-400
