# One gpurun call: the long differential-fuzz configurations (default, big scenes, glibc arithmetic, latency mode forced, shortcuts off).
set -u
mkdir -p gpurun_out/ev
run() { # name, env..., args
  local name=$1; shift
  timeout -k 10 420 env "$@" > gpurun_out/ev/fuzz_$name.log 2>&1; local rc=$?; tail -1 gpurun_out/ev/fuzz_$name.log | cut -c1-330; if [ $rc -ge 124 ]; then echo "$name timed out"; fi
}
run default python3 tools/fuzz_parity.py 2000000 40000
run big python3 tools/fuzz_parity.py 30000 600 big
run math1 FT_MATH=1 python3 tools/fuzz_parity.py 2100000 10000
run k64 FT_TAIL_K=64 python3 tools/fuzz_parity.py 2200000 3000
run noshortcuts FT_CULL=0 FT_ESCAPE=0 FT_LAZY_UNION=0 python3 tools/fuzz_parity.py 2000000 5000
