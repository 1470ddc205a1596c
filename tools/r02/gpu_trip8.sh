#!/bin/bash
# sell_window v2 sweep, then the whole GPU test tier, then the default bench line
set -o pipefail
mkdir -p gpurun_out
run() { echo "$*: $(timeout -k 10 300 python tools/run_one.py "$@" --iters 200 2>&1 | tail -1)"; }
for cfg in "4 2" "4 4" "2 4" "2 8" "1 8" "1 16"; do set -- $cfg
  run --workload cant --format sell_c_sigma --opt sell_window=1 --opt sell_split=$1 --opt sell_group=$2
done
run --workload cant --format sell_c_sigma
run --workload cant --dtype f32 --format sell_c_sigma
run --workload cant --dtype f32 --format csr_stream
for cfg in "1 8" "1 16" "2 8" "4 4"; do set -- $cfg
  run --workload pwtk --dtype f32 --format sell_c_sigma --opt sell_window=1 --opt sell_split=$1 --opt sell_group=$2
done
run --workload pwtk --dtype f32 --format sell_c_sigma
for cfg in "1 8" "1 16" "2 8"; do set -- $cfg
  run --workload pwtk --format sell_c_sigma --opt sell_window=1 --opt sell_split=$1 --opt sell_group=$2
done
run --workload pwtk --format sell_c_sigma
run --workload scircuit --format sell_c_sigma
run --workload scircuit --format csr_vector
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_tests_trip8.log 2>&1; rc=$?
tail -4 gpurun_out/r02_gpu_tests_trip8.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_trip8.json 2> gpurun_out/r02_bench_trip8.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r02_bench_trip8.json; tail -5 gpurun_out/r02_bench_trip8.err
