#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sell" > gpurun_out/r02_parity_sellw4.log 2>&1; rc=$?
tail -3 gpurun_out/r02_parity_sellw4.log
[ $rc -ne 0 ] && exit $rc
run() { echo "$*: $(timeout -k 10 300 python tools/run_one.py "$@" --iters 200 2>&1 | tail -1)"; }
run --workload cant --format sell_c_sigma
run --workload cant --format sell_c_sigma --opt sell_window=1 --opt sell_split=2 --opt sell_group=8
run --workload cant --format sell_c_sigma --opt sell_window=1 --opt sell_split=4 --opt sell_group=2
run --workload cant --format csr_stream
run --workload cant --dtype f32 --format sell_c_sigma
run --workload pwtk --dtype f32 --format sell_c_sigma
run --workload pwtk --dtype f32 --format sell_c_sigma --opt sell_window=1 --opt sell_split=1 --opt sell_group=16
run --workload pwtk --dtype f32 --format csr_stream
run --workload pwtk --format sell_c_sigma
run --workload pwtk --format csr_stream
