#!/bin/bash
# SELL with the x window in LDS: parity, then (waves per slice, slices per group) on the cant and pwtk twins; final column-blocked kernel
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sell or coo or merge or blocked" > gpurun_out/r02_parity_sellw.log 2>&1; rc=$?
tail -3 gpurun_out/r02_parity_sellw.log
[ $rc -ne 0 ] && exit $rc
run() { echo "$*: $(timeout -k 10 300 python tools/run_one.py "$@" --iters 200 2>&1 | tail -1)"; }
for cfg in "4 1" "4 2" "4 4" "2 2" "2 4" "2 8" "1 4" "1 8" "1 16"; do set -- $cfg
  run --workload cant --format sell_c_sigma --opt sell_window=1 --opt sell_split=$1 --opt sell_group=$2
done
run --workload cant --format sell_c_sigma
run --workload cant --format csr_stream
for cfg in "1 4" "1 8" "1 16" "2 4" "2 8" "4 4"; do set -- $cfg
  run --workload pwtk --dtype f32 --format sell_c_sigma --opt sell_window=1 --opt sell_split=$1 --opt sell_group=$2
done
run --workload pwtk --dtype f32 --format sell_c_sigma
run --workload pwtk --dtype f32 --format csr_stream
run --workload pwtk --format sell_c_sigma
run --workload pwtk --format csr_stream
echo "soc-LJ f64: $(timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format coo --opt col_blocks=-1 --iters 30 2>&1 | tail -1)"
echo "soc-LJ f32: $(timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format coo --dtype f32 --opt col_blocks=-1 --iters 30 2>&1 | tail -1)"
echo "soc-LJ merge: $(timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format csr_merge --iters 30 2>&1 | tail -1)"
