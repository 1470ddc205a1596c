#!/bin/bash
# round 2: branch-free / triple-buffered column-blocked kernel: parity, then barrier cadence, block size and the cost of the LDS atomics
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "coo or merge or blocked" > gpurun_out/r02_parity_coo2.log 2>&1; rc=$?
tail -3 gpurun_out/r02_parity_coo2.log
[ $rc -ne 0 ] && exit $rc
for sync in 1 2 4 0; do
  for cb in -1 74; do
    echo "SYNC=$sync cb=$cb: $(SPMV_MI355X_COOB_SYNC=$sync timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format coo --opt col_blocks=$cb --iters 30 2>&1 | tail -1)"
  done
done
echo "DEBUG=1 (no atomics, wrong sums) cb=74: $(SPMV_MI355X_COOB_DEBUG=1 timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format coo --opt col_blocks=74 --iters 30 2>&1 | tail -1)"
echo "f32 cb=-1: $(timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format coo --dtype f32 --opt col_blocks=-1 --iters 30 2>&1 | tail -1)"
