#!/bin/bash
# gpurun -- tools/gpu_coob.sh : parity of the column-blocked COO + its time on the graph twin
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "coo" > gpurun_out/coob_tests.log 2>&1; rc=$?; tail -3 gpurun_out/coob_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
for rows in 8192 4096 2048; do
for b in 16 32; do
  echo -n "rows_cap $rows: "
  SPMV_MI355X_COOB_ROWS=$rows timeout -k 10 200 python tools/run_one.py --workload soc-LiveJournal1 --format coo --opt col_blocks=$b --iters 50 2>&1 | grep -v amdgpu.ids || exit 1
done; done
