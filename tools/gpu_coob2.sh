#!/bin/bash
# gpurun -- tools/gpu_coob2.sh : column-block count scan of the column-blocked COO on the graph twin
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "coo" > gpurun_out/coob_tests.log 2>&1; rc=$?; tail -2 gpurun_out/coob_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
for b in -1 8 32 96 256; do
  timeout -k 10 200 python tools/run_one.py --workload soc-LiveJournal1 --format coo --opt col_blocks=$b --iters 50 2>&1 | grep -v amdgpu.ids || exit 1
done
