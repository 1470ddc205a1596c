#!/bin/bash
# round 2, first GPU trip: the refactored create() + the new column-blocked layout (parity), then its timing on the soc-LiveJournal1 twin
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "coo or merge or caching or blocked or beta" > gpurun_out/r02_parity_coo.log 2>&1; rc=$?
tail -3 gpurun_out/r02_parity_coo.log
[ $rc -ne 0 ] && exit $rc
for cb in -1 74 148 37; do
  timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format coo --opt col_blocks=$cb --iters 30 2>&1 | tail -1
done
timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format csr_merge --iters 30 2>&1 | tail -1
timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format csr_merge --opt col_blocks=-2 --iters 30 2>&1 | tail -1
