#!/bin/bash
# equi-depth column blocks: parity, then timing (plain / sc1 gathers; uniform blocks for comparison; barrier cadence)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "coo or merge or blocked" > gpurun_out/r02_parity_coo4.log 2>&1; rc=$?
tail -3 gpurun_out/r02_parity_coo4.log
[ $rc -ne 0 ] && exit $rc
for dbg in 0 3; do for sync in 1 4; do
  echo "equi-depth DEBUG=$dbg SYNC=$sync: $(SPMV_MI355X_COOB_DEBUG=$dbg SPMV_MI355X_COOB_SYNC=$sync timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format coo --opt col_blocks=-1 --iters 30 2>&1 | tail -1)"
done; done
echo "uniform 99: $(timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format coo --opt col_blocks=99 --iters 30 2>&1 | tail -1)"
echo "merge: $(timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format csr_merge --iters 30 2>&1 | tail -1)"
echo "f32: $(timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format coo --dtype f32 --opt col_blocks=-1 --iters 30 2>&1 | tail -1)"
