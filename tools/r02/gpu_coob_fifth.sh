#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for dbg in 0 3 4; do
  echo "equi-depth DEBUG=$dbg: $(SPMV_MI355X_COOB_DEBUG=$dbg timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format coo --opt col_blocks=-1 --iters 30 2>&1 | tail -1)"
  echo "uniform99 DEBUG=$dbg: $(SPMV_MI355X_COOB_DEBUG=$dbg timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format coo --opt col_blocks=99 --iters 30 2>&1 | tail -1)"
done
echo "f32: $(timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format coo --dtype f32 --opt col_blocks=-1 --iters 30 2>&1 | tail -1)"
