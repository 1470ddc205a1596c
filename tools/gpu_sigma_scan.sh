#!/bin/bash
# gpurun -- tools/gpu_sigma_scan.sh : sort-window scan of the SELL-64-sigma-delta headline kernel (one process per value)
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/sigma_scan.log
for s in 0 4096 65536 262144 1048576 4194304; do
  o=""; [ $s -gt 0 ] && o="--opt sell_sigma=$s"
  timeout -k 10 200 python tools/run_one.py --workload nlpkkt240 --format sell_c_sigma --iters 50 $o >> gpurun_out/sigma_scan.log 2>&1 || { echo "sigma $s failed" >> gpurun_out/sigma_scan.log; break; }
done
grep -v "amdgpu.ids" gpurun_out/sigma_scan.log
