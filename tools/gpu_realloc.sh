#!/bin/bash
# gpurun -- tools/gpu_realloc.sh : tools/realloc_experiment.py twice (two processes on the same box)
set -o pipefail
mkdir -p gpurun_out
for k in 1 2; do
  timeout -k 10 400 python tools/realloc_experiment.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/realloc.txt || exit 1
  echo "-- next process" | tee -a gpurun_out/realloc.txt
done
