#!/bin/bash
# gpurun -- tools/gpu_one_test.sh <pytest args> : a subset of the GPU tier (hard 300 s limit: new kernels run here first)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest "$@" -x -q -m gpu > gpurun_out/one_test.log 2>&1; rc=$?
tail -40 gpurun_out/one_test.log
exit $rc
