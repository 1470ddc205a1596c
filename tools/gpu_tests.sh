#!/bin/bash
# gpurun -- tools/gpu_tests.sh : the whole GPU test tier in one process (log: gpurun_out/gpu_tests.log)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -5 gpurun_out/gpu_tests.log
exit $rc
