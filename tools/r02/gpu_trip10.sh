#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_tests_trip10.log 2>&1; rc=$?
tail -4 gpurun_out/r02_gpu_tests_trip10.log
exit $rc
