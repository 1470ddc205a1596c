#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu --durations=6 > gpurun_out/fullsize.log 2>&1; rc=$?
tail -25 gpurun_out/fullsize.log
exit $rc
