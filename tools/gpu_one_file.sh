#!/bin/bash
# gpurun -- TEST=tests/<file>.py tools/gpu_one_file.sh : one test file of the GPU tier
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -m pytest ${TEST} -x -q -m gpu > gpurun_out/one_file.log 2>&1; rc=$?; tail -30 gpurun_out/one_file.log | cut -c1-400
exit $rc
