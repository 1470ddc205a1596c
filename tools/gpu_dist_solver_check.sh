#!/bin/bash
# gpurun -- tools/gpu_dist_solver_check.sh : the distributed solver tests (gloo ranks sharing the one GPU), both partitions
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_gpu_dist_solver.py -x -q -m gpu > gpurun_out/dist_solver_tests.log 2>&1; rc=$?; tail -25 gpurun_out/dist_solver_tests.log | cut -c1-300
exit $rc
