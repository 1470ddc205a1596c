#!/bin/bash
# gpurun -- SEEDS=n tools/gpu_stress_run.sh : randomised kernel stress, then the solver stress (progress in gpurun_out/stress.log)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python tests/stress/gpu_stress.py ${SEEDS:-6} > gpurun_out/stress.log 2>&1; rc=$?; tail -3 gpurun_out/stress.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tests/stress/gpu_solver_stress.py > gpurun_out/solver_stress.log 2>&1; rc=$?; tail -3 gpurun_out/solver_stress.log
exit $rc
