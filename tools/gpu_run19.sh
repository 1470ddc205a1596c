#!/bin/bash
# solver tests on the GPU
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_solvers.py -x -q -m gpu 2>&1 | tail -30 | tee gpurun_out/solver_tests.log
