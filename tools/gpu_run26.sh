#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "symmetric or keep_symmetry" > gpurun_out/sym_tests.log 2>&1; rc=$?
tail -30 gpurun_out/sym_tests.log
exit $rc
