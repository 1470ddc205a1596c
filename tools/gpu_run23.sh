#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "csr_stream" > gpurun_out/win_tests.log 2>&1; rc=$?
tail -15 gpurun_out/win_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python tools/sweep.py --workloads cant,scircuit,pwtk --formats csr_stream --out gpurun_out/sweep_win.json > gpurun_out/sweep_win.log 2>&1
grep -v "^#" gpurun_out/sweep_win.log | grep -E "WINDOW|auto|STREAMD_r(4|8)_" | cut -c1-150
timeout -k 10 900 python tools/sweep.py --workloads pwtk,cant --dtypes f32 --formats csr_stream --out gpurun_out/sweep_win32.json > gpurun_out/sweep_win32.log 2>&1
grep -v "^#" gpurun_out/sweep_win32.log | grep -E "WINDOW" | cut -c1-150
