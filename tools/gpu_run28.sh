#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python tools/sweep.py --out gpurun_out/sweep_r01.json > gpurun_out/sweep_r01.log 2>&1; echo "sweep rc=$?"
timeout -k 10 600 python tools/sweep.py --workloads pwtk,cant --dtypes f32 --out gpurun_out/sweep_r01_f32.json > gpurun_out/sweep_r01_f32.log 2>&1; echo "sweep f32 rc=$?"
grep -E " auto| nt=0" gpurun_out/sweep_r01_f32.log | grep -E "VECTOR|SELLD" | cut -c1-120
