#!/bin/bash
# gpurun -- tools/gpu_sweep.sh : every kernel x tunable on the five BASELINE.json twins (fp64) and pwtk/cant in fp32,
# rendered to profiles/sweep_r01.md by tools/sweep_md.py (copy gpurun_out/sweep_r01*.json back first)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python tools/sweep.py --out gpurun_out/sweep_r01.json > gpurun_out/sweep_r01.log 2>&1; echo "sweep rc=$?"
timeout -k 10 600 python tools/sweep.py --workloads pwtk,cant --dtypes f32 --out gpurun_out/sweep_r01_f32.json > gpurun_out/sweep_r01_f32.log 2>&1; echo "sweep f32 rc=$?"
python tools/sweep_md.py gpurun_out/sweep_r01.md gpurun_out/sweep_r01.json gpurun_out/sweep_r01_f32.json
