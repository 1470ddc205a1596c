#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python tools/xcd_experiment.py --stencil 280 --chunks 64,1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/xcd_experiment_stencil.log
