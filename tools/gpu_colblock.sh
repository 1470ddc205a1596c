#!/bin/bash
# gpurun -- tools/gpu_colblock.sh : tools/colblock_experiment.py
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python tools/colblock_experiment.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/colblock.txt
