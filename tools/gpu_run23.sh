#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python tools/sweep.py --workloads cant,scircuit,pwtk,nlpkkt240 --formats csr_vector --out gpurun_out/sweep_vec.json > gpurun_out/sweep_vec.log 2>&1
grep -v "^#" gpurun_out/sweep_vec.log | cut -c1-150 | tail -60
