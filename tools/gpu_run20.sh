#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_solvers.py -x -q -m gpu 2>&1 | tail -5 &&
timeout -k 10 800 python tools/solver_bench.py --grid 160 --iters 300 --out gpurun_out/solver_bench_160.json 2> gpurun_out/solver_bench.log | tail -c 3000 &&
timeout -k 10 300 python tools/solver_bench.py --grid 40 --iters 300 --out gpurun_out/solver_bench_40.json 2>> gpurun_out/solver_bench.log | tail -c 3000
