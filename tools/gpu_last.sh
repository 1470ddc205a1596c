#!/bin/bash
# gpurun -- tools/gpu_last.sh : whole GPU tier, then the graph-twin bench line with its new default kernel
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/final_tests.log 2>&1; rc=$?; tail -4 gpurun_out/final_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload soc-LiveJournal1 > gpurun_out/bench_r01_soc-LiveJournal1.json 2> gpurun_out/bench_r01_soc.err && tail -c 1200 gpurun_out/bench_r01_soc-LiveJournal1.json
