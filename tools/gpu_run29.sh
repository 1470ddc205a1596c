#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
for w in cant scircuit pwtk soc-LiveJournal1; do timeout -k 10 300 python bench.py --workload $w --steps 1000 --warmup 200 --cpu-baseline-seconds 3 > gpurun_out/bench_r01_$w.json 2> gpurun_out/bench_r01_$w.err; echo "bench $w rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/bench_r01_$w.json').read().strip().splitlines()[-1]); print(d['config']['format'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"; done
