#!/bin/bash
# gpurun -- tools/gpu_window_check.sh : parity of the CSR kernels + the bench lines of the two window-kernel configs
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/window_tests.log 2>&1; rc=$?; tail -2 gpurun_out/window_tests.log
[ $rc -eq 0 ] || exit $rc
for w in pwtk cant; do
  timeout -k 10 300 python bench.py --workload $w > gpurun_out/bench_r01_$w.json 2> gpurun_out/bench_r01_$w.err || exit 1
  python -c "
import json,sys
j=json.loads(open('gpurun_out/bench_r01_$w.json').read().strip().splitlines()[-1])
print(j['config']['format'], j['dtype'], j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['roofline']['traffic'])"
done
timeout -k 10 200 python tools/run_one.py --workload pwtk --format csr_stream --dtype f64 --iters 50
