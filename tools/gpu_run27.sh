#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('device convert', d['setup_s'], d['value'], d['roofline']['frac'])" &&
SPMV_MI355X_HOST_CONVERT=1 timeout -k 10 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('host convert  ', d['setup_s'], d['value'], d['roofline']['frac'])"
