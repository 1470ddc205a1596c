#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { echo "$*: $(timeout -k 10 300 python tools/run_one.py "$@" --iters 200 2>&1 | tail -1)"; }
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_tests_trip9.log 2>&1; rc=$?
tail -4 gpurun_out/r02_gpu_tests_trip9.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_trip9.json 2> gpurun_out/r02_bench_trip9.err; echo "bench rc=$?"
tail -c 3500 gpurun_out/r02_bench_trip9.json; tail -5 gpurun_out/r02_bench_trip9.err
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 4 --backend gloo --steps 5 --warmup 2 > gpurun_out/r02_bench_gloo4_fullsize.json 2> gpurun_out/r02_bench_gloo4_fullsize.err; echo "gloo4 rc=$?"
tail -c 2500 gpurun_out/r02_bench_gloo4_fullsize.json; tail -3 gpurun_out/r02_bench_gloo4_fullsize.err
