#!/bin/bash
# gpurun -- tools/gpu_partition_check.sh : partition GPU tests + bench.py multi-rank path (gloo ranks on one GPU) small and full size
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_gpu_partition.py -x -q -m gpu > gpurun_out/partition_tests.log 2>&1; rc=$?; tail -2 gpurun_out/partition_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 3 --backend gloo --steps 5 --warmup 2 --scale 0.02 > gpurun_out/part_gloo3_small.json 2> gpurun_out/part_gloo3_small.err \
  && tail -c 900 gpurun_out/part_gloo3_small.json \
  && timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29518 \
  bench.py --gpus 4 --backend gloo --steps 3 --warmup 1 > gpurun_out/part_gloo4_full.json 2> gpurun_out/part_gloo4_full.err \
  && tail -c 1300 gpurun_out/part_gloo4_full.json
echo "rc=$?"
grep "\[bench\]\|Error\|error" gpurun_out/part_gloo3_small.err gpurun_out/part_gloo4_full.err | cut -c1-300 | head -10
true
