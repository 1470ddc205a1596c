#!/bin/bash
# gpurun -- tools/gpu_partition_full.sh : the 4-rank full-size gloo rehearsal of bench.py alone (ranks share the one GPU)
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${RANKS:-4} --master-addr 127.0.0.1 --master-port 29518 \
  bench.py --gpus ${RANKS:-4} --backend gloo --steps 3 --warmup 1 $BENCH_ARGS > gpurun_out/part_gloo4_full.json 2> gpurun_out/part_gloo4_full.err
echo "rc=$?"
tail -c 2500 gpurun_out/part_gloo4_full.json
grep "\[bench\]" gpurun_out/part_gloo4_full.err | cut -c1-400 | head -20
true
