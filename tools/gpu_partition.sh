#!/bin/bash
# gpurun -- tools/gpu_partition.sh : the graph partition on the GPU box — (1) bench.py's multi-rank path at small scale
# (gloo, ranks share the one GPU: correctness only), (2) per-rank kernel times and exchange volumes of the 2- and 8-way
# partitions at full size, one rank at a time (tools/partition_probe.py), (3) the 4-rank full-size gloo rehearsal.
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 4 --backend gloo --steps 5 --warmup 2 --scale 0.02 > gpurun_out/part_gloo4_small.json 2> gpurun_out/part_gloo4_small.err \
  && tail -c 1500 gpurun_out/part_gloo4_small.json \
  && timeout -k 10 700 python tools/partition_probe.py --worlds ${PROBE_WORLDS:-2,8} --modes ${PROBE_MODES:-rows,graph,graph-original} --out gpurun_out/partition_probe.json > gpurun_out/partition_probe.log 2>&1 \
  && grep -v "rank" gpurun_out/partition_probe.log \
  && timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29518 \
  bench.py --gpus 4 --backend gloo --steps 3 --warmup 1 > gpurun_out/part_gloo4_full.json 2> gpurun_out/part_gloo4_full.err \
  && tail -c 2500 gpurun_out/part_gloo4_full.json
echo "rc=$?"
tail -5 gpurun_out/part_gloo4_small.err gpurun_out/partition_probe.log gpurun_out/part_gloo4_full.err 2>/dev/null | cut -c1-300
