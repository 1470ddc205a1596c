#!/bin/bash
# gpurun -- tools/gpu_final.sh : end-of-round check on a fresh box — the whole GPU test tier, the 2-rank full-size gloo
# rehearsal of bench.py (largest per-rank blocks: 383 M non-zeros each), smoke(), and the default bench line.
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/final_tests.log 2>&1; rc=$?; tail -3 gpurun_out/final_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
  bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 > gpurun_out/part_gloo2_full.json 2> gpurun_out/part_gloo2_full.err \
  && tail -c 1800 gpurun_out/part_gloo2_full.json \
  && timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/final_smoke.log 2>&1 \
  && tail -2 gpurun_out/final_smoke.log \
  && timeout -k 10 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err \
  && tail -c 1500 gpurun_out/bench_final.json \
  && timeout -k 10 600 python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/bench_final_f32.json 2> gpurun_out/bench_final_f32.err \
  && tail -c 900 gpurun_out/bench_final_f32.json
echo "rc=$?"
grep "\[bench\]" gpurun_out/part_gloo2_full.err | cut -c1-300 | head -5
true
