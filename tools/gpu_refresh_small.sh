#!/bin/bash
# gpurun -- tools/gpu_refresh_small.sh : re-measure the three small twins after the 16-bit window indices — sweep (fp64 all
# three, fp32 cant/pwtk) and the HBM-side traffic of the kernels bench.py now picks for cant and pwtk.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python tools/sweep.py --workloads cant,scircuit,pwtk --out gpurun_out/sweep_small.json > gpurun_out/sweep_small.log 2>&1; echo "sweep rc=$?"
timeout -k 10 400 python tools/sweep.py --workloads pwtk,cant --dtypes f32 --out gpurun_out/sweep_small_f32.json > gpurun_out/sweep_small_f32.log 2>&1; echo "sweep f32 rc=$?"
rm -rf gpurun_out/traffic
bash tools/collect_traffic.sh "cant:csr_stream:f64 pwtk:csr_stream:f32 pwtk:csr_stream:f64"
python tools/collect_traffic.py gpurun_out/traffic gpurun_out/traffic_small.json
