#!/bin/bash
# gpurun -- tools/gpu_traffic_f32.sh : HBM traffic (PMC passes) of the fp32 headline kernel
set -o pipefail
mkdir -p gpurun_out
rm -rf gpurun_out/traffic
bash tools/collect_traffic.sh "nlpkkt240:sell_c_sigma:f32"
python tools/collect_traffic.py gpurun_out/traffic gpurun_out/traffic_f32.json
