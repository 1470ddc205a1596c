#!/bin/bash
# gpurun -- tools/gpu_traffic_coob.sh : HBM-side traffic (PMC passes) of the column-blocked COO on the graph twin
set -o pipefail
mkdir -p gpurun_out
rm -rf gpurun_out/traffic
bash tools/collect_traffic.sh "soc-LiveJournal1:coo:f64:col_blocks=-1"
python tools/collect_traffic.py gpurun_out/traffic gpurun_out/traffic_coob.json
