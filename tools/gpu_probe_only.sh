#!/bin/bash
# gpurun -- tools/gpu_probe_only.sh : tools/partition_probe.py alone (PROBE_WORLDS, PROBE_MODES)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python tools/partition_probe.py --worlds ${PROBE_WORLDS:-8,2} --modes ${PROBE_MODES:-graph-original-rows} --out gpurun_out/partition_probe.json > gpurun_out/partition_probe.log 2>&1
echo "rc=$?"; grep -v "amdgpu.ids" gpurun_out/partition_probe.log | cut -c1-330
