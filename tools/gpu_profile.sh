#!/bin/bash
# gpurun -- tools/gpu_profile.sh : rocprofv3 --kernel-trace --stats of the default bench.py run (the summary that goes to
# profiles/<round>_bench_nlpkkt240_*_kernel_stats.csv); no PMC here (tools/collect_traffic.sh does the counter passes)
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python bench.py --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1; echo "rocprof rc=$?"
f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-200 "$f" | head -8
find gpurun_out/prof_bench -name "*kernel_trace.csv" -size +20M -delete
tail -c 400 gpurun_out/prof_bench.log
