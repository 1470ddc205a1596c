#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_solver
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_solver -o solver -- python tools/solver_bench.py --grid 160 --iters 300 --formats sell_c_sigma --host-iters 0 > gpurun_out/prof_solver.log 2>&1
echo rc=$?
find gpurun_out/prof_solver -name "*kernel_stats*" | head
f=$(find gpurun_out/prof_solver -name "*kernel_stats.csv" | head -1)
cut -c1-220 "$f" | head -30
# keep only the stats (traces are large)
find gpurun_out/prof_solver -name "*kernel_trace.csv" -size +20M -delete
