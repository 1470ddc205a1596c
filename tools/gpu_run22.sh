#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
echo "--- default env, library-side clamp"
timeout -k 10 600 python tools/solver_bench.py --grid 160 --iters 300 --host-iters 3 --out gpurun_out/solver_bench_160.json 2>&1 | grep -E "pcg|pbicg|flow" | cut -c1-260 &&
timeout -k 10 600 python tools/solver_bench.py --grid 40 --iters 300 --host-iters 3 --out gpurun_out/solver_bench_40.json 2>&1 | grep -E "pcg|pbicg|flow" | cut -c1-260 &&
grep -E "throttled" /sys/fs/cgroup/cpu.stat &&
timeout -k 10 600 python bench.py 2>/dev/null | tail -c 1500 &&
grep -E "throttled" /sys/fs/cgroup/cpu.stat
