#!/bin/bash
# gpurun -- tools/gpu_bench_all.sh : bench.py on every BASELINE.json config (default = nlpkkt240 last), one JSON line each
set -o pipefail
mkdir -p gpurun_out
for w in cant scircuit pwtk soc-LiveJournal1; do
	timeout -k 10 300 python bench.py --workload $w --steps 1000 --warmup 200 --cpu-baseline-seconds 3 > gpurun_out/bench_r01_$w.json 2> gpurun_out/bench_r01_$w.err; echo "bench $w rc=$?"
done
timeout -k 10 600 python bench.py > gpurun_out/bench_r01_nlpkkt240.json 2> gpurun_out/bench_r01_nlpkkt240.err; echo "bench nlpkkt240 rc=$?"
tail -c 600 gpurun_out/bench_r01_nlpkkt240.json
