#!/bin/bash
# gpurun -- bash tools/gpu_profiles.sh : the evidence that goes under profiles/ for a round:
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command (per-kernel durations of the headline and of configs 1-4)
#   2. the same bench line un-profiled (what the numbers are quoted from)
#   3. PMC traffic passes (FETCH_SIZE / WRITE_SIZE / TCC_EA0 request sizes / L2 hits) of the named and the best kernel of every config
set -o pipefail
ROUND=${1:-r03}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# PMC passes first: the bench lines that follow attach traffic records only when they were collected on the same kernel sources
if [ -z "$ONLY_BENCH" ]; then
  bash tools/collect_traffic.sh "nlpkkt240:sell_c_sigma:f64:placement=1 nlpkkt240:csr_stream:f64:placement=1 cant:sell_c_sigma:f64 cant:csr_vector:f64 cant:csr_stream:f64 cant:sell_c_sigma:f64:sym=1 cant:sell_c_sigma:f64:sym=2 scircuit:csr_vector:f64 scircuit:csr_vector:f64:lanes_per_row=64,rows_per_group=2 pwtk:sell_c_sigma:f32 pwtk:csr_stream:f32 pwtk:sell_c_sigma:f32:sym=1 pwtk:sell_c_sigma:f32:sym=2 soc-LiveJournal1:coo:f64:col_blocks=-1 soc-LiveJournal1:csr_merge:f64 soc-LiveJournal1:csr_merge:f64:col_blocks=-1"
  python tools/collect_traffic.py gpurun_out/traffic gpurun_out/traffic_${ROUND}.json
  cp gpurun_out/traffic_${ROUND}.json profiles/traffic_${ROUND}.json
fi
timeout -k 10 900 python bench.py > gpurun_out/${ROUND}_bench_default.json 2> gpurun_out/${ROUND}_bench_default.err; echo "bench rc=$?"
rm -rf gpurun_out/${ROUND}_prof_bench
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${ROUND}_prof_bench -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/${ROUND}_bench_profiled.json 2> gpurun_out/${ROUND}_bench_profiled.err; echo "profiled bench rc=$?"
cp $(ls -S gpurun_out/${ROUND}_prof_bench/*/*_kernel_stats.csv | head -1) gpurun_out/${ROUND}_bench_kernel_stats.csv
# the headline handle alone (no floor / jitter legs, no configs; pools without the array search): every launch of the headline kernel the
# profiler sees is this handle's — the 1000 timed steps, warm-up and settle phase, and the ~80 trial launches of the placement walk — so
# the kernel's average in the stats can be held against roofline.kernel_ms of the same run
rm -rf gpurun_out/${ROUND}_prof_headline
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${ROUND}_prof_headline -- python bench.py --headline-only --placement 1 --no-cpu-baseline > gpurun_out/${ROUND}_bench_headline_profiled.json 2> gpurun_out/${ROUND}_bench_headline_profiled.err; echo "profiled headline rc=$?"
cp $(ls -S gpurun_out/${ROUND}_prof_headline/*/*_kernel_stats.csv | head -1) gpurun_out/${ROUND}_headline_kernel_stats.csv
