cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py > gpurun_out/bench_r01c.json 2> gpurun_out/bench_r01c.err; echo "bench rc=$?"; cat gpurun_out/bench_r01c.json
timeout -k 10 900 python bench.py --format csr_stream --no-cpu-baseline > gpurun_out/bench_r01c_csr.json 2> gpurun_out/bench_r01c_csr.err; echo "bench csr rc=$?"; cat gpurun_out/bench_r01c_csr.json
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench_selld -- python bench.py --no-cpu-baseline > gpurun_out/prof_bench_selld.log 2>&1; echo "rocprof rc=$?"
rm -rf gpurun_out/traffic
bash tools/collect_traffic.sh "nlpkkt240:sell_c_sigma:f64 nlpkkt240:csr_stream:f64 nlpkkt240:csr_merge:f64 nlpkkt240:sell_c_sigma:f64:sell_delta=2"
