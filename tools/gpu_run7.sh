cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# 1. headline bench (default = nlpkkt240, csr_stream) and the same under rocprofv3 --kernel-trace --stats
timeout -k 10 900 python bench.py > gpurun_out/bench_r01_nlpkkt240.json 2> gpurun_out/bench_r01_nlpkkt240.err; echo "bench rc=$?"; cat gpurun_out/bench_r01_nlpkkt240.json
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python bench.py --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1; echo "rocprof rc=$?"
ls -R gpurun_out/prof_bench | head
# 2. traffic passes for the headline and the SELL comparison
bash tools/collect_traffic.sh "nlpkkt240:csr_stream:f64 nlpkkt240:sell_c_sigma:f64 nlpkkt240:csr_merge:f64"
