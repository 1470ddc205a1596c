set -x
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 && \
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/parity2.log 2>&1 ; echo "pytest exit=$?" >> gpurun_out/parity2.log
for w in cant scircuit pwtk soc-LiveJournal1; do timeout -k 10 300 python bench.py --workload $w --steps 500 --warmup 100 --cpu-baseline-seconds 3 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; done
timeout -k 10 600 python bench.py > gpurun_out/bench_nlpkkt240.json 2> gpurun_out/bench_nlpkkt240.err
tail -3 gpurun_out/smoke.log; tail -3 gpurun_out/parity2.log; cat gpurun_out/bench_*.json
