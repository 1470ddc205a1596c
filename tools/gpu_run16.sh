cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/parity11.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/parity11.log; tail -3 gpurun_out/parity11.log
timeout -k 10 1000 python tools/sweep.py --out gpurun_out/sweep_r01.json > gpurun_out/sweep_r01.log 2>&1; echo "sweep rc=$?"
timeout -k 10 600 python tools/sweep.py --workloads pwtk,cant --dtypes f32 --out gpurun_out/sweep_r01_f32.json > gpurun_out/sweep_r01_f32.log 2>&1; echo "sweep f32 rc=$?"
for w in cant scircuit pwtk soc-LiveJournal1; do timeout -k 10 300 python bench.py --workload $w --steps 1000 --warmup 200 --cpu-baseline-seconds 3 > gpurun_out/bench_r01_$w.json 2> gpurun_out/bench_r01_$w.err; echo "bench $w rc=$?"; done
