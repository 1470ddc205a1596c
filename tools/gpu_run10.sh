cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/parity8.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/parity8.log; tail -3 gpurun_out/parity8.log
timeout -k 10 1000 python tools/sweep.py --out gpurun_out/sweep_r01.json > gpurun_out/sweep_r01.log 2>&1; echo "sweep rc=$?"
timeout -k 10 600 python tools/sweep.py --workloads pwtk,cant --dtypes f32 --out gpurun_out/sweep_r01_f32.json > gpurun_out/sweep_r01_f32.log 2>&1; echo "sweep f32 rc=$?"
timeout -k 10 600 python bench.py > gpurun_out/bench_r01b_nlpkkt240.json 2> gpurun_out/bench_r01b.err; cat gpurun_out/bench_r01b_nlpkkt240.json
