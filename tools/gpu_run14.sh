timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sell or footprint or row_blocks" > gpurun_out/parity10.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/parity10.log; tail -3 gpurun_out/parity10.log
if [ $rc -eq 0 ]; then
timeout -k 10 900 python tools/sweep.py --workloads cant,pwtk,scircuit,nlpkkt240 --formats sell_c_sigma --out gpurun_out/sweep10.json > gpurun_out/sweep10.log 2>&1
timeout -k 10 600 python tools/sweep.py --workloads pwtk,cant --dtypes f32 --formats sell_c_sigma,csr_stream >> gpurun_out/sweep10.log 2>&1
grep -v "^#" gpurun_out/sweep10.log
fi
