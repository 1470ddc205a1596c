timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "csr_stream or row_blocks" > gpurun_out/parity5.log 2>&1; echo "pytest exit=$?" >> gpurun_out/parity5.log; tail -2 gpurun_out/parity5.log
timeout -k 10 900 python tools/sweep.py --workloads nlpkkt240,cant,pwtk,scircuit --formats csr_stream,sell_c_sigma --remap 1,3 --out gpurun_out/sweep6.json > gpurun_out/sweep6.log 2>&1
grep -v "^#" gpurun_out/sweep6.log
