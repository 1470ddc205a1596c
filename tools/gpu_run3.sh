set -x
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "csr_stream or sell or row_blocks" > gpurun_out/parity3.log 2>&1; echo "pytest exit=$?" >> gpurun_out/parity3.log
timeout -k 10 600 python tools/sweep.py --workloads cant,pwtk,nlpkkt240 --formats csr_stream,sell_c_sigma --out gpurun_out/sweep2.json > gpurun_out/sweep2.log 2>&1
tail -3 gpurun_out/parity3.log; grep -v "^#" gpurun_out/sweep2.log
