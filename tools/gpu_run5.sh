timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/parity4.log 2>&1; echo "pytest exit=$?" >> gpurun_out/parity4.log; tail -2 gpurun_out/parity4.log
timeout -k 10 900 python tools/sweep.py --workloads nlpkkt240,cant,pwtk --formats csr_stream,sell_c_sigma,csr_vector,csr_merge --remap 0,2 --out gpurun_out/sweep4.json > gpurun_out/sweep4.log 2>&1
grep -v "^#" gpurun_out/sweep4.log
