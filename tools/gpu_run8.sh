timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "csr_stream or row_blocks" > gpurun_out/parity6.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/parity6.log; tail -3 gpurun_out/parity6.log
if [ $rc -eq 0 ]; then
timeout -k 10 900 python tools/sweep.py --workloads nlpkkt240,cant,pwtk,scircuit,soc-LiveJournal1 --formats csr_stream,csr_vector --remap 0 --out gpurun_out/sweep7.json > gpurun_out/sweep7.log 2>&1
grep -v "^#" gpurun_out/sweep7.log
fi
