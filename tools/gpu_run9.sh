timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/parity7.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/parity7.log; tail -3 gpurun_out/parity7.log
timeout -k 10 900 python tools/sweep.py --workloads nlpkkt240,cant,pwtk,scircuit,soc-LiveJournal1 --formats csr_stream --out gpurun_out/sweep8.json > gpurun_out/sweep8.log 2>&1
grep -v "^#" gpurun_out/sweep8.log
