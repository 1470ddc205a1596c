timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sell or csr_stream or row_blocks or footprint" > gpurun_out/parity12.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/parity12.log; tail -3 gpurun_out/parity12.log
if [ $rc -eq 0 ]; then
for f in sell_c_sigma csr_stream; do python tools/run_one.py --format $f --iters 30 2>/dev/null | grep nlpkkt; done
for w in cant pwtk scircuit; do for f in sell_c_sigma csr_stream; do python tools/run_one.py --workload $w --format $f --iters 2000 2>/dev/null | grep $w; done; done
python tools/run_one.py --workload pwtk --dtype f32 --format sell_c_sigma --iters 2000 2>/dev/null | grep pwtk
python tools/run_one.py --workload pwtk --dtype f32 --format csr_stream --iters 2000 2>/dev/null | grep pwtk
fi
