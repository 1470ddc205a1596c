cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ov in 1 0; do
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --scale 0.02 --steps 5 --warmup 2 --overlap $ov > gpurun_out/bench_gloo2_ov$ov.json 2> gpurun_out/bench_gloo2_ov$ov.err; echo "gloo2 overlap=$ov rc=$?"; tail -c 1500 gpurun_out/bench_gloo2_ov$ov.json; tail -3 gpurun_out/bench_gloo2_ov$ov.err
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 4 --backend gloo --workload soc-LiveJournal1 --scale 0.05 --steps 3 --warmup 1 > gpurun_out/bench_gloo4.json 2> gpurun_out/bench_gloo4.err; echo "gloo4 rc=$?"; tail -c 800 gpurun_out/bench_gloo4.json; tail -3 gpurun_out/bench_gloo4.err
# reference-protocol driver on a twin and on a golden .mtx
./spmv-research_amd/bin/spmv_mi355x_bench --twin cant > gpurun_out/driver_cant.out 2> gpurun_out/driver_cant.csv; echo "driver rc=$?"; cat gpurun_out/driver_cant.out gpurun_out/driver_cant.csv
SPMV_MI355X_FORMAT=csr_stream ./spmv-research_amd/bin/spmv_mi355x_bench tests/golden/banded_symmetric.mtx > gpurun_out/driver_mtx.out 2> gpurun_out/driver_mtx.csv; echo "driver rc=$?"; tail -4 gpurun_out/driver_mtx.out; cat gpurun_out/driver_mtx.csv
