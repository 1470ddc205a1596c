cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ex in auto p2p; do
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 2 --backend gloo --scale 0.02 --steps 5 --warmup 2 --exchange $ex > gpurun_out/bench_gloo2_$ex.json 2> gpurun_out/bench_gloo2_$ex.err; echo "gloo2 exchange=$ex rc=$?"; tail -c 700 gpurun_out/bench_gloo2_$ex.json; grep -v "hostname\|amdgpu.ids\|Gloo" gpurun_out/bench_gloo2_$ex.err | tail -5
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 4 --backend gloo --scale 0.03 --steps 3 --warmup 1 > gpurun_out/bench_gloo4.json 2> gpurun_out/bench_gloo4.err; echo "gloo4 rc=$?"; tail -c 600 gpurun_out/bench_gloo4.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
