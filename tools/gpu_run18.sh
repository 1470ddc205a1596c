cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tools/sweep.py --workloads cant,scircuit,pwtk,soc-LiveJournal1 --formats csr_vector,csr_stream,csr_merge,sell_c_sigma --cold --out gpurun_out/sweep_r01_cold.json > gpurun_out/sweep_r01_cold.log 2>&1; echo "cold sweep rc=$?"
grep -E "auto|nt=0" gpurun_out/sweep_r01_cold.log | grep -v "^#" | awk '{print}' | head -80
