#!/bin/bash
# how much of the headline rests on the regularity of the nlpkkt240 twin: index modes switched off, rows jittered; PMC for three of them
set -o pipefail
mkdir -p gpurun_out/sens
b() { tag=$1; shift; timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --configs off "$@" > gpurun_out/sens/$tag.json 2> gpurun_out/sens/$tag.err; python - <<PY
import json
j=json.load(open("gpurun_out/sens/$tag.json"))
print("$tag", j["config"]["format"], "B/nnz", j["config"]["stored_bytes_per_nnz"], "ms", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"])
PY
}
b base
b modes1 --index-modes-off 1
b modes2 --index-modes-off 2
b modes3 --index-modes-off 3
b jit05 --jitter 0.05
b jit25 --jitter 0.25
b jit100 --jitter 1.0
b jit100_span64 --jitter 1.0 --jitter-span 64
b plain_sell --opt sell_delta=2
b csr_stream --format csr_stream
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
pmc() { tag=$1; shift
  mkdir -p gpurun_out/traffic/$tag
  timeout -k 5 300 python tools/run_one.py --workload nlpkkt240 --format sell_c_sigma --iters 30 --meta gpurun_out/traffic/$tag/meta.json "$@" > gpurun_out/traffic_${tag}_meta.log 2>&1
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum"; do
    i=$((i+1))
    timeout -k 5 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/traffic/$tag/pass$i -- python tools/run_one.py --workload nlpkkt240 --format sell_c_sigma --iters 3 "$@" > gpurun_out/traffic_${tag}_pass$i.log 2>&1
    echo "traffic $tag pass$i rc=$?"
  done
}
pmc nlpkkt240_sell_c_sigma_f64
SPMV_MI355X_SELL_MODES_OFF=3 pmc nlpkkt240_sell_c_sigma_f64_modesoff3
pmc nlpkkt240_sell_c_sigma_f64_jitter25 --jitter 0.25
python tools/collect_traffic.py gpurun_out/traffic gpurun_out/traffic_r02_sens.json
