# HBM traffic of the dominant kernel from rocprofv3 PMC counters, one pass per counter group (TCC has 4 slots;
# FETCH_SIZE takes 3, WRITE_SIZE 2: /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: bash tools/collect_traffic.sh "<workload>:<format>:<dtype>[:opt=val,...] ..."   -> gpurun_out/traffic/<tag>/passN
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for spec in $1; do
  IFS=: read wl fmt dt opts <<< "$spec"
  optstr=""; for o in $(echo $opts | tr ',' ' '); do optstr="$optstr --opt $o"; done
  tag=$(echo $spec | tr ':,=' '___')
  mkdir -p gpurun_out/traffic/$tag
  timeout -k 5 300 python tools/run_one.py --workload $wl --format $fmt --dtype $dt $optstr --iters 50 --meta gpurun_out/traffic/$tag/meta.json > gpurun_out/traffic_${tag}_meta.log 2>&1
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 5 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/traffic/$tag/pass$i -- python tools/run_one.py --workload $wl --format $fmt --dtype $dt $optstr --iters 3 > gpurun_out/traffic_${tag}_pass$i.log 2>&1
    echo "traffic $spec pass$i rc=$?"
  done
done
