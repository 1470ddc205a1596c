#!/bin/bash
# gpurun -- tools/gpu_state_probe.sh : what state is this box's GPU in (clocks, power, partition modes) next to the
# headline kernel's time — the pool shows two timing states ~11 % apart; this collects what differs between them.
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/state_probe_$(date +%H%M%S).txt
{
  echo "== $(date -u +%FT%TZ)"
  rocm-smi --showperflevel --showmaxpower --showcomputepartition --showmemorypartition 2>&1 | grep "GPU\[0\]"
  echo "== idle"; rocm-smi --showclocks --showpower 2>&1 | grep "GPU\[0\]" | tr -s '\t ' ' ' | tr '\n' ';'; echo
  timeout -k 10 200 python tools/run_one.py --workload nlpkkt240 --format sell_c_sigma --iters 2000 > gpurun_out/state_run.txt 2>&1 &
  pid=$!
  for i in $(seq 1 40); do
    kill -0 $pid 2>/dev/null || break
    echo -n "t+$i "; rocm-smi --showclocks --showpower 2>&1 | grep "GPU\[0\]" | sed 's/GPU\[0\]\s*: //' | tr -s '\t ' ' ' | tr '\n' ';'; echo
    sleep 1
  done
  wait $pid
  grep -v amdgpu.ids gpurun_out/state_run.txt
} > $out 2>&1
cut -c1-260 $out
