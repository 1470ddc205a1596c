#!/bin/bash
# gather flavours of the column-blocked kernel (plain / nt / sc1) + PMC of the plain one
set -o pipefail
mkdir -p gpurun_out
for dbg in 0 2 3; do
  echo "DEBUG=$dbg: $(SPMV_MI355X_COOB_DEBUG=$dbg timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format coo --opt col_blocks=-1 --iters 30 2>&1 | tail -1)"
done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_coob3/pass$i -- python tools/run_one.py --workload soc-LiveJournal1 --format coo --opt col_blocks=-1 --iters 3 > gpurun_out/pmc_coob3_pass$i.log 2>&1
  echo "pass$i rc=$?"
done
python tools/parse_pmc.py gpurun_out/pmc_coob3/pass* 2>&1 | tail -40
