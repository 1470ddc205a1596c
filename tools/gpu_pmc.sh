# usage: bash tools/gpu_pmc.sh "<fmt[:opt=val,...]> ..." [workload] ; writes gpurun_out/pmc/<fmt>/<set>/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
FMTS=${1:-"csr_stream sell_c_sigma csr_merge"}
WL=${2:-nlpkkt240}
for spec in $FMTS; do
  fmt=${spec%%:*}; optstr=""; if [[ "$spec" == *:* ]]; then for o in $(echo ${spec#*:} | tr ',' ' '); do optstr="$optstr --opt $o"; done; fi
  tagf=$(echo $spec | tr ':,=' '___')
  i=0
  for set in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" \
             "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
             "GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
             "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
             "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum"; do
    i=$((i+1))
    timeout -k 5 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc/${WL}_${tagf}/set$i -- python tools/run_one.py --workload $WL --format $fmt $optstr --iters 3 > gpurun_out/pmc_${WL}_${tagf}_set$i.log 2>&1
    echo "done $spec set$i rc=$?"
  done
done
