cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
spec=${1:-csr_stream}; WL=${2:-nlpkkt240}
fmt=${spec%%:*}; optstr=""; if [[ "$spec" == *:* ]]; then for o in $(echo ${spec#*:} | tr ',' ' '); do optstr="$optstr --opt $o"; done; fi
tagf=$(echo $spec | tr ':,=' '___')
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc2/${WL}_${tagf}/set$i -- python tools/run_one.py --workload $WL --format $fmt $optstr --iters 3 > gpurun_out/pmc2_${WL}_${tagf}_set$i.log 2>&1
  echo "done $spec set$i rc=$?"
done
