cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
for fmt in csr_stream sell_c_sigma csr_merge; do
  for set in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc/${fmt}/${tag} -- python tools/run_one.py --format $fmt --iters 3 > gpurun_out/pmc_${fmt}_${tag}.log 2>&1
  done
done
ls -R gpurun_out/pmc | head -50
