# Side-by-side PMC counters (SQ / TA / TCP / TCC) of one kernel: separate --pmc passes of a few counters each (the guide's slot limits),
# summarised by tools/parse_pmc.py. usage: bash tools/collect_pmc.sh <tag> <run_one.py arguments ...>   -> gpurun_out/pmc_<tag>.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; shift
rm -rf gpurun_out/pmc/$tag; mkdir -p gpurun_out/pmc/$tag
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc/$tag/pass$i -- python tools/run_one.py "$@" --iters 3 > gpurun_out/pmc_${tag}_pass$i.log 2>&1
  echo "pmc $tag pass$i rc=$?"
done
python tools/parse_pmc.py gpurun_out/pmc/$tag > gpurun_out/pmc_$tag.txt 2>&1
cat gpurun_out/pmc_$tag.txt
