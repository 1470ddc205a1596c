#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for dbg in 4 5; do
  echo "f64 equi-depth DEBUG=$dbg: $(SPMV_MI355X_COOB_DEBUG=$dbg timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format coo --opt col_blocks=-1 --iters 30 2>&1 | tail -1)"
done
for dbg in 0 1 4 5; do
  echo "f32 equi-depth DEBUG=$dbg: $(SPMV_MI355X_COOB_DEBUG=$dbg timeout -k 10 300 python tools/run_one.py --workload soc-LiveJournal1 --format coo --dtype f32 --opt col_blocks=-1 --iters 30 2>&1 | tail -1)"
done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_coob_f32 -- python tools/run_one.py --workload soc-LiveJournal1 --format coo --dtype f32 --opt col_blocks=-1 --iters 3 > gpurun_out/kt_coob_f32.log 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/kt_coob_f32/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'coo_blocked' in r['Kernel_Name']:
            print({k:r[k] for k in r if k in ('Kernel_Name','LDS_Block_Size','Workgroup_Size','Grid_Size','VGPR_Count','Start_Timestamp','End_Timestamp')}, int(r['End_Timestamp'])-int(r['Start_Timestamp']))
            break
PY
