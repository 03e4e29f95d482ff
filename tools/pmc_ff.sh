#!/bin/bash
# PMC passes for the S1 (feed-forward + exp) kernel; writes csv under gpurun_out/pmc_ff
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ff/$tag -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
for f in glob.glob(R+'/gpurun_out/pmc_ff/*/*/*counter_collection.csv'):
    acc=collections.defaultdict(lambda: [0,0.0])
    for row in csv.DictReader(open(f)):
        k=(row['Kernel_Name'].split('(')[0][:40], row['Counter_Name'])
        acc[k][0]+=1; acc[k][1]+=float(row['Counter_Value'])
    for k,v in sorted(acc.items()):
        if 'k_ff' in k[0] or 'k_affine_lds' in k[0] or 'k_gru_lanes' in k[0]:
            print(k[0], k[1], v[0], v[1]/v[0])
PY
