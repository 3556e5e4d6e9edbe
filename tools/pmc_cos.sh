#!/bin/bash
# SQ counters for the cosine kernel (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT unset)}
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcc_$i -- python3 $R/bench.py --steps 1 --warmup 1 --stages mfcc,cosine --no-cpu-baseline --utts 2000 > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
for kn in ('cosine_reg_kernel',):
    print('==',kn)
    for f in sorted(glob.glob('$R/gpurun_out/pmcc_*/*/*_counter_collection.csv')):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kn in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items(): print(k, '%.4g'%(sum(v)/len(v)))
PY
