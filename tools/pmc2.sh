#!/bin/bash
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT unset)}
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc2_${tag}_$i -- python3 $R/bench.py --steps 2 --warmup 1 --stages mfcc --no-cpu-baseline --utts 20000 "$@" > $R/gpurun_out/pmc2_${tag}_$i.log 2>&1 || echo "pass $i failed (see gpurun_out/pmc2_${tag}_$i.log)"
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob('$R/gpurun_out/pmc2_${tag}_*/*/*_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'mfcc_fused' in r['Kernel_Name'] or 'mfcc_stream' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(k, '%.4g'%(sum(v)/len(v)))
PY
