#!/bin/bash
# HBM traffic counters of the MFCC fast kernel, separate passes (run on the GPU box): tools/pmc_hbm.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT unset)}
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/hbm_${tag}_$i -- python3 $R/bench.py --steps 2 --warmup 1 --stages mfcc --no-cpu-baseline "$@" > $R/gpurun_out/hbm_${tag}_$i.log 2>&1 || echo "pass $i failed (see gpurun_out/hbm_${tag}_$i.log)"
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob('$R/gpurun_out/hbm_${tag}_*/*/*_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'mfcc_fused' in r['Kernel_Name'] or 'mfcc_stream' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(k, '%.6g'%(sum(v)/len(v)), 'n=%d'%len(v))
PY
