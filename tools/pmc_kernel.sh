#!/bin/bash
# SQ / LDS / MFMA counters of ONE kernel (name substring) from bench.py stages; run on the GPU box:
#   tools/pmc_kernel.sh <kernel-substring> <tag> <bench.py args ...>     -> gpurun_out/pmc_<tag>.txt
# (counter groups in separate passes, --pmc alone: no trace domains; PMC_EXTRA / PMC_EXTRA2 = further groups, e.g. instruction-cache counters)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT unset)}
kn=$1; tag=$2; shift; shift
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" ${PMC_EXTRA:+"$PMC_EXTRA"} ${PMC_EXTRA2:+"$PMC_EXTRA2"}; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmck_${tag}_$i
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmck_${tag}_$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-env "$@" > /dev/null 2>&1
done
python3 - > $R/gpurun_out/pmc_$tag.txt <<PY
import csv,glob,collections
kn='$kn'
print('== kernel substring', kn)
tot={}
for f in sorted(glob.glob('$R/gpurun_out/pmck_${tag}_*/*/*_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kn in r['Kernel_Name'] and '${PMC_ALSO:-}' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        tot[k]=sum(v)/len(v)
        print(k, '%.5g'%(sum(v)/len(v)), 'n=%d'%len(v))
if 'GRBM_GUI_ACTIVE' in tot and 'SQ_VALU_MFMA_BUSY_CYCLES' in tot:
    cyc=tot['GRBM_GUI_ACTIVE']/8.0   # summed over the 8 XCDs
    print('kernel cycles ~ %.4g; mfma busy fraction (of 1024 SIMDs x 4? see gmm_mfma_util.json) %.3f' % (cyc, tot['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024.0)))
PY
cat $R/gpurun_out/pmc_$tag.txt
