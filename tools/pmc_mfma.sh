#!/bin/bash
# MFMA-pipe counters of the two scoring kernels at FULL size (configs[2]: 100000 utterances x 298 frames x 51 models x 64 mixtures;
# configs[4]: 1e6 x 1251 x 256), one counter group per pass (rocprofv3 --pmc only: no trace domains beside it).  Run on the GPU box:
#   tools/pmc_mfma.sh            -> gpurun_out/pmc_mfma/{gmm,cos}_<i>.csv ; tools/store_mfma_pmc.py turns them into profiles/*_mfma_util.json
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/pmc_mfma
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/g_$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --stages mfcc,gmm --no-cpu-baseline --no-env --detail $O/gmm_detail_$i.json > $O/gmm_$i.log 2>&1 || echo "gmm pass $i failed"
  cp $(find $O/g_$i -name "*counter_collection.csv" | head -1) $O/gmm_$i.csv 2>/dev/null; rm -rf $O/g_$i
  echo "gmm pass $i: $(wc -l < $O/gmm_$i.csv) rows"
  rocprofv3 --pmc $c --output-format csv -d $O/c_$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --stages mfcc,cosine --utts 2000 --no-cpu-baseline --no-env --detail $O/cos_detail_$i.json > $O/cos_$i.log 2>&1 || echo "cosine pass $i failed"
  cp $(find $O/c_$i -name "*counter_collection.csv" | head -1) $O/cos_$i.csv 2>/dev/null; rm -rf $O/c_$i
  echo "cos pass $i: $(wc -l < $O/cos_$i.csv) rows"
done
