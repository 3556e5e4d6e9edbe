#!/bin/bash
# On the GPU box: everything profiles/<round>_* is made from.   tools/profile_round.sh <round> [what ...]
#   what: bench | stats | stages | pmc512 | pmc2k   (default: all)
# Writes gpurun_out/<round>/...; tools/store_round.py <round> turns that into profiles/<round>_*.
R=${1:?round tag}; shift
WHAT=${*:-bench stats stages pmc512 pmc2k}
ROOT=$PWD
O=$ROOT/gpurun_out/$R
mkdir -p $O
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }

if has bench; then
  # the driver's own command line; the last stdout line is the compact line, the full result goes to --detail
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_detail.json > $O/bench.log 2> $O/bench.err && tail -1 $O/bench.log > $O/bench_line.json
  echo "bench: $(python3 -c "import json;d=json.load(open('$O/bench_line.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],'line bytes',len(open('$O/bench_line.json').read()))")"
fi
if has stats; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_all -o run -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-env --detail $O/prof_all_detail.json > $O/prof_all.log 2>&1
  cp $(find $O/prof_all -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv   # (per launch: the host-fed stages launch the headline kernel per slice, which the per-kernel average mixes in)
  cp $(find $O/prof_all -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv && rm -rf $O/prof_all
  echo "stats: $(wc -l < $O/kernel_stats.csv) rows"
fi
if has stages; then
  # one kernel trace per stage: the dominant kernel's launches can be read per launch (no other stage's launches of the same kernel in the file)
  for st in mfcc ref26 inrepo librosa gmm cosine plp; do
    extra=""; [ $st = gmm ] && extra="--no-gmm4-full"
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$st -o run -- python3 bench.py --steps 10 --warmup 3 --stages $st --no-cpu-baseline --no-env $extra --detail $O/stage_${st}_bench_detail.json > $O/stage_$st.log 2> $O/stage_$st.err
    tail -1 $O/stage_$st.log > $O/stage_${st}_bench_line.json
    cp $(find $O/prof_$st -name "*kernel_trace.csv" | head -1) $O/stage_${st}_kernel_trace.csv
    cp $(find $O/prof_$st -name "*kernel_stats.csv" | head -1) $O/stage_${st}_kernel_stats.csv
    rm -rf $O/prof_$st
    echo "stage $st: $(wc -l < $O/stage_${st}_kernel_trace.csv) dispatches"
  done
fi
pmc_pass() {  # tag, counters, bench args...
  local tag=$1 ctr=$2; shift; shift
  rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$tag -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-env --detail $O/pmc_${tag}_detail.json "$@" > $O/pmc_$tag.log 2>&1 || echo "pmc pass $tag failed"
  cp $(find $O/pmc_$tag -name "*counter_collection.csv" | head -1) $O/pmc_$tag.csv 2>/dev/null; rm -rf $O/pmc_$tag
}
if has pmc512; then
  cd /tmp
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
    i=$((i+1)); pmc_pass 512_$i "$c" --stages mfcc
  done
  cd $ROOT; echo "pmc512: $(ls $O/pmc_512_*.csv | wc -l) passes"
fi
if has pmc2k; then
  cd /tmp
  i=0
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    i=$((i+1)); pmc_pass 2k_$i "$c" --stages librosa
  done
  cd $ROOT; echo "pmc2k: $(ls $O/pmc_2k_*.csv | wc -l) passes"
fi
