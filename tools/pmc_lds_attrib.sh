#!/bin/bash
# LDS bank-conflict attribution of the headline MFCC kernel by ablation builds (GPU box): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE /
# SQ_INSTS_LDS of the first kernel per launch at configs[1], one rocprofv3 --pmc pass per library variant (tools/scratch/variants/<name>.so,
# '-' = the in-tree library).   tools/pmc_lds_attrib.sh <variant> ...   -> gpurun_out/lds_attrib.txt
# The round-6 attribution (profiles/r06_lds_attrib.txt) used, built here beforehand with tools/variant_flags.sh <name> mfcc_stream,mfcc_stream_walk <flags>:
#   psh0 = -DSSP_S_PSHIFT=0 (the round-5 P-row layout) | nomel = -DSSP_S_NOMEL (no piece filterbank) | pm64 = -DSSP_S_PM64 (partner through a b64 read)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/lds_attrib; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $ROOT/gpurun_out/lds_attrib.txt
for v in "$@"; do
  if [ "$v" = "-" ]; then unset SSP_LIB_PATH; else export SSP_LIB_PATH=$ROOT/tools/scratch/variants/$v.so; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/p_$v -- python3 $ROOT/bench.py --steps 2 --warmup 1 --stages mfcc --no-cpu-baseline --no-env --detail $O/d_$v.json > $O/$v.log 2>&1
  python3 - $v $(find $O/p_$v -name "*counter_collection.csv" | head -1) >> $ROOT/gpurun_out/lds_attrib.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    if "mfcc_stream512_kernel" in r["Kernel_Name"] and ", 0>(" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
q = 100000 * 75.0
print(sys.argv[1], " ".join("%s/quad=%.1f" % (k.replace("SQ_", ""), sum(v) / len(v) / q) for k, v in sorted(agg.items())), "launches", len(next(iter(agg.values()))) if agg else 0)
PY
  rm -rf $O/p_$v
done
cat $ROOT/gpurun_out/lds_attrib.txt
