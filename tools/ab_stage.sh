#!/bin/bash
# A/B timing of libsspgpu variants on one bench stage:  tools/ab_stage.sh <rounds> <stage> <json-path> <lib-or-'-'> ...
#   e.g. tools/ab_stage.sh 3 librosa mfcc_librosa.roofline.kernel_ms - regtab8     ('-' = the in-tree library)
rounds=$1; stage=$2; path=$3; shift; shift; shift
declare -A acc
for r in $(seq $rounds); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset SSP_LIB_PATH; else export SSP_LIB_PATH=$PWD/tools/scratch/variants/$v.so; fi
    ms=$(python bench.py --full-line --steps ${STEPS:-5} --warmup 1 --stages $stage --no-cpu-baseline --no-env ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in '$path'.split('.'): d=d[k]
print('%.3f' % d)")
    acc[$v]="${acc[$v]} $ms"
  done
done
for v in "$@"; do echo "$v:${acc[$v]}  median $(echo ${acc[$v]} | tr ' ' '\n' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')"; done
