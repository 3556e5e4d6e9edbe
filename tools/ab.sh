#!/bin/bash
# A/B timing of libsspgpu variants on one box:  tools/ab.sh <rounds> <lib-or-'-'>[:variant] ...   ('-' = the in-tree library; :N = --variant N)
# prints the per-variant list and median of the fused MFCC kernel's hipEvent time (ms) on configs[1]
rounds=$1; shift
declare -A acc
for r in $(seq $rounds); do
  for spec in "$@"; do
    v=${spec%%:*}; kv=0; [[ "$spec" == *:* ]] && kv=${spec##*:}
    if [ "$v" = "-" ]; then unset SSP_LIB_PATH; else export SSP_LIB_PATH=$PWD/tools/scratch/variants/$v.so; fi
    ms=$(python bench.py --full-line --steps ${STEPS:-15} --warmup 3 --stages mfcc --no-cpu-baseline --no-env --variant $kv 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
    acc[$spec]="${acc[$spec]} $ms"
  done
done
for v in "$@"; do echo "$v:${acc[$v]}  median $(echo ${acc[$v]} | tr ' ' '\n' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')"; done
