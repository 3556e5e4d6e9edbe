#!/bin/bash
# bf16x3 / fp32 GMM stage timing per library variant
for v in "$@"; do
  if [ "$v" = "-" ]; then unset SSP_LIB_PATH; else export SSP_LIB_PATH=$PWD/tools/scratch/variants/$v.so; fi
  python bench.py --steps 3 --warmup 1 --stages mfcc,gmm --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$v', 'fp32', round(d['gmm']['roofline']['kernel_ms'],2), 'bf16x3', round(d['gmm_bf16x3']['roofline']['kernel_ms'],2), 'rescored', d['gmm_bf16x3'].get('utterances_rescored_in_fp32'), 'agree', d['gmm_bf16x3'].get('argmax_agreement_vs_fp32_path'))"
done
