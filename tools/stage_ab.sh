#!/bin/bash
for v in "$@"; do
  if [ "$v" = "-" ]; then unset SSP_LIB_PATH; else export SSP_LIB_PATH=$PWD/tools/scratch/variants/$v.so; fi
  python bench.py --steps 3 --warmup 1 --stages mfcc,cosine,em,dnn --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$v', 'cosine', round(d['cosine']['roofline']['kernel_ms'],3), 'em', json.dumps({k:v for k,v in d.get('gmm_em',d.get('em',{})).items() if 'ms' in k}), 'dnn', round(d['dvector_dnn']['kernel_ms'],3))"
done
