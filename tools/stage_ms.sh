#!/bin/bash
# per-stage kernel ms of libsspgpu variants on one box:  tools/stage_ms.sh <stages> <lib-or-'-'> ...   ('-' = the in-tree library)
stages=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset SSP_LIB_PATH; else export SSP_LIB_PATH=$PWD/tools/scratch/variants/$v.so; fi
  python bench.py --full-line --steps 5 --warmup 2 --stages $stages --no-cpu-baseline --no-gmm4-full 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
out={'mfcc': round(d['roofline']['kernel_ms'],3)}
for k in ('mfcc_ref26_cmvn','mfcc_librosa'):
    if k in d: out[k]=round(d[k]['roofline']['kernel_ms'],3)
if 'mfcc_inrepo' in d: out['inrepo']={t: round(x['roofline']['kernel_ms'],3) for t,x in d['mfcc_inrepo'].items() if isinstance(x,dict) and 'roofline' in x}
if 'plp' in d: out['plp_front']=round(d['plp']['front_ms'],3); out['plp_back']=round(d['plp']['back_ms'],3)
if 'gmm' in d: out['gmm']=round(d['gmm']['roofline']['kernel_ms'],2); out['gmm_bf16x3']=round(d['gmm_bf16x3']['roofline']['kernel_ms'],2)
print('$v', json.dumps(out))"
done
