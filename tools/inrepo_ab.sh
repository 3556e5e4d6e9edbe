#!/bin/bash
# in-repo MFCC stage under the stream kernel's launch options
run() { echo "== $*"; env "$@" python bench.py --steps 5 --warmup 2 --stages mfcc,inrepo --no-cpu-baseline --inrepo-variant ${IV:-3} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('headline', d['roofline']['kernel_ms'])
for k,v in d['mfcc_inrepo'].items(): print(k, v['roofline']['kernel'], round(v['roofline']['kernel_ms'],3), round(v['roofline']['frac'],3))"; }
IV=2 run A=1
run A=1
run SSP_STREAM_OCC=3
run SSP_STREAM_OCC=3 SSP_STREAM_WG_WAVES=1
run SSP_STREAM_OCC=3 SSP_STREAM_WG_WAVES=2
run SSP_STREAM_OCC=3 SSP_STREAM_WG_WAVES=3
run SSP_STREAM_OCC=2 SSP_STREAM_WG_WAVES=2
