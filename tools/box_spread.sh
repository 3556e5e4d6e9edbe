#!/bin/bash
# one row per GPU box: the headline stage's kernel time next to what the box's clocks / power / calibration kernels say
#   tools/box_spread.sh [lib-or-'-' ...]   -> appends to gpurun_out/box_spread.jsonl   (run once per gpurun call = per fresh box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
[ $# -eq 0 ] && set -- -
for v in "$@"; do
  if [ "$v" = "-" ]; then unset SSP_LIB_PATH; else export SSP_LIB_PATH=$PWD/tools/scratch/variants/$v.so; fi
  python bench.py --full-line --stages mfcc --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys, socket, time
d = json.loads(sys.stdin.read())
e = d.get('env', {})
s = e.get('sustained_mfcc', {})
c = e.get('calibration_kernels', {})
row = {'time': time.strftime('%H:%M:%S'), 'host': socket.gethostname(), 'card': e.get('card'), 'lib': '$v', 'kernel_ms': d['roofline']['kernel_ms'], 'ms_per_step': d['ms_per_step'],
       'sustained_ms': e.get('sustained_mfcc_kernel_ms', {}).get('median'), 'sclk_mhz': s.get('sclk_mhz', {}).get('mean'), 'power_w': s.get('power_w', {}).get('mean'),
       'junction_c': s.get('junction_c', {}).get('mean'), 'copy_gbs': c.get('copy_gbs'), 'fma_tflops': c.get('fma_tflops'),
       'ms_per_step_normalised': d.get('value_normalised', {}).get('ms_per_step')}
print(json.dumps(row))
" >> gpurun_out/box_spread.jsonl
done
tail -$# gpurun_out/box_spread.jsonl
