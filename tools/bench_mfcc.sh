#!/bin/bash
# quick MFCC-only bench line: prints value, ms/step, roofline frac
python bench.py --full-line --steps ${STEPS:-10} --warmup 3 --stages mfcc --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('frames/s %.4g  ms/step %.3f  hbm frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
