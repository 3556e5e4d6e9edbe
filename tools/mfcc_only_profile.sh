#!/bin/bash
# on the GPU box: rocprofv3 kernel trace of the MFCC-only bench (3 warm-up + 10 timed launches of the fused kernel) + its own JSON line
set -e
mkdir -p gpurun_out/final_mfcc
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_mfcc/prof -o run -- python3 bench.py --steps 10 --warmup 3 --stages mfcc --no-cpu-baseline > gpurun_out/final_mfcc/bench.log 2> gpurun_out/final_mfcc/prof.log
tail -1 gpurun_out/final_mfcc/bench.log > gpurun_out/final_mfcc/bench_line.json
cp $(find gpurun_out/final_mfcc/prof -name "*kernel_trace.csv" | head -1) gpurun_out/final_mfcc/kernel_trace.csv
cp $(find gpurun_out/final_mfcc/prof -name "*kernel_stats.csv" | head -1) gpurun_out/final_mfcc/kernel_stats.csv
find gpurun_out/final_mfcc/prof -type f -delete
python3 - <<'PY'
import csv, json
rows = [r for r in csv.DictReader(open('gpurun_out/final_mfcc/kernel_trace.csv')) if 'mfcc_stream512' in r['Kernel_Name'] or 'mfcc_fused512' in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
j = json.load(open('gpurun_out/final_mfcc/bench_line.json'))
print("launches %d; all: %s" % (len(d), " ".join("%.3f" % x for x in d)))
print("rocprofv3 mean of the last 10 (the timed steps): %.3f ms; hipEvent mean inside bench.py (same run): %.3f ms" % (sum(d[-10:]) / 10, j['roofline']['kernel_ms']))
PY
