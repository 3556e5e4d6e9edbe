#!/bin/bash
# on the GPU box: the default bench line + the rocprofv3 kernel summary of the same command, into gpurun_out/final/
set -e
mkdir -p gpurun_out/final
[ -n "$SKIP_BENCH" ] || python bench.py --steps 10 --warmup 3 > gpurun_out/final/bench.log 2>gpurun_out/final/bench.err
[ -n "$SKIP_BENCH" ] || tail -1 gpurun_out/final/bench.log > gpurun_out/final/bench_line.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof -o run -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final/prof.log 2>&1
find gpurun_out/final/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/final/kernel_stats.csv
find gpurun_out/final/prof -type f ! -name "*stats.csv" -delete
tail -1 gpurun_out/final/prof.log | cut -c1-300
