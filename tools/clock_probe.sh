#!/bin/bash
# diagnostic (GPU box): engine clock and socket power WHILE a bench stage loops, sampled through rocm-smi next to the running process
#   tools/clock_probe.sh <stage> [steps] [stage-reps]   ->  gpurun_out/clock_<stage>.txt
# (bench.py's timed region is ~0.2 s; this repeats the stage for several seconds so that the governor's steady state is what is read)
cd "$(dirname "$0")/.."
stage=${1:-mfcc}; steps=${2:-600}
mkdir -p gpurun_out
out=gpurun_out/clock_$stage.txt
: > $out
rocm-smi --showclocks --showpower --showmaxpower >> $out 2>&1
echo "=== idle above; running stage $stage x $steps ===" >> $out
SSP_BENCH_STAGE_REPS=${3:-3} python3 bench.py --full-line --steps $steps --warmup 3 --stages $stage --no-cpu-baseline > gpurun_out/clock_$stage.line 2> gpurun_out/clock_$stage.err &
pid=$!
while kill -0 $pid 2>/dev/null; do
    date +%s.%N >> $out
    rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power|power" >> $out
    sleep 0.4
done
wait $pid; rc=$?
echo "bench rc=$rc" >> $out
tail -c 600 gpurun_out/clock_$stage.line >> $out
exit $rc
