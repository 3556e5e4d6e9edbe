#!/bin/bash
# GPU box: socket power and engine clock while ONE instruction kind runs on every SIMD (issue_bench power mode), at 1 and 3 waves per SIMD.
#   tools/microbench/energy.sh > gpurun_out/energy.txt      (then: python tools/microbench/energy_report.py gpurun_out/energy.txt)
cd "$(dirname "$0")/../.."
B=tools/microbench/issue_bench
[ -x $B ] || { echo "build $B first (hipcc --offload-arch=gfx950 -O3 -o $B $B.hip)"; exit 1; }
#        SNOP VMOV FMA PKFMA PKADD PKMUL ADD DPPMOV LOGF DSR32 DSR64 DSR128 DSW32 DSW64 BPERM MFMA16 MFMA16_PK4 MFMA32BF MFMA32BF_PK4 CVTBF PERM32 SPLIT2
for op in ${OPS:-29   30   0   1     2     3     4   6      9    16    12    13     15    14    25    17     27 31 32 33 34 35}; do
  for wps in 1 3; do
    $B power $op $wps 3.2 > /tmp/eb.out 2>&1 &
    pid=$!
    sleep 1.4
    for k in 1 2 3; do
      rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Socket" | sed -e 's/.*sclk clock level: [^(]*(\([0-9]*\)Mhz)/sclk \1/' -e 's/.*Power (W): /watts /' | tr '\n' ' '
      echo
      sleep 0.45
    done
    wait $pid
    grep RESULT /tmp/eb.out
  done
done
