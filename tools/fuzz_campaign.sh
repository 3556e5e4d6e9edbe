#!/bin/bash
# the randomised cross-checks on the shipped library, one after the other (GPU box): every tool's closing "OK" line -> gpurun_out/fuzz_campaign/summary.txt
#   tools/fuzz_campaign.sh [scale] [seed offset]      (scale 1 = about ten minutes)
S=${1:-1}
B=${2:-0}
O=gpurun_out/fuzz_campaign; mkdir -p $O; : > $O/summary.txt
run() {  # name, env assignments ("-" = none), script, seed, cases
  local name=$1 envs=$2 script=$3 seed=$4 cases=$5
  [ "$envs" = "-" ] && envs=""
  if env $envs timeout -k 10 900 python3 tools/$script $seed $cases > $O/$name.log 2>&1; then
    echo "$name | $script $seed $cases | $envs | $(tail -1 $O/$name.log)" | tee -a $O/summary.txt
  else
    echo "$name | $script $seed $cases | $envs | FAILED: $(tail -3 $O/$name.log | tr '\n' ' ')" | tee -a $O/summary.txt
    exit 1
  fi
}
for s in 61 62 63; do run mfcc_$s - fuzz_mfcc.py $((s + B)) $((1500 * S)); done
for s in 64 65; do run scoring_$s - fuzz_scoring.py $((s + B)) $((400 * S)); done
for s in 66 67 68; do run batch_$s - fuzz_mfcc_batch.py $((s + B)) $((200 * S)); done
run batch_junk05 "FUZZ_JUNK_FRAC=0.5 FUZZ_DIALECTS=sidekit,sidekit,inrepo" fuzz_mfcc_batch.py $((69 + B)) $((60 * S))
run batch_junk1 "FUZZ_JUNK_FRAC=1.0 FUZZ_DIALECTS=sidekit" fuzz_mfcc_batch.py $((70 + B)) $((30 * S))
for s in 71 72 73; do run scbatch_$s - fuzz_scoring_batch.py $((s + B)) $((150 * S)); done
for s in 74 75 76 77; do run hostfed_$s - fuzz_hostfed.py $((s + B)) $((60 * S)); done
timeout -k 10 600 python3 tools/soak.py "sidekit 39-d" "sidekit 26-d + scaling" "in-repo 512 / 256" > $O/soak.log 2>&1 && echo "soak | $(tail -1 $O/soak.log)" | tee -a $O/summary.txt
echo "campaign done"
