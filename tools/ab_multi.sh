#!/bin/bash
# A/B/A... timing of libsspgpu variants over several bench stages in ONE box:  tools/ab_multi.sh <rounds> <stages> <lib-or-'-'> ...
#   ('-' = the in-tree library; other names are tools/scratch/variants/<name>.so).  One bench.py run per (round, variant), the variants
#   alternate inside a round; per variant and stage the list of kernel times (hipEvent averages over STEPS launches) and their median.
rounds=$1; stages=$2; shift; shift
out=${AB_OUT:-gpurun_out/ab_multi.txt}
mkdir -p "$(dirname $out)"
: > $out.raw
for r in $(seq $rounds); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset SSP_LIB_PATH; else export SSP_LIB_PATH=$PWD/tools/scratch/variants/$v.so; fi
    python bench.py --full-line --steps ${STEPS:-10} --warmup 2 --stages $stages --no-cpu-baseline --no-env 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
paths = {'mfcc': 'roofline.kernel_ms', 'ref26': 'mfcc_ref26_cmvn.roofline.kernel_ms', 'inrepo16k': 'mfcc_inrepo.16k.roofline.kernel_ms',
         'inrepo8k': 'mfcc_inrepo.8k.roofline.kernel_ms', 'librosa': 'mfcc_librosa.roofline.kernel_ms', 'plp': 'plp.front_roofline.kernel_ms'}
for k, p in paths.items():
    o = d
    try:
        for s in p.split('.'): o = o[s]
        print('$v', k, '%.3f' % o)
    except (KeyError, TypeError):
        pass
" >> $out.raw
    echo "round $r $v done" >&2
  done
done
python - $out.raw "$@" > $out <<'EOF'
import sys, statistics
rows = [l.split() for l in open(sys.argv[1])]
names = sys.argv[2:]
stages = []
for r in rows:
    if r[1] not in stages: stages.append(r[1])
print("%-14s" % "variant" + "".join("%-12s" % s for s in stages))
for n in names:
    print("%-14s" % n + "".join("%-12.3f" % statistics.median([float(r[2]) for r in rows if r[0] == n and r[1] == s] or [float('nan')]) for s in stages))
print()
for n in names:
    for s in stages:
        print(n, s, " ".join(r[2] for r in rows if r[0] == n and r[1] == s))
EOF
cat $out
