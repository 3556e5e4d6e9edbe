#!/bin/bash
# local driver: N fresh boxes, one tools/box_spread.sh row pair each, accumulated in profiles/r05_box_spread.jsonl
#   tools/box_more.sh [n] [jsonl]      (the final library's rows of round 5 went to profiles/r05_box_spread_final.jsonl)
cd "$(dirname "$0")/.."
export BOX_JSONL=${2:-profiles/r05_box_spread.jsonl}
touch $BOX_JSONL
for i in $(seq ${1:-3}); do
  /usr/local/graft/bin/gpurun --timeout 600 -- 'tools/box_spread.sh - r3 > /dev/null' > /dev/null 2>&1
  python3 - <<'PY'
import json, os
P = os.environ['BOX_JSONL']
seen = {l.strip() for l in open(P)}
with open(P, 'a') as f:
    for l in open('gpurun_out/box_spread.jsonl'):
        r = json.loads(l); r.pop('host', None); r.pop('ms_per_step', None)
        s = json.dumps(r)
        if s not in seen: f.write(s + "\n")
PY
done
wc -l $BOX_JSONL
