#!/bin/bash
# local driver: N fresh boxes, one tools/box_spread.sh row pair each, accumulated in profiles/r05_box_spread.jsonl
cd "$(dirname "$0")/.."
for i in $(seq ${1:-3}); do
  /usr/local/graft/bin/gpurun --timeout 600 -- 'tools/box_spread.sh - r3 > /dev/null' > /dev/null 2>&1
  python3 - <<'PY'
import json
seen = {l.strip() for l in open('profiles/r05_box_spread.jsonl')}
with open('profiles/r05_box_spread.jsonl', 'a') as f:
    for l in open('gpurun_out/box_spread.jsonl'):
        r = json.loads(l); r.pop('host', None); r.pop('ms_per_step', None)
        s = json.dumps(r)
        if s not in seen: f.write(s + "\n")
PY
done
wc -l profiles/r05_box_spread.jsonl
