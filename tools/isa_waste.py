"""Scan hipcc -S output for library expansions that hide in hot loops: per kernel, the counts of opcodes that only appear in
expansions of sqrtf / division / pow / fp64 arithmetic / spills (the 2048-point kernel carried 290 such instructions per frame unnoticed).
   python tools/isa_waste.py file.s [min_total]"""
import re, sys, collections
path = sys.argv[1]
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pats = {"div": r"v_div_(scale|fmas|fixup)", "sqrt": r"v_sqrt_f", "rcp/rsq": r"v_(rcp|rsq)_", "class/ldexp/frexp": r"v_(cmp_class|ldexp|frexp)", "f64": r"_f64", "cndmask": r"v_cndmask",
        "scratch": r"scratch_", "mul_lo/hi": r"v_mul_(lo|hi)_", "64-bit add": r"v_lshl_add_u64|v_addc_co", "exp/log": r"v_(exp|log)_f", "cvt": r"v_cvt_"}
cur = None
cnt = collections.OrderedDict()
tot = collections.Counter()
for l in open(path):
    m = re.match(r"^(_Z\S+):", l)
    if m:
        cur = m.group(1)
        cnt[cur] = collections.Counter()
        continue
    t = l.strip()
    if cur is None or not t or t.startswith((";", ".")):
        continue
    op = t.split()[0]
    if op == "s_endpgm":
        cur = None
        continue
    tot[cur] += 1
    for k, p in pats.items():
        if re.search(p, op):
            cnt[cur][k] += 1
import subprocess
for k, c in cnt.items():
    if sum(c.values()) >= thr:
        try:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:110]
        except OSError:
            name = k[:110]
        print("%-110s %6d instr | %s" % (name, tot[k], ", ".join("%s %d" % kv for kv in c.most_common())))
