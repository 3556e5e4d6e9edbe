"""Instruction census of a kernel's ISA (hipcc -S output): VALU issue cycles by class per basic block.
   python tools/isa_census.py file.s [kernel-substring]"""
import re, sys, collections
path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
lines = open(path).read().split("\n")
# find kernel body
start = 0
for i, l in enumerate(lines):
    if re.match(r"^_Z\S+:", l) and sub in l.split(":")[0]:
        start = i
        break
blocks = collections.OrderedDict()
cur = "entry"
blocks[cur] = []
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        cur = m.group(1) + (" " + t.split(";")[1].strip() if ";" in t else "")
        blocks[cur] = []
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    blocks[cur].append(t.split()[0])
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_pk_"): return "pk"
    if op.endswith("_dpp"): return "dpp"
    if re.match(r"v_(log|exp|rcp|rsq|sqrt|sin|cos)_", op): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("scratch_") or op.startswith("flat_"): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"
W = {"pk": 4, "dpp": 4, "trans": 8, "valu": 2, "mfma": 8}
tot = collections.Counter()
print("%-60s %5s %5s %5s %5s %5s %5s %5s %5s %6s" % ("block", "pk", "dpp", "trans", "valu", "mfma", "lds", "vmem", "salu", "cycles"))
for name, ops in blocks.items():
    c = collections.Counter(cls(o) for o in ops)
    cyc = sum(W.get(k, 0) * v for k, v in c.items())
    if sum(c.values()) >= 8:
        print("%-60s %5d %5d %5d %5d %5d %5d %5d %5d %6d" % (name[:60], c["pk"], c["dpp"], c["trans"], c["valu"], c["mfma"], c["lds"], c["vmem"], c["salu"], cyc))
