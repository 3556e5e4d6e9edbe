"""VGPR bank census of a kernel's quad loop (hipcc -S output): for every vector instruction in blocks of loop depth >= 2, the number of
source-register pairs that fall into the same bank (register index mod 4), by instruction class.  A diagnostic for runs whose instruction
streams are equal up to register names but whose times differ.    python tools/bank_census.py file.s kernel-substring"""
import re, sys, collections
path, sub = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l) and sub in l)
depth = 0
stat = collections.defaultdict(lambda: [0, 0, 0])  # class -> [instructions, with a same-bank source pair, same-bank pairs]
for l in lines[start + 1:]:
    t = l.split(";")[0].strip()
    if t.startswith("s_endpgm"):
        break
    m = re.match(r"^\.LBB\d+_\d+:", t)
    if m:
        dm = re.search(r"Depth=(\d+)", l)
        depth = int(dm.group(1)) if dm else 0
        continue
    if depth < 2 or not t.startswith("v_"):
        continue
    op = t.split()[0]
    args = t[len(op):]
    regs = []  # per operand: list of registers
    for a in args.split(","):
        a = a.strip()
        m2 = re.match(r"^-?\|?v\[(\d+):(\d+)\]", a)
        m1 = re.match(r"^-?\|?v(\d+)\b", a)
        if m2:
            regs.append(list(range(int(m2.group(1)), int(m2.group(2)) + 1)))
        elif m1:
            regs.append([int(m1.group(1))])
        else:
            regs.append([])
    srcs = regs[1:]  # first operand = destination
    flat = []
    for i, r in enumerate(srcs):
        for x in r:
            flat.append((i, x))
    pairs = 0
    for i in range(len(flat)):
        for j in range(i + 1, len(flat)):
            if flat[i][0] != flat[j][0] and flat[i][1] != flat[j][1] and flat[i][1] % 4 == flat[j][1] % 4:
                pairs += 1
    cls = "mfma" if op.startswith("v_mfma") else "pk" if op.startswith("v_pk") else "dpp" if "dpp" in t else "valu"
    s = stat[cls]
    s[0] += 1
    s[1] += 1 if pairs else 0
    s[2] += pairs
tot = [sum(v[k] for v in stat.values()) for k in range(3)]
for k, v in sorted(stat.items()):
    print("%-5s instructions %5d  with a same-bank source pair %5d  pairs %5d" % (k, v[0], v[1], v[2]))
print("%-5s instructions %5d  with a same-bank source pair %5d  pairs %5d" % ("all", tot[0], tot[1], tot[2]))
