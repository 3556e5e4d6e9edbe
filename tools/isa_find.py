"""diagnostic: list where given instruction patterns sit in one kernel of an assembly listing (loop depth from the compiler's block comments)
   python tools/isa_find.py <file.s> <kernel-name-substring> <regex> [...]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and re.match(r'^_Z\S+:', l)][0]
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
depth = 0
pats = [re.compile(p) for p in sys.argv[3:]]
for i, l in enumerate(lines[start:end]):
    m = re.search(r'Depth=(\d+)', l)
    if l.startswith('.LBB'):
        depth = int(m.group(1)) if m else 0
    if any(p.search(l) for p in pats):
        print(i, 'depth', depth, l.strip()[:110])
print('instructions:', sum(1 for l in lines[start:end] if l.startswith('\t') and not l.strip().startswith(('.', ';'))))
