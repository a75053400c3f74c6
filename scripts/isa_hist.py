"""Opcode histogram of one kernel in a hipcc -S file: whole kernel and its longest loop body.
usage: isa_hist.py file.s mangled_substring"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().split(':')[0].endswith('E'))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start + 1:end]
labels = {}
ins = []
for l in body:
    t = l.strip()
    if not t or t.startswith((';', '.')) and not re.match(r'^\.LBB\w+:', t):
        continue
    m = re.match(r'^(\.LBB\w+):', t)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    ins.append(t)
# back-branches
best = (0, 0, 0)
for i, t in enumerate(ins):
    m = re.match(r's_cbranch\w*\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)', t)
    if m:
        tgt = labels.get(m.group(1) or m.group(2))
        if tgt is not None and tgt < i and i - tgt > best[0]:
            best = (i - tgt, tgt, i)
def hist(seq, title):
    c = collections.Counter(re.split(r'\s+', t)[0] for t in seq)
    groups = collections.Counter()
    for op, n in c.items():
        g = ('dpp' if False else op)
        groups[g] += n
    dpp = sum(1 for t in seq if 'row_shr' in t or 'row_shl' in t or 'wave_sh' in t or 'row_bcast' in t or 'quad_perm' in t or 'dpp' in t)
    print('==', title, 'total', len(seq), 'valu', sum(n for o, n in c.items() if o.startswith('v_')), 'salu', sum(n for o, n in c.items() if o.startswith('s_')), 'dpp-mod', dpp)
    for op, n in c.most_common(45):
        print('  %-28s %d' % (op, n))
hist(ins, 'kernel')
hist(ins[best[1]:best[2] + 1], 'longest loop')
