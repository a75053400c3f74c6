"""VALU pipe cycles per loop iteration of a kernel in a hipcc -S file, by instruction class, with the per-class costs
measured by scripts/valu_probe.hip on MI355X (cycles per wave64 instruction at >= 2 waves per SIMD: plain 2.7, packed
fp32 4.2, DPP 4.3, transcendental 8, 32-bit integer multiply 8).  usage: valu_cost.py file.s mangled_substring"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().split(':')[0].endswith('E'))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
labels, ins = {}, []
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith((';', '.')) and not re.match(r'^\.LBB\w+:', t):
        continue
    m = re.match(r'^(\.LBB\w+):', t)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    ins.append(t)
best = (0, 0, 0)
for i, t in enumerate(ins):
    m = re.match(r's_cbranch\w*\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)', t)
    if m:
        tgt = labels.get(m.group(1) or m.group(2))
        if tgt is not None and tgt < i and i - tgt > best[0]:
            best = (i - tgt, tgt, i)
loop = ins[best[1]:best[2] + 1]
c = collections.Counter(re.split(r'\s+', t)[0] for t in loop)
cat, cnt = collections.Counter(), collections.Counter()
for op, n in c.items():
    if not op.startswith('v_'):
        continue
    if op.startswith('v_pk_'): k, cy = 'packed', 4.2
    elif 'dpp' in op: k, cy = 'dpp', 4.3
    elif op.startswith(('v_rcp', 'v_exp', 'v_log', 'v_sqrt', 'v_rsq', 'v_sin', 'v_cos')): k, cy = 'transcendental', 8
    elif op.startswith(('v_mul_lo', 'v_mul_hi')): k, cy = 'mul32', 8
    elif op.startswith('v_mov') or op.startswith('v_accvgpr'): k, cy = 'mov', 2.7
    elif op.startswith(('v_cndmask', 'v_cmp')): k, cy = 'cmp/select', 2.7
    elif op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): k, cy = 'lane', 2.7
    else: k, cy = 'plain', 2.7
    cat[k] += n * cy
    cnt[k] += n
tot = sum(cat.values())
print('loop: %d instructions, %d VALU, ~%.0f VALU pipe cycles per iteration' % (len(loop), sum(cnt.values()), tot))
for k, v in cat.most_common():
    print('  %-15s %4d instr  %6.0f cycles  %4.1f%%' % (k, cnt[k], v, 100 * v / tot))
