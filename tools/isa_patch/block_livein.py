import re, sys
f, label = sys.argv[1], sys.argv[2]
L = open(f).read().split('\n')
b0 = next(i for i, l in enumerate(L) if l.startswith(label + ':'))
b1 = next(i for i in range(b0 + 1, len(L)) if re.match(r'^\.LBB\d+_\d+:', L[i]))
def regs(tok, pfx):
    out = []
    for m in re.finditer(r'\b%s\[(\d+):(\d+)\]|\b%s(\d+)\b' % (pfx, pfx), tok):
        if m.group(3) is not None: out.append(int(m.group(3)))
        else: out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    return out
written = {'v': set(), 's': set()}; livein = {'v': {}, 's': {}}
k = 0
for i in range(b0 + 1, b1):
    s = L[i].split(';')[0].strip()
    if not s or s.startswith('.'): continue
    parts = s.split(None, 1); op = parts[0]
    args = [a.strip() for a in parts[1].split(',')] if len(parts) > 1 else []
    # destination: first arg for most; stores/ds_write have no dest; swap writes both
    nodst = op.startswith(('ds_write', 'global_store', 's_cmp', 's_waitcnt', 's_barrier', 's_nop', 's_cbranch', 's_branch', 'v_cmp'))
    dst = [] if nodst or not args else [args[0]]
    srcs = args if nodst else args[1:]
    if 'swap' in op: dst = args[:2]; srcs = args[:2]
    if op.startswith('v_fmac') : srcs = args  # accumulates
    if op in ('s_addc_u32', 's_cselect_b32'): pass
    for pfx in 'vs':
        for a in srcs:
            for r in regs(a, pfx):
                if r not in written[pfx] and r not in livein[pfx]: livein[pfx][r] = (k, s)
    for pfx in 'vs':
        for a in dst:
            for r in regs(a, pfx): written[pfx].add(r)
    k += 1
for pfx in 'vs':
    print(pfx, 'live-in:')
    for r in sorted(livein[pfx]): print('  %s%d first read at %d: %s' % (pfx, r, livein[pfx][r][0], livein[pfx][r][1]))
