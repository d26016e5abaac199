import re, sys
f = sys.argv[1]
lines = open(f).read().split('\n')
blocks = []; cur = ['<entry>', []]
for ln in lines:
    m = re.match(r'^(\.LBB\d+_\d+):', ln)
    if m:
        blocks.append(cur); cur = [m.group(1), []]
    else:
        s = ln.strip()
        if s and not s.startswith(';') and not s.startswith('.'):
            cur[1].append(s.split(';')[0].strip())
blocks.append(cur)
for name, ins in blocks:
    nb = sum(1 for i in ins if i.startswith('s_barrier'))
    br = [i for i in ins if i.startswith('s_cbranch') or i.startswith('s_branch')]
    nd = sum(1 for i in ins if 'dpp' in i)
    if nb or len(ins) > 150:
        print(f'{name:12s} n={len(ins):5d} barriers={nb} dpp={nd} ds_r={sum(i.startswith("ds_read") for i in ins)} ds_w={sum(i.startswith("ds_write") for i in ins)} gl={sum(i.startswith("global_load") for i in ins)} gs={sum(i.startswith("global_store") for i in ins)}  {" | ".join(br)}')
