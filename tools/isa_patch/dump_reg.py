#!/usr/bin/env python
"""Patch a kernel's ISA so that an existing `global_store_dwordx2 vADDR, vDATA, off` writes a register of interest instead:
after the OCC-th instruction matching ANCHOR inside block LABEL of function FUNC, v[REG:REG+1] (or, with --lane L, lane L of
its DPP row, through the dead temporaries v[TMP:TMP+1]) is copied into a[0:1] (the consumer wavefronts of these kernels use
no AGPRs), and the store matching STORE takes a[0:1] as its data.  The output array of that store then holds the register,
per step, with the kernel otherwise untouched (tools/isa_patch/repack.sh builds the library).
usage: dump_reg.py in.s out.s FUNC LABEL ANCHOR OCC REG STORE [--lane L --tmp TMP]"""
import re
import sys

a = sys.argv[1:]
lane = tmp = None
if "--lane" in a:
    i = a.index("--lane"); lane = int(a[i + 1]); del a[i:i + 2]
    i = a.index("--tmp"); tmp = int(a[i + 1]); del a[i:i + 2]
src, dst, func, label, anchor, occ, reg, store = a
occ, reg = int(occ), int(reg)
L = open(src).read().split("\n")
f0 = next(i for i, l in enumerate(L) if l.startswith(func + ":"))
b0 = next(i for i in range(f0, len(L)) if L[i].startswith(label + ":"))
b1 = next(i for i in range(b0 + 1, len(L)) if re.match(r"^\.LBB\d+_\d+:", L[i]))
out, n, done = L[:b0 + 1], 0, [0, 0]
for i in range(b0 + 1, b1):
    l = L[i]
    s = l.split(";")[0].strip()
    if re.search(store, s):
        m = re.match(r"(global_store_dwordx2 v\[\d+:\d+\]), v\[\d+:\d+\], off", s)
        out += ["\ts_nop 7", "\t" + m.group(1) + ", a[0:1], off"]
        done[1] += 1
        continue
    out.append(l)
    if re.search(anchor, s):
        n += 1
        if n == occ:
            out += ["\ts_waitcnt vmcnt(0) lgkmcnt(0)", "\ts_nop 7"]
            if lane is None:
                lo, hi = reg, reg + 1
            else:
                out += ["\tv_mov_b32_dpp v%d, v%d row_newbcast:%d row_mask:0xf bank_mask:0xf" % (tmp + k, reg + k, lane) for k in (0, 1)]
                out.append("\ts_nop 7")
                lo, hi = tmp, tmp + 1
            out += ["\tv_accvgpr_write_b32 a0, v%d" % lo, "\tv_accvgpr_write_b32 a1, v%d" % hi, "\ts_nop 7"]
            done[0] += 1
out += L[b1:]
assert done == [1, 1], "anchor / store matched %s times" % done
open(dst, "w").write("\n".join(out))
