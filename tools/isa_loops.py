#!/usr/bin/env python
"""Instruction mix of the loops of one kernel in a hipcc -S listing (tuning aid).
usage: isa_loops.py file.s <substring of the kernel's mangled name> [min loop instructions]"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(("E", ":")) or
                 (l.startswith("_Z") and key in l and ":" in l))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels, instrs = {}, []
    for l in body:
        s = l.strip()
        if not s or s.startswith((";", ".")) and not re.match(r"\.LBB\d+_\d+:", s):
            continue
        m = re.match(r"(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if s.startswith(";") or s.endswith(":"):
            continue
        instrs.append(s.split(";")[0].strip())
    print("kernel %s: %d instructions" % (body[0].split(":")[0][:90], len(instrs)))
    for i, ins in enumerate(instrs):
        m = re.match(r"s_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", ins)
        if not m:
            continue
        tgt = labels.get(m.group(1) or m.group(2))
        if tgt is None or tgt > i or i - tgt < minlen:
            continue
        loop = instrs[tgt:i + 1]
        mix = collections.Counter()
        for x in loop:
            op = x.split()[0]
            if "dpp" in x and op.startswith("v_fma"): k = "v_fmac_dpp"
            elif op.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64")): k = "fp64 " + op[:9]
            elif op.startswith("v_permlane"): k = "permlane"
            elif op.startswith(("v_mov", "v_accvgpr")): k = op[:14]
            elif op.startswith("v_"): k = "valu other"
            elif op.startswith("ds_"): k = op
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): k = op
            elif op.startswith("s_waitcnt"): k = "s_waitcnt"
            elif op.startswith("s_nop"): k = "s_nop"
            else: k = "salu/other"
            mix[k] += 1
        print("loop at %d..%d: %d instructions" % (tgt, i, len(loop)))
        for k, v in sorted(mix.items(), key=lambda kv: -kv[1]):
            print("   %-22s %d" % (k, v))


if __name__ == "__main__":
    main()
