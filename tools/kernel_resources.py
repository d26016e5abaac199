"""Registers / scratch / LDS of every kernel in an AMDGPU assembly file (hipcc -save-temps or --cuda-device-only -S).
Usage: python tools/kernel_resources.py file.s [name filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if flt in name:
        print("%-90s vgpr %s agpr %s sgpr %s scratch %s spill %s lds %s" % (
            name, g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("private_segment_fixed_size"),
            g("vgpr_spill_count"), g("group_segment_fixed_size")))
