#!/usr/bin/env python
"""SQ wavefront-cycle counters of a profiles/run_profile.sh tag that has a pmc_sq pass (tools/gpu_r5_profiles.sh adds one to
r5_twoend): mean per launch of the kernel whose name contains <substr>, and each counter as a fraction of SQ_WAVE_CYCLES
(SQ_WAIT_ANY = parked at s_waitcnt / barrier, SQ_WAIT_INST_ANY = stalled at issue, SQ_ACTIVE_INST_ANY = issuing)
-> profiles/<tag>/sq_counters.json.   Usage: python profiles/summarize_sq.py <tag> <substr>"""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, sub = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
acc = {}
with open(os.path.join(src, "pmc_sq", "bench_counter_collection.csv")) as f:
    for row in csv.DictReader(f):
        if sub in row["Kernel_Name"]:
            d = acc.setdefault(row["Counter_Name"], {})
            d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
out = {c: sum(d.values()) / len(d) for c, d in sorted(acc.items())}
w = out.get("SQ_WAVE_CYCLES")
if w:
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
        if c in out:
            out[c + "_frac_of_wave_cycles"] = out[c] / w
sha = os.path.join(src, "csrc_sha16.txt")
out["csrc_sha16"] = open(sha).read().strip() if os.path.isfile(sha) else None
json.dump(out, open(os.path.join(ROOT, "profiles", tag, "sq_counters.json"), "w"), indent=1)
print({k: (round(v, 3) if isinstance(v, float) and v < 10 else v) for k, v in out.items() if "frac" in k or k == "csrc_sha16"})
