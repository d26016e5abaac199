#!/usr/bin/env python
"""Where the GPU time of a profiled training step goes, by kernel family, from the --stats table of a
tools/prof_generic.sh run: profiles/<tag>/kernel_stats.csv -> profiles/<tag>/breakdown.json.  The number of steps is the
call count of the kernel named by <once-per-step substr> (default: lds_forward_pair_kernel, one launch per step of
make_gradfun).   Usage: python profiles/summarize_breakdown.py <tag> [once-per-step substr]"""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
once = sys.argv[2] if len(sys.argv) > 2 else "lds_forward_pair_kernel"
rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", tag, "kernel_stats.csv"))))
steps = sum(int(r["Calls"]) for r in rows if once in r["Name"])
assert steps > 0, "no kernel matching %r" % once


def group(name):
    if "svae::" in name:
        return "library kernels (svae::)"
    if name.startswith("Cijk_") or "gemm" in name.lower() or "rocblas" in name.lower():
        return "torch: rocBLAS / hipBLASLt GEMMs (MLP layers)"
    if "copyBuffer" in name or "memcpy" in name.lower() or "fillBuffer" in name:
        return "copies"
    return "torch: elementwise / reductions (tanh, softplus, add bias, sums, losses)"


groups, top = {}, []
for r in rows:
    us, calls = float(r["TotalDurationNs"]) / 1e3 / steps, int(r["Calls"]) / steps
    g = groups.setdefault(group(r["Name"]), {"us_per_step": 0.0, "launches_per_step": 0.0})
    g["us_per_step"] += us
    g["launches_per_step"] += calls
    top.append({"name": r["Name"][:110], "us_per_step": us, "calls_per_step": calls})
top.sort(key=lambda d: -d["us_per_step"])
sha = os.path.join(ROOT, "gpurun_out", "prof_" + tag, "csrc_sha16.txt")
out = {"steps_profiled": steps, "gpu_us_per_step_total": sum(g["us_per_step"] for g in groups.values()),
       "groups": dict(sorted(groups.items(), key=lambda kv: -kv[1]["us_per_step"])), "top_kernels": top[:12],
       "csrc_sha16": open(sha).read().strip() if os.path.isfile(sha) else None}
json.dump(out, open(os.path.join(ROOT, "profiles", tag, "breakdown.json"), "w"), indent=1)
print("steps %d, %.0f us of GPU time per step:" % (steps, out["gpu_us_per_step_total"]),
      {k: round(v["us_per_step"]) for k, v in out["groups"].items()})
