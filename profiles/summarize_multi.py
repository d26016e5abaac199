#!/usr/bin/env python
"""Like summarize.py, for runs with several kernels of interest (tools/prof_train.sh): copies the --stats
table and writes pmc_hbm.json with one entry per kernel whose name contains one of the given substrings.
Usage: python profiles/summarize_multi.py <tag> <substr> [<substr> ...]"""
import csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, subs = sys.argv[1], sys.argv[2:]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    if os.path.isfile(os.path.join(src, "bench.log")):
        shutil.copy(os.path.join(src, "bench.log"), os.path.join(dst, "bench.log"))
    per = {}
    for counter, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        with open(os.path.join(src, d, "bench_counter_collection.csv")) as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name", "")
                if row.get("Counter_Name") != counter or not any(s in name for s in subs):
                    continue
                k = per.setdefault(name, {}).setdefault(counter, {})
                k[row["Dispatch_Id"]] = k.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    out = {"note": "rocprofv3 --pmc, separate passes (tools/prof_train.sh); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KB "
                   "(gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md, HBM)", "kernels": {}}
    for name, cs in sorted(per.items()):
        f = list(cs.get("FETCH_SIZE", {}).values()); w = list(cs.get("WRITE_SIZE", {}).values())
        fm, wm = sum(f) / max(1, len(f)), sum(w) / max(1, len(w))
        out["kernels"][name] = {"FETCH_SIZE_KB_per_launch_mean": fm, "WRITE_SIZE_KB_per_launch_mean": wm,
                                "launches": len(f), "hbm_bytes_per_launch_corrected": 1024.0 * (2 * fm + wm)}
    json.dump(out, open(os.path.join(dst, "pmc_hbm.json"), "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
