#!/usr/bin/env python
"""Summaries of a tools/prof_generic.sh run (gpurun_out/prof_<tag>/) for profiles/<tag>/: kernel_stats.csv (the --stats
table), bench.log (the command's own output) and pmc.json -- per kernel whose name contains one of the given substrings
(default: every kernel of the svae:: namespace): mean per launch of FETCH_SIZE / WRITE_SIZE (KiB), the HBM bytes
(2 FETCH + WRITE) KiB (gfx950 correction, calibrated in profiles/r3_fetch_calibration), SQ_INSTS_VALU, and where the
wavefront cycles went (SQ_WAIT_ANY = parked at s_waitcnt / barrier, SQ_WAIT_INST_ANY = stalled at issue,
SQ_ACTIVE_INST_ANY = issuing; as fractions of SQ_WAVE_CYCLES).
Usage: python profiles/summarize_all.py <tag> [substr ...]"""
import csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, subs):
    out = {}
    if not os.path.isfile(path):
        return out
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if not any(s in name for s in subs):
                continue
            d = out.setdefault(name, {}).setdefault(row["Counter_Name"], {})
            d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    return out


def main():
    tag, subs = sys.argv[1], (sys.argv[2:] or ["svae::"])
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    if os.path.isfile(os.path.join(src, "bench.log")):
        shutil.copy(os.path.join(src, "bench.log"), os.path.join(dst, "bench.log"))
    res = {}
    for d in ("pmc_fetch", "pmc_write", "pmc_valu", "pmc_sq"):
        for name, cs in per_kernel(os.path.join(src, d, "bench_counter_collection.csv"), subs).items():
            r = res.setdefault(name, {})
            for counter, disp in cs.items():
                r[counter + "_per_launch_mean"] = sum(disp.values()) / max(1, len(disp))
                r["launches"] = len(disp)
    for name, r in res.items():
        if "FETCH_SIZE_per_launch_mean" in r and "WRITE_SIZE_per_launch_mean" in r:
            r["hbm_bytes_per_launch_corrected"] = 1024.0 * (2 * r["FETCH_SIZE_per_launch_mean"] + r["WRITE_SIZE_per_launch_mean"])
        w = r.get("SQ_WAVE_CYCLES_per_launch_mean")
        if w:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c + "_per_launch_mean" in r:
                    r[c + "_frac_of_wave_cycles"] = r[c + "_per_launch_mean"] / w
    sha = os.path.join(src, "csrc_sha16.txt")
    out = {"csrc_sha16": open(sha).read().strip() if os.path.isfile(sha) else None,
           "note": "rocprofv3 --pmc, one counter family per pass (tools/prof_generic.sh); FETCH / WRITE in KiB; hbm_bytes = "
                   "(2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE factor 2, profiles/r3_fetch_calibration)",
           "kernels": dict(sorted(res.items()))}
    json.dump(out, open(os.path.join(dst, "pmc.json"), "w"), indent=1)
    for name, r in sorted(res.items()):
        print("%-90s launches %4d  HBM %8.1f MB  VALU %.3g  wait %.2f issue-stall %.2f active %.2f" % (
            name[:90], r.get("launches", 0), r.get("hbm_bytes_per_launch_corrected", 0) / 1e6,
            r.get("SQ_INSTS_VALU_per_launch_mean", 0), r.get("SQ_WAIT_ANY_frac_of_wave_cycles", 0),
            r.get("SQ_WAIT_INST_ANY_frac_of_wave_cycles", 0), r.get("SQ_ACTIVE_INST_ANY_frac_of_wave_cycles", 0)))


if __name__ == "__main__":
    main()
