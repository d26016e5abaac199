#!/usr/bin/env python
"""Turn the raw rocprofv3 output of profiles/run_profile.sh (gpurun_out/prof_<tag>/) into the small
summaries committed under profiles/<tag>/: kernel_stats.csv (copy of the --stats table) and
pmc_hbm.json (HBM bytes per launch of the dominant kernel from the FETCH_SIZE / WRITE_SIZE passes,
with the gfx950 correction of MI355X_MICROARCH.md: wide coalesced reads are counted at 1/2).
Usage: python profiles/summarize.py <tag> [kernel-name-substring]"""
import csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else "lds_estep"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    out = {}
    disp = {}
    for counter, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write"), ("SQ_INSTS_VALU", "pmc_valu")):
        # per kernel NAME the mean over its dispatches; a launch of the hot path that is several kernels (the tile E-step
        # beyond one workgroup per CU: forward-half + backward-half instance) counts as their sum
        per = {}
        path = os.path.join(src, d, "bench_counter_collection.csv")
        if not os.path.isfile(path):
            continue
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter or sub not in row.get("Kernel_Name", ""):
                    continue
                disp = per.setdefault(row["Kernel_Name"], {})
                disp[row["Dispatch_Id"]] = disp.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
        total = sum(sum(v.values()) / max(1, len(v)) for v in per.values())
        launches = min((len(v) for v in per.values()), default=0)
        if counter == "SQ_INSTS_VALU":       # wavefront-level VALU instructions issued per launch (summed over the chip)
            out["SQ_INSTS_VALU_per_launch_mean"] = total
            continue
        out[counter + "_KB_per_launch_mean"] = total
        out[counter + "_launches"] = launches
        out["kernel"] = " + ".join(sorted(per))
    out["note"] = ("rocprofv3 --pmc, separate passes (profiles/run_profile.sh); gfx950: FETCH_SIZE counts wide "
                   "coalesced reads at 1/2 (MI355X_MICROARCH.md, HBM) -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KB")
    sha = os.path.join(src, "csrc_sha16.txt")
    out["csrc_sha16"] = open(sha).read().strip() if os.path.isfile(sha) else None     # sources the counters belong to
    out["hbm_bytes_per_launch_corrected"] = 1024.0 * (2 * out["FETCH_SIZE_KB_per_launch_mean"]
                                                       + out["WRITE_SIZE_KB_per_launch_mean"])
    json.dump(out, open(os.path.join(dst, "pmc_hbm.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
