#!/usr/bin/env python
"""Reads the two rocprofv3 passes of tools/ubench/fetch_calibration.hip (FETCH_SIZE, WRITE_SIZE) and prints, per access
pattern, counter bytes / known bytes.  Usage: python tools/ubench/fetch_calibration.py <dir with fetch/ and write/> [out.json]"""
import csv, glob, json, os, re, sys

N = 1 << 27
EXPECT = {"cal_read16": N * 8, "cal_read8": N * 8, "cal_write16": N * 8, "cal_write8": N * 8,
          "cal_rows<16, false>": 4096 * 32768 * 8, "cal_rows<10, false>": 4096 * 32760 * 8,
          "cal_rows<16, true>": 4096 * 32768 * 8, "cal_rows<10, true>": 4096 * 32760 * 8}


def per_kernel(root, sub, counter):
    out = {}
    for path in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            d = out.setdefault(row["Kernel_Name"], {})
            d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in out.items()}


def main():
    root = sys.argv[1]
    res = {}
    fetch, write = per_kernel(root, "fetch", "FETCH_SIZE"), per_kernel(root, "write", "WRITE_SIZE")
    for name, kb in sorted(fetch.items()):
        key = next((k for k in EXPECT if k.replace(" ", "") in name.replace(" ", "")), None)
        if key is None:
            continue
        res[key] = {"known_bytes": EXPECT[key], "FETCH_SIZE_KB": kb, "WRITE_SIZE_KB": write.get(name),
                    "fetch_counter_over_known": kb * 1024.0 / EXPECT[key],
                    "write_counter_over_known": (write.get(name) or 0.0) * 1024.0 / EXPECT[key]}
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
