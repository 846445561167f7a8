#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE output of tools/fetch_calibration.py -> FETCH_SIZE per launch beside the known byte counts.
usage: python tools/fetch_calibration_summary.py <rocprof output dir> > profiles/r03/fetch_calibration.json"""
import csv
import glob
import json
import os
import sys


def main():
    d = sys.argv[1]
    per = {}
    for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            if row["Counter_Name"] != "FETCH_SIZE":
                continue
            per.setdefault(row["Kernel_Name"][:120], []).append(float(row["Counter_Value"]))
    n = 1 << 30
    out = {"buffer_bytes": n, "note": "FETCH_SIZE in KB per launch; the three reads of tools/fetch_calibration.py are the torch reduce kernels "
                                      "with the largest counts (in launch order: wide, strided8, strided4, three repetitions each)",
           "kernels": {k: {"launches": len(v), "FETCH_SIZE_KB": v[:12], "bytes_if_KB": [x * 1024 for x in v[:12]],
                           "ratio_to_buffer": [x * 1024 / n for x in v[:12]]} for k, v in per.items() if max(v) * 1024 > n / 16}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
