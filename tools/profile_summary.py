#!/usr/bin/env python3
"""Collect rocprofv3 output directories into the small summaries committed under profiles/<round>/.

  python tools/profile_summary.py --stats gpurun_out/prof_stats --fetch gpurun_out/prof_fetch --write gpurun_out/prof_write \
      --bench gpurun_out/bench.json --out profiles/r01_final

kernel_stats.csv  = rocprofv3 --kernel-trace --stats (copied as is)
pmc_traffic.json  = per kernel, HBM bytes per dispatch from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE).  Both
                    counters are in KB; on gfx950 FETCH_SIZE counts half of the bytes of wide coalesced reads
                    (/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so read bytes = 2 * FETCH_SIZE * 1024.
The same file is also written to profiles/pmc_traffic_latest.json, where bench.py picks up `roofline.traffic`.
"""
import argparse
import csv
import glob
import json
import os
import re
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_sha import kernel_source_sha

KERNELS = ("transcode_copy", "transcode_dec128", "transcode_string", "transcode_misc", "encode_fixed", "encode_string")


def short(name):
    m = re.search(r"(transcode_\w+|encode_\w+|filter_\w+|agg_sum_product|gather_\w+|lz4_\w+|zstd_\w+|k8_\w+)", name)
    return m.group(1) if m else None


def counter_avg(directory, counter):
    out = {}
    for path in glob.glob(os.path.join(directory, "**", "*_counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if k is None or row["Counter_Name"] != counter:
                    continue
                out.setdefault(k, []).append(float(row["Counter_Value"]))
    return {k: {"avg_KB": sum(v) / len(v), "dispatches": len(v)} for k, v in sorted(out.items())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--bench", nargs="*", default=[])
    ap.add_argument("--note", default="")
    ap.add_argument("--latest", default="", help="also write the traffic summary here (profiles/pmc_traffic_latest.json)")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if a.stats:
        for path in glob.glob(os.path.join(a.stats, "**", "*_kernel_stats.csv"), recursive=True):
            # our kernels only (torch's fill / copy kernels of the harness are noise here), names shortened
            with open(path, newline="") as f, open(os.path.join(a.out, "kernel_stats.csv"), "w", newline="") as g:
                rd = csv.reader(f)
                wr = csv.writer(g)
                head = next(rd)
                wr.writerow(head)
                for row in rd:
                    k = short(row[0])
                    if k:
                        wr.writerow([k] + row[1:])
    if a.fetch and a.write:
        fetch, write = counter_avg(a.fetch, "FETCH_SIZE"), counter_avg(a.write, "WRITE_SIZE")
        traffic = {k: int(2 * fetch[k]["avg_KB"] * 1024 + write.get(k, {"avg_KB": 0})["avg_KB"] * 1024) for k in fetch}
        doc = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes). " + a.note +
                       " Values are KB per dispatch; on gfx950 FETCH_SIZE counts 1/2 of wide coalesced reads "
                       "(MI355X_MICROARCH.md, HBM) so read bytes = 2*FETCH_SIZE*1024.",
               "kernel_source_sha": kernel_source_sha(),   # bench.py: traffic_stale when the kernels have changed since
               "FETCH_SIZE": fetch, "WRITE_SIZE": write, "traffic_bytes_per_launch": traffic}
        targets = [os.path.join(a.out, "pmc_traffic.json")]
        if a.latest:   # the headline workload: bench.py picks `roofline.traffic` up from here
            targets.append(a.latest)
        for p in targets:
            with open(p, "w") as f:
                json.dump(doc, f, indent=1)
    for b in a.bench:
        if os.path.exists(b):
            shutil.copy(b, os.path.join(a.out, os.path.basename(b)))
    print("wrote", sorted(os.listdir(a.out)))


if __name__ == "__main__":
    main()
