#!/usr/bin/env python3
"""In-process A/B of kernel variants (cdna_hip_programming.md rule 24: interleaved rounds in ONE process).
usage: python tools/ab_bench.py [--sf 10] [--rounds 7] "copy=1,string=1" "copy=2,string=1" ...
Prints the median / min HIP-event time of every kernel class per configuration."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=10.0)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--no-validity", action="store_true")
    ap.add_argument("configs", nargs="+")
    args = ap.parse_args()
    import torch
    import duckdb_arrow_amd as da
    from duckdb_arrow_amd import _ffi
    from duckdb_arrow_amd.hbm import HbmStream
    torch.cuda.set_device(0)
    buf, info = da.synth_lineitem_stream(scale_factor=args.sf, seed=42, with_validity=not args.no_validity)
    hs = HbmStream(da.Context(0), buf)
    stream = torch.cuda.current_stream().cuda_stream
    cs = hs.plan.class_stats()
    defaults = dict(copy=2, dec128=2, string=2, grid=0, tile_table=1)

    def apply(cfg):
        knobs = dict(defaults)
        for kv in cfg.split(","):
            if kv:
                k, v = kv.split("=")
                knobs[k] = int(v)
        for k, v in knobs.items():
            _ffi.check(_ffi.lib().mi_tune(k.encode(), v))

    times = {c: [] for c in args.configs}
    for c in args.configs:  # warm every variant
        apply(c)
        hs.launch(stream)
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for c in args.configs:
            apply(c)
            times[c].append(hs.plan.launch_timed(stream))
    assert hs.status() == 0
    total_bytes = sum(x["bytes_read"] + x["bytes_written"] for x in cs)
    for c in args.configs:
        t = np.array(times[c])
        med, mn = np.median(t, axis=0), t.min(axis=0)
        tot = np.median(t.sum(axis=1))
        line = "%-40s total %.3f ms (%.1f%% of 8 TB/s)" % (c, tot, 100 * total_bytes / (tot * 1e-3) / 8e12)
        for i, x in enumerate(cs):
            if x["tiles"]:
                b = x["bytes_read"] + x["bytes_written"]
                line += " | %s med %.3f min %.3f (%.0f GB/s)" % (x["kernel"].replace("transcode_", ""), med[i], mn[i], b / (med[i] * 1e-3) / 1e9)
        print(line)


if __name__ == "__main__":
    main()
