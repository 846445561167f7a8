#!/usr/bin/env python3
"""BASELINE config 4 at kernel level: DuckDB vectors resident in HBM (the output of the decode) -> Arrow buffers with the
K7 encode kernels, one plan for the whole table.  Prints per-kernel-class HIP-event times and algorithmic GB/s.
usage: python tools/encode_bench.py [--sf 1] [--rounds 5]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def encode_leg(torch, da, _ffi, ctx, hs, buf, info, rounds=5, per_column=False):
    """K7 over the decoded vectors of `hs` (an HbmStream that has been launched): one plan for the whole table, HIP-event
    time per kernel class, algorithmic GB/s, and a check of the first record batch's offsets and data buffers against the
    source stream.  Used by this tool and by bench.py's `config4_encode_kernels` block."""
    args = argparse.Namespace(rounds=rounds, per_column=per_column)
    in_base, out_base = hs.in_ptr, hs.out_ptr
    enc_kind = {_ffi.K_COPY: _ffi.K_ENC_COPY, _ffi.K_DEC128: _ffi.K_ENC_DEC128, _ffi.K_STR32: _ffi.K_ENC_STR32}
    total = 0
    spans = []
    for lay in hs.layout:
        n = lay["nrows"]
        for e in lay["columns"]:
            nb = 3 if e["kind"] == _ffi.K_STR32 else 2
            sz = [(n + 7) // 8, e["buffers"][1][1], e["buffers"][2][1] if nb == 3 else 0]
            offs = []
            for s in sz:
                offs.append(total)
                total += (s + 63) // 64 * 64 + 64
            spans.append((lay, e, offs, sz))
    arena = torch.zeros(total + 256, dtype=torch.uint8, device="cuda")
    ab = arena.data_ptr()
    tasks = []
    names = []
    for lay, e, offs, sz in spans:
        names.append(e["name"])
        n = lay["nrows"]
        is_str = e["kind"] == _ffi.K_STR32
        tasks.append(da.make_task(enc_kind[e["kind"]], n, out_base + e["data_off"], ab + offs[1], validity=out_base + e["valid_off"],
                                  out_validity=ab + offs[0], out_aux=(ab + offs[2]) if is_str else 0, buf2=in_base, ptr_base=0,
                                  buf2_len=sz[2] if is_str else 0, param=0 if is_str else e["param"]))
    plan = da.Plan(ctx, tasks)
    stream = torch.cuda.current_stream().cuda_stream
    plan.launch(stream)
    assert plan.status() == 0
    times = np.array([plan.launch_timed(stream) for _ in range(args.rounds)])
    med = np.median(times, axis=0)
    cs = plan.class_stats()
    out = {"rows": info["n_rows"], "ms_total": float(np.median(times.sum(axis=1))), "kernels": []}
    for i, c in enumerate(cs):
        if c["tiles"]:
            b = c["bytes_read"] + c["bytes_written"]
            out["kernels"].append({"kernel": c["kernel"], "ms": float(med[i]), "algorithmic_bytes": b, "GBps": b / (med[i] * 1e-3) / 1e9})
    out["rows_per_s"] = info["n_rows"] / (out["ms_total"] * 1e-3)
    # check: first record batch, the offsets and data buffers of every column equal the source stream's buffers
    ok = True
    for lay, e, offs, sz in spans[: len(hs.layout[0]["columns"])]:
        body = lay["body_off"]
        for b in (1, 2):
            if sz[b]:
                got = arena[offs[b]: offs[b] + sz[b]].cpu().numpy()
                want = buf[body + e["buffers"][b][0]: body + e["buffers"][b][0] + sz[b]]
                ok = ok and bool(np.array_equal(got, want))
    out["payload_matches_source"] = ok
    if args.per_column:
        out["per_column"] = {}
        for nm in dict.fromkeys(names):
            p1 = da.Plan(ctx, [t for t, x in zip(tasks, names) if x == nm])
            p1.launch(stream)
            ts = np.array([p1.launch_timed(stream) for _ in range(args.rounds)])
            st = p1.stats()
            ms = float(np.median(ts.sum(axis=1)))
            out["per_column"][nm] = {"ms": ms, "GBps": (st["bytes_read"] + st["bytes_written"]) / ms / 1e6}
            p1.close()
    plan.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=1.0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--per-column", action="store_true", help="also time one plan per column (diagnostic)")
    args = ap.parse_args()
    import torch
    import duckdb_arrow_amd as da
    from duckdb_arrow_amd import _ffi
    from duckdb_arrow_amd.hbm import HbmStream
    torch.cuda.set_device(0)
    buf, info = da.synth_lineitem_stream(scale_factor=args.sf, seed=42)
    ctx = da.Context(0)
    hs = HbmStream(ctx, buf)
    hs.launch()
    assert hs.status() == 0
    torch.cuda.synchronize()
    print(json.dumps(encode_leg(torch, da, _ffi, ctx, hs, buf, info, args.rounds, args.per_column)))



if __name__ == "__main__":
    main()
