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


class _DeviceBytes:
    """A device allocation of the library as something torch.as_tensor can view (no copy): the CUDA array interface."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def native_heap_leg(torch, da, _ffi, ctx, hs, buf, info, rounds=5):
    """encode_string_1p on DuckDB-NATIVE string heaps: the long strings of a vector back to back in the heap, strings of <= 12
    bytes inline only (they leave no gap in the heap).  The decoded vectors of `hs` point into the Arrow data buffer, where a
    wave's long strings already lie as they will lie in the output (one coalesced copy per wave); here every (record batch,
    string column) gets a heap of its own with only the long strings in it and rewritten pointers, so a wave's long strings
    are contiguous in the heap but NOT at a constant distance from their places in the output: the kernel's per-row path.
    Same output as the source stream's buffers (checked on the first record batch)."""
    d_in = torch.as_tensor(_DeviceBytes(hs.in_ptr, hs.host.size + 64), device="cuda")
    d_out = torch.as_tensor(_DeviceBytes(hs.out_ptr, hs.out_bytes), device="cuda")
    cols = [(lay, e) for lay in hs.layout for e in lay["columns"] if e["kind"] == _ffi.K_STR32]
    heap_total = sum(e["buffers"][2][1] + 64 for _, e in cols)
    heaps = torch.zeros(heap_total + 256, dtype=torch.uint8, device="cuda")
    rows_total = sum(lay["nrows"] for lay, _ in cols)
    vectors = torch.empty(rows_total * 16 + 256, dtype=torch.uint8, device="cuda")   # the rewritten string_t rows
    total = 0
    spans, tasks = [], []
    hb = vb = 0
    long_bytes = 0
    for lay, e in cols:
        n = lay["nrows"]
        s = vectors[vb: vb + 16 * n]
        s.copy_(d_out[e["data_off"]: e["data_off"] + 16 * n])
        s32, s64 = s.view(torch.int32).view(n, 4), s.view(torch.int64).view(n, 2)
        lens = s32[:, 0].to(torch.int64)
        is_long = lens > 12
        L = torch.where(is_long, lens, torch.zeros_like(lens))
        new_off = torch.cumsum(L, 0) - L
        nbytes = int(L.sum().item())
        if nbytes:
            rows_long = torch.nonzero(is_long).flatten()
            row_of_byte = torch.repeat_interleave(rows_long, L[rows_long])
            src = s64[:, 1][row_of_byte] + (torch.arange(nbytes, device="cuda") - new_off[row_of_byte])
            heaps[hb: hb + nbytes] = d_in[src]
            s64[:, 1][rows_long] = hb + new_off[rows_long]     # pointer = position in `heaps` (ptr_base 0)
        long_bytes += nbytes
        sz = [(n + 7) // 8, e["buffers"][1][1], e["buffers"][2][1]]
        offs = []
        for x in sz:
            offs.append(total)
            total += (x + 63) // 64 * 64 + 64
        spans.append((lay, e, offs, sz))
        tasks.append((n, vectors.data_ptr() + vb, offs, sz, e))
        hb += (nbytes + 63) // 64 * 64
        vb += 16 * n
    arena = torch.zeros(total + 256, dtype=torch.uint8, device="cuda")
    ab = arena.data_ptr()
    out_base = hs.out_ptr
    ctasks = [da.make_task(_ffi.K_ENC_STR32, n, vptr, ab + offs[1], validity=out_base + e["valid_off"], out_validity=ab + offs[0],
                           out_aux=ab + offs[2], buf2=heaps.data_ptr(), ptr_base=0, buf2_len=sz[2], param=0) for n, vptr, offs, sz, e in tasks]
    plan = da.Plan(ctx, ctasks)
    stream = torch.cuda.current_stream().cuda_stream
    plan.launch(stream)
    assert plan.status() == 0
    times = np.array([plan.launch_timed(stream) for _ in range(rounds)])
    med = np.median(times, axis=0)
    out = {"string_columns": len(set(e["name"] for _, e in cols)), "long_string_bytes_in_native_heaps": long_bytes, "kernels": []}
    for i, c in enumerate(plan.class_stats()):
        if c["tiles"]:
            b = c["bytes_read"] + c["bytes_written"]
            out["kernels"].append({"kernel": c["kernel"], "ms": float(med[i]), "algorithmic_bytes": b, "GBps": b / (med[i] * 1e-3) / 1e9})
    ok = True
    for lay, e, offs, sz in spans[: sum(1 for c in hs.layout[0]["columns"] if c["kind"] == _ffi.K_STR32)]:
        body = lay["body_off"]
        for b in (1, 2):
            got = arena[offs[b]: offs[b] + sz[b]].cpu().numpy()
            want = buf[body + e["buffers"][b][0]: body + e["buffers"][b][0] + sz[b]]
            ok = ok and bool(np.array_equal(got, want))
    out["payload_matches_source"] = ok
    plan.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=1.0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--per-column", action="store_true", help="also time one plan per column (diagnostic)")
    ap.add_argument("--native-heap", action="store_true", help="also: encode_string_1p over DuckDB-native string heaps (long strings back to back)")
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
    out = encode_leg(torch, da, _ffi, ctx, hs, buf, info, args.rounds, args.per_column)
    if args.native_heap:
        out["native_heap"] = native_heap_leg(torch, da, _ffi, ctx, hs, buf, info, args.rounds)
    print(json.dumps(out))



if __name__ == "__main__":
    main()
