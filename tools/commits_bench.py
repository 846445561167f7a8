#!/usr/bin/env python3
"""BASELINE configs[4]: dictionary-encoded + nullable VARCHAR-heavy record batches (arrow-commits shape, SURVEY 8d
"Config 5") resident in HBM.  Not bench.py's `value`: a secondary workload that stresses the offset / bitmap / dictionary
arms of the decode kernels.  The stream is generated here with numpy + pyarrow (seeded), decoded through the C ABI, timed
per kernel class with HIP events, and a sample of record batches is compared bit for bit with the CPU oracle.

  python tools/commits_bench.py [--rows 10000000] [--steps 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def commits_stream(n_rows, rows_per_batch=122880, seed=7, null_frac=0.10):
    import pyarrow as pa
    import pyarrow.ipc as ipc
    rng = np.random.default_rng(seed)
    authors = pa.array(["author %04d <a%04d@example.org>" % (i, i) for i in range(2000)])
    components = pa.array(["C++", "Python", "Rust", "Java", "Go", "R", "JS", "C#", "Ruby", "MATLAB", "Docs", "CI", "Format",
                           "Flight", "Parquet", "Gandiva", "Dataset", "Compute", "Packaging", "Release"])
    hexd = np.frombuffer(b"0123456789abcdef", np.uint8)
    sink = pa.BufferOutputStream()
    writer = None
    for first in range(0, n_rows, rows_per_batch):
        n = min(rows_per_batch, n_rows - first)

        def validity():
            bits = rng.random(n) >= null_frac
            return pa.py_buffer(np.packbits(bits, bitorder="little").tobytes()), int(n - bits.sum())

        # commit: 40 hex characters
        commit_data = hexd[rng.integers(0, 16, n * 40, dtype=np.uint8)]
        commit = pa.Array.from_buffers(pa.string(), n, [None, pa.py_buffer((np.arange(n + 1, dtype=np.int32) * 40).tobytes()),
                                                        pa.py_buffer(commit_data.tobytes())])
        # message: 9..513 bytes, mean ~68 (log-normal), 10 % NULL
        lens = np.clip(np.exp(rng.normal(3.95, 0.72, n)), 9, 513).astype(np.int32)
        offs = np.zeros(n + 1, np.int32)
        np.cumsum(lens, out=offs[1:])
        msg_data = rng.integers(97, 123, int(offs[-1]), dtype=np.uint8)
        vb, nc = validity()
        message = pa.Array.from_buffers(pa.string(), n, [vb, pa.py_buffer(offs.tobytes()), pa.py_buffer(msg_data.tobytes())], null_count=nc)
        vb, nc = validity()
        t = pa.Array.from_buffers(pa.timestamp("us", "UTC"), n,
                                  [vb, pa.py_buffer((1_450_000_000_000_000 + rng.integers(0, 3 * 10**14, n)).astype(np.int64).tobytes())],
                                  null_count=nc)
        vb, nc = validity()
        files = pa.Array.from_buffers(pa.int32(), n, [vb, pa.py_buffer(rng.integers(1, 200, n).astype(np.int32).tobytes())], null_count=nc)
        vb, nc = validity()
        merge = pa.Array.from_buffers(pa.bool_(), n, [vb, pa.py_buffer(np.packbits(rng.random(n) < 0.2, bitorder="little").tobytes())],
                                      null_count=nc)
        a_idx = pa.array(rng.integers(0, len(authors), n).astype(np.int32), mask=rng.random(n) < null_frac)
        c_idx = pa.array(rng.integers(0, len(components), n).astype(np.int32), mask=rng.random(n) < null_frac)
        batch = pa.record_batch([commit, t, files, merge, message, pa.DictionaryArray.from_arrays(a_idx, authors),
                                 pa.DictionaryArray.from_arrays(c_idx, components)],
                                names=["commit", "time", "files", "merge", "message", "author", "component"])
        if writer is None:
            writer = ipc.new_stream(sink, batch.schema)
        writer.write_batch(batch)
    writer.close()
    return np.frombuffer(sink.getvalue(), np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    import torch
    import duckdb_arrow_amd as da
    from duckdb_arrow_amd.hbm import HbmStream
    from oracle import pyoracle as po

    t0 = time.time()
    buf = commits_stream(args.rows)
    t_gen = time.time() - t0
    ctx = da.Context(0)
    hs = HbmStream(ctx, buf, accept_dictionaries=True)
    stream = torch.cuda.current_stream().cuda_stream
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 1.0:
        hs.launch(stream)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        hs.launch(stream)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        hs.launch(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    assert hs.status() == 0
    per_class = np.zeros(7)
    for _ in range(5):
        per_class += np.array(hs.plan.launch_timed(stream))
    per_class /= 5
    st = hs.stats()
    kernels = []
    for cs, ms in zip(hs.plan.class_stats(), per_class):
        if cs["tiles"]:
            b = cs["bytes_read"] + cs["bytes_written"]
            kernels.append({"kernel": cs["kernel"], "ms": float(ms), "algorithmic_bytes": b, "achieved_GBps": b / ms / 1e6,
                            "frac_of_8TBps": b / ms / 1e6 / 8000.0})
    # parity: sampled record batches vs the oracle (dictionary batch travels with the sample)
    rb = [m for m in po.walk_stream(buf) if m["type"] == po.MSG_RECORD_BATCH][:2]
    end = rb[-1]["body_off"] + rb[-1]["body_len"]
    _, want = po.decode_stream(buf[:end])          # a prefix keeps every stream position (= string pointer) unchanged
    got = hs.fetch(batches=list(range(len(want))))
    ok = True
    for gb, wb in zip(got, want):
        for gc, wc in zip(gb["columns"], wb["columns"]):
            ok = ok and np.array_equal(gc["data"], wc["data"]) and np.array_equal(gc["validity"], wc["validity"])
            if "dictionary" in wc:
                ok = ok and np.array_equal(gc["dictionary"]["data"], wc["dictionary"]["data"])
    parity = {"checked_batches": list(range(len(want))), "bit_exact": bool(ok)}
    alg = st["bytes_read"] + st["bytes_written"]
    print(json.dumps({"workload": "arrow-commits shape: commit(40 B) / time ts[us,UTC] / files int32 / merge bool / message "
                                  "(9-513 B, mean ~68) with 10 % NULLs + author, component dictionary-encoded (int32 indices)",
                      "rows": args.rows, "stream_bytes": int(buf.size), "ms_per_step": dt / args.steps * 1e3,
                      "rows_per_s": args.rows * args.steps / dt, "algorithmic_bytes_per_row": alg / args.rows,
                      "achieved_GBps_whole_step": alg * args.steps / dt / 1e9, "kernels": kernels, "parity": parity,
                      "generate_seconds": t_gen}))


if __name__ == "__main__":
    main()
