#!/usr/bin/env python3
"""K8 end to end: the lineitem table as an LZ4_FRAME-compressed IPC stream (what pyarrow / Feather V2 write), scanned by a
device-resident consumer with the bodies decompressed (a) by the reader's host threads, (b) in HBM by the K8 kernels,
beside the uncompressed file.  usage: python tools/lz4_bench.py [--sf 10] [--dir /dev/shm]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=10.0)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--legs", default="", help="comma-separated leg names to run (default: all)")
    ap.add_argument("--codec", default="lz4", choices=["lz4", "zstd"], help="zstd: what the reference's benchmark writes (benchmark/lineitem.py:135); leg names keep their lz4_ prefix")
    args = ap.parse_args()
    import pyarrow as pa
    import pyarrow.ipc as ipc
    import duckdb_arrow_amd as da
    buf, info = da.synth_lineitem_stream(scale_factor=args.sf, seed=42)
    plain = os.path.join(args.dir, "mi_lz4_plain_sf%g.arrows" % args.sf)
    packed = os.path.join(args.dir, "mi_lz4_packed_sf%g.arrows" % args.sf)
    out = {"rows": info["n_rows"], "plain_bytes": int(buf.size), "codec": args.codec, "pipeline_depth": args.depth,
           "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "unset: the library asks for 24 at load time"), "MI_IO_THREADS": os.environ.get("MI_IO_THREADS", "unset (8)")}
    try:
        buf.tofile(plain)
        t0 = time.perf_counter()
        reader = ipc.open_stream(pa.py_buffer(buf))
        with ipc.new_stream(packed, reader.schema, options=ipc.IpcWriteOptions(compression=args.codec)) as w:
            for b in reader:
                w.write_batch(b)
        out["pyarrow_compress_seconds"] = time.perf_counter() - t0
        out["lz4_bytes"] = os.path.getsize(packed)
        del buf
        con = da.Connection(0)
        hbm = {}   # K8 is the default of a device-resident scan (ZSTD: when the process has hardware queues for 16 record batches)
        legs = [("plain", plain, {"device_resident": True}), ("lz4_host_threads", packed, {"host_decompress": True, "device_resident": True}),
                ("lz4_in_hbm", packed, dict(hbm, device_resident=True)),
                # the default consumer: vectors (and, for K8, the decompressed string payloads) travel back to pinned host memory
                ("host_consumer_plain", plain, {}), ("host_consumer_lz4_host_threads", packed, {"host_decompress": True}),
                ("host_consumer_lz4_in_hbm", packed, {"host_decompress": "gpu"})]
        only = set(x for x in args.legs.split(",") if x)
        for tag, path, kw in legs:
            if only and tag not in only:
                continue
            best, st = None, None
            for _ in range(2):
                rel = con.read_arrow(path, pipeline_depth=args.depth, **kw)
                t0 = time.perf_counter()
                got = rel.count(detail=True)
                dt = time.perf_counter() - t0
                st = rel.stats()
                rel.close()
                assert got["rows"] == info["n_rows"]
                best = dt if best is None else min(best, dt)
            out[tag] = {"seconds": best, "rows_per_s": info["n_rows"] / best, "stats": st}
            print(tag, "%.3f s" % best, file=sys.stderr, flush=True)
        # TPC-H Q6 fused on the GPU (mi_scan_sum_product): 4 of 16 columns are read, decompressed and decoded; 32 bytes come back
        for tag, path, kw in (("q6_plain", plain, {}), ("q6_lz4_host_threads", packed, {"host_decompress": True}), ("q6_lz4_in_hbm", packed, hbm)):
            if only and tag not in only:
                continue
            best, res = None, None
            for _ in range(2):
                rel = con.read_arrow(path, device_resident=True, pipeline_depth=args.depth, **kw)
                t0 = time.perf_counter()
                res = rel.sum_product("l_extendedprice", "l_discount", [("l_shipdate", 8766, 9131), ("l_discount", 5, 8), ("l_quantity", 0, 2400)])
                dt = time.perf_counter() - t0
                st = rel.stats()
                rel.close()
                best = dt if best is None else min(best, dt)
            out[tag] = {"seconds": best, "rows_per_s": info["n_rows"] / best, "sum": str(res[0]), "rows_selected": res[1], "h2d_bytes": st["h2d_bytes"]}
            print(tag, "%.3f s" % best, file=sys.stderr, flush=True)
        if not only:
            assert out["q6_plain"]["sum"] == out["q6_lz4_in_hbm"]["sum"] == out["q6_lz4_host_threads"]["sum"]
    finally:
        for p in (plain, packed):
            if os.path.exists(p):
                os.unlink(p)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
