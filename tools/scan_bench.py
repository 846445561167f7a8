#!/usr/bin/env python3
"""PCIe-inclusive numbers of the operator path (never bench.py's `value`): read_arrow over a lineitem.arrows FILE with a
host consumer (pinned staging -> H2D -> kernels -> D2H -> 2048-row chunks), with a device-resident consumer (no D2H),
and COPY ... TO 'out.arrows' of the scanned table.  usage: python tools/scan_bench.py [--sf 1] [--dir /dev/shm]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=1.0)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--files", type=int, default=8, help="config 3: also scan the table split into this many files")
    ap.add_argument("--zstd", action="store_true", help="also: the reference benchmark's zstd-compressed Arrow IPC *file* "
                                                         "(benchmark/lineitem.py:128-145), written here by pyarrow")
    args = ap.parse_args()
    import duckdb_arrow_amd as da
    buf, info = da.synth_lineitem_stream(scale_factor=args.sf, seed=42)
    path = os.path.join(args.dir, "mi_lineitem_sf%g.arrows" % args.sf)
    buf.tofile(path)
    con = da.Connection(0)
    out = {"rows": info["n_rows"], "file_bytes": int(buf.size), "sf": args.sf}
    try:
        for mode, opts in (("host_consumer", {}), ("device_resident", {"device_resident": True}),
                           ("host_consumer_zero_copy_direct", {"zero_copy_direct": True}),
                           ("device_resident_zero_copy_direct", {"device_resident": True, "zero_copy_direct": True})):
            best = None
            for _ in range(args.repeat):
                t0 = time.perf_counter()
                rel = con.read_arrow(path, **opts)
                n = rel.count()          # native pull loop (a Python loop over 29 k chunks would be the bottleneck)
                dt = time.perf_counter() - t0
                rel.close()
                assert n == info["n_rows"]
                best = dt if best is None else min(best, dt)
            out[mode] = {"seconds": best, "rows_per_s": info["n_rows"] / best, "file_GBps": buf.size / best / 1e9}
        # the pipeline alone -- bodies already in host memory (scan_arrow_ipc over caller buffers, zero-copy
        # on the host side), H2D -> kernels -> D2H overlapped, no file I/O
        for mode, opts in (("buffers_host_consumer", {}), ("buffers_device_resident", {"device_resident": True})):
            best = None
            for _ in range(args.repeat):
                t0 = time.perf_counter()
                n = con.scan_arrow_ipc([buf], **opts).count()
                dt = time.perf_counter() - t0
                assert n == info["n_rows"]
                best = dt if best is None else min(best, dt)
            out[mode] = {"seconds": best, "rows_per_s": info["n_rows"] / best, "GBps_in": buf.size / best / 1e9}
        # filter pushdown: only the selection vector matters to the consumer
        t0 = time.perf_counter()
        rel = con.read_arrow(path).project(["l_shipdate", "l_extendedprice", "l_discount", "l_quantity"]).filter_range("l_shipdate", 8766, 9131)
        sel = rel.count()
        out["q6_columns_filter_pushdown"] = {"seconds": time.perf_counter() - t0, "selected": sel,
                                             "selectivity": sel / info["n_rows"]}
        # fused consumer: Q6 evaluated on the GPU, 32 bytes come back; only the 4 Q6 columns are read from the file
        best = None
        for _ in range(args.repeat):
            t0 = time.perf_counter()
            total, selected, scanned = con.read_arrow(path).sum_product(
                "l_extendedprice", "l_discount", [("l_shipdate", 8766, 9131), ("l_discount", 5, 8), ("l_quantity", -2**63, 2400)])
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out["q6_fused_on_gpu"] = {"seconds": best, "rows_per_s": scanned / best, "revenue_scale4": total, "selected": selected}
        # BASELINE configs[3] end to end: COPY (FROM read_arrow(file)) TO 'out.arrows' (row_group_size 122880): scan
        # (pread, H2D, decode, D2H) -> sink (host staging, H2D, K7 encode, D2H) -> write(); host staging is one thread
        opath = os.path.join(args.dir, "mi_copy_out_sf%g.arrows" % args.sf)
        try:
            res = {}
            for threads in ([int(x) for x in os.environ.get("MI_BENCH_WRITER_THREADS", "1,2,4,6").split(",")]):
                os.environ["MI_WRITER_THREADS"] = str(threads)
                best = None
                for _ in range(2):
                    t0 = time.perf_counter()
                    con.copy_to(con.read_arrow(path), opath, row_group_size=122880)
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                res["sink_threads_%d" % threads] = {"seconds": best, "rows_per_s": info["n_rows"] / best, "out_bytes": os.path.getsize(opath)}
            os.environ.pop("MI_WRITER_THREADS", None)
            t0 = time.perf_counter()
            con.copy_to(con.read_arrow(path), opath, row_group_size=122880)
            dt = time.perf_counter() - t0
            out["copy_scan_to_file"] = dict(res, default={"seconds": dt, "rows_per_s": info["n_rows"] / dt}, seconds=dt, rows_per_s=info["n_rows"] / dt)
        finally:
            if os.path.exists(opath):
                os.unlink(opath)
        # BASELINE configs[2] shape on one GPU: the table as a list of files, l_shipdate filter pushed into the scan,
        # row groups sharded rank / world (each rank of an N-GPU job runs exactly this with its own rank)
        if args.files > 1:
            per = (info["n_batches"] + args.files - 1) // args.files * 122880
            paths = []
            for i in range(args.files):
                first = i * per
                if first >= info["n_rows"]:
                    break
                part, _ = da.synth_lineitem_stream(scale_factor=args.sf, seed=42, n_rows=min(per, info["n_rows"] - first), first_row=first)
                pth = os.path.join(args.dir, "mi_lineitem_sf%g_part%d.arrows" % (args.sf, i))
                part.tofile(pth)
                paths.append(pth)
            try:
                res = {}
                for world in (1, 2):
                    sel, rows, secs = 0, 0, []
                    for rank in range(world):
                        best = None
                        for _ in range(args.repeat):
                            t0 = time.perf_counter()
                            d = con.read_arrow(paths, rank=rank, world=world).filter_range("l_shipdate", 8766, 9131).count(detail=True)
                            dt = time.perf_counter() - t0
                            best = dt if best is None else min(best, dt)
                        secs.append(best)
                        sel += d["selected"]
                        rows += d["rows"]
                    res["world_%d" % world] = {"seconds_per_rank": secs, "rows": rows, "selected": sel}
                assert res["world_1"]["selected"] == res["world_2"]["selected"] and res["world_1"]["rows"] == info["n_rows"]
                # the in-library multi-device scan (here: several contexts on the one GPU of the box) and late materialisation
                for name, kw in (("contexts_1_compact", dict(filter_compact=True)), ("contexts_2", dict(contexts=2)),
                                 ("contexts_2_compact", dict(contexts=2, filter_compact=True)), ("contexts_4_compact", dict(contexts=4, filter_compact=True))):
                    best = None
                    for _ in range(args.repeat):
                        kw2 = dict(kw)
                        nctx = kw2.pop("contexts", 0)
                        if nctx:
                            kw2["contexts"] = [da.Context(0) for _ in range(nctx)]
                        t0 = time.perf_counter()
                        d = con.read_arrow(paths, **kw2).filter_range("l_shipdate", 8766, 9131).count(detail=True)
                        dt = time.perf_counter() - t0
                        best = dt if best is None else min(best, dt)
                    assert d["selected"] == res["world_1"]["selected"] and d["rows"] == info["n_rows"]
                    res[name] = {"seconds": best, "rows_per_s": info["n_rows"] / best}
                out["multi_file_filter_pushdown"] = dict(files=len(paths), **res)
            finally:
                for pth in paths:
                    os.unlink(pth)
        if args.zstd:
            import pyarrow as pa
            import pyarrow.ipc as ipc
            table = ipc.open_stream(path).read_all()
            zpath = os.path.join(args.dir, "mi_lineitem_sf%g_zstd.arrow" % args.sf)
            with ipc.new_file(zpath, table.schema, options=ipc.IpcWriteOptions(compression="zstd")) as w:
                w.write_table(table, max_chunksize=122880)
            del table
            try:
                z = {"file_bytes": os.path.getsize(zpath)}
                for name, fn in (("count_all_columns", lambda: con.read_arrow(zpath).count()),
                                 ("q6_fused_on_gpu", lambda: con.read_arrow(zpath).sum_product(
                                     "l_extendedprice", "l_discount", [("l_shipdate", 8766, 9131), ("l_discount", 5, 8), ("l_quantity", -2**63, 2400)])[2]),
                                 ("pyarrow_read_all_cpu", lambda: ipc.open_file(zpath).read_all().num_rows)):
                    best = None
                    for _ in range(args.repeat):
                        t0 = time.perf_counter()
                        n = fn()
                        dt = time.perf_counter() - t0
                        best = dt if best is None else min(best, dt)
                    assert n == info["n_rows"]
                    z[name] = {"seconds": best, "rows_per_s": info["n_rows"] / best}
                out["zstd_ipc_file"] = z
            finally:
                os.unlink(zpath)
    finally:
        os.unlink(path)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
