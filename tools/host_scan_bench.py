#!/usr/bin/env python3
"""The host-consumer operator path alone (bench.py's operator_path.full_scan_host_consumer leg): SF10 lineitem as 8 files in
/dev/shm, read_arrow -> count through pread -> pinned -> H2D -> kernels -> D2H, with the bytes that crossed PCIe per row
(mi_scan_get_stats).  usage: [MI_IO_THREADS=n] python tools/host_scan_bench.py [--sf 10] [--depth 3]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=10.0)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--files", type=int, default=8)
    ap.add_argument("--depth", default="3,6")
    ap.add_argument("--no-bind", action="store_true", help="leave this process's own threads (the file writer, the consumer) where the scheduler puts them")
    args = ap.parse_args()
    import duckdb_arrow_amd as da
    con = da.Connection(0)
    # the host program on the GPU's NUMA node: the page cache of the files it writes, the thread that takes the chunks (the
    # library's own threads go there by themselves unless MI_NUMA_BIND=0)
    node = -1 if args.no_bind else con.ctx.bind_this_thread()
    buf, info = da.synth_lineitem_stream(scale_factor=args.sf, seed=42)
    d = os.path.join(args.dir, "mi_hostscan_%d" % os.getpid())
    os.makedirs(d, exist_ok=True)
    paths = [os.path.join(d, "lineitem_%d.arrows" % i) for i in range(args.files)]
    offs, nb = info["batch_offsets"], info["n_batches"]
    per = (nb + args.files - 1) // args.files
    out = {"rows": info["n_rows"], "MI_IO_THREADS": os.environ.get("MI_IO_THREADS", "default"), "gpu_numa_node": con.ctx.numa()[0],
           "process_bound_to_node": node, "MI_NUMA_BIND": os.environ.get("MI_NUMA_BIND", "default (on)")}
    try:
        for i, p in enumerate(paths):
            lo, hi = offs[min(nb, i * per)], offs[min(nb, (i + 1) * per)]
            with open(p, "wb") as f:
                f.write(buf[: offs[0]].tobytes())
                f.write(memoryview(buf[lo:hi]))
                f.write(b"\xff\xff\xff\xff\x00\x00\x00\x00")
        del buf
        for depth in [int(x) for x in args.depth.split(",")]:
            for tag, kw in (("materialise_all", {"zero_copy_direct": False}), ("default_alias_plain_columns", {})):
                best, st = None, None
                for _ in range(3):
                    rel = con.read_arrow(paths, pipeline_depth=depth, **kw)
                    t0 = time.perf_counter()
                    got = rel.count(detail=True)
                    dt = time.perf_counter() - t0
                    st = rel.stats()
                    rel.close()
                    assert got["rows"] == info["n_rows"]
                    best = dt if best is None else min(best, dt)
                out["%s_depth%d" % (tag, depth)] = {"seconds": best, "rows_per_s": info["n_rows"] / best,
                                                    "h2d_bytes_per_row": st["h2d_bytes"] / info["n_rows"], "d2h_bytes_per_row": st["d2h_bytes"] / info["n_rows"],
                                                    "aliased_bytes_per_row": st["aliased_bytes"] / info["n_rows"]}
        con.close()
    finally:
        for p in paths:
            if os.path.exists(p):
                os.remove(p)
        os.rmdir(d)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
