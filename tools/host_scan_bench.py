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
    args = ap.parse_args()
    import duckdb_arrow_amd as da
    buf, info = da.synth_lineitem_stream(scale_factor=args.sf, seed=42)
    d = os.path.join(args.dir, "mi_hostscan_%d" % os.getpid())
    os.makedirs(d, exist_ok=True)
    paths = [os.path.join(d, "lineitem_%d.arrows" % i) for i in range(args.files)]
    offs, nb = info["batch_offsets"], info["n_batches"]
    per = (nb + args.files - 1) // args.files
    out = {"rows": info["n_rows"], "MI_IO_THREADS": os.environ.get("MI_IO_THREADS", "default")}
    try:
        for i, p in enumerate(paths):
            lo, hi = offs[min(nb, i * per)], offs[min(nb, (i + 1) * per)]
            with open(p, "wb") as f:
                f.write(buf[: offs[0]].tobytes())
                f.write(memoryview(buf[lo:hi]))
                f.write(b"\xff\xff\xff\xff\x00\x00\x00\x00")
        del buf
        con = da.Connection(0)
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
