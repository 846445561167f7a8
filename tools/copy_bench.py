#!/usr/bin/env python3
"""COPY (FROM read_arrow(file)) TO 'out.arrows' end to end (BASELINE config 4 through the operator path), per sink-thread
count and output strategy, with the writer's stage timers.  usage: python tools/copy_bench.py [--sf 10] [--dir /dev/shm]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=10.0)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--threads", default="1,2,4,6")
    args = ap.parse_args()
    os.environ["MI_WRITER_TIMING"] = "1"
    import duckdb_arrow_amd as da
    buf, info = da.synth_lineitem_stream(scale_factor=args.sf, seed=42)
    path = os.path.join(args.dir, "mi_copy_in_sf%g.arrows" % args.sf)
    opath = os.path.join(args.dir, "mi_copy_out_sf%g.arrows" % args.sf)
    buf.tofile(path)
    con = da.Connection(0)
    out = {"rows": info["n_rows"], "file_bytes": int(buf.size)}
    try:
        t0 = time.perf_counter()
        n = con.read_arrow(path).count()
        out["scan_only_seconds"] = time.perf_counter() - t0
        # what the box gives a plain writer: 21 MB write() calls into a fresh tmpfs file, one thread
        piece = bytes(21 << 20)
        fd = os.open(opath, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        t0 = time.perf_counter()
        for _ in range(200):
            os.write(fd, piece)
        dt = time.perf_counter() - t0
        os.close(fd)
        os.unlink(opath)
        out["raw_tmpfs_write_one_thread"] = {"GBps": 200 * len(piece) / dt / 1e9, "seconds_for_this_table": buf.size / (200 * len(piece) / dt)}
        legs = [(opath, "fused_file", 0), ("/dev/null", "fused_null_sink", 0)]
        for target, tag in ((opath, "file"), ("/dev/null", "null_sink")):
            legs += [(target, tag, t) for t in [int(x) for x in args.threads.split(",")]]
        for target, tag, threads in legs:
            for _once in (0,):
                # threads 0: the fused pump (record batches encoded where the scan decoded them); else the sink-thread pump
                if threads:
                    os.environ["MI_WRITER_THREADS"] = str(threads)
                    os.environ["MI_WRITER_NO_FUSED"] = "1"
                else:
                    os.environ.pop("MI_WRITER_NO_FUSED", None)
                best = None
                for _ in range(2):
                    if target == opath and os.path.exists(opath):
                        os.unlink(opath)
                    t0 = time.perf_counter()
                    con.copy_to(con.read_arrow(path), target, row_group_size=122880)
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                out[("%s_threads_%d" % (tag, threads)) if threads else tag] = {"seconds": best, "rows_per_s": info["n_rows"] / best, "GBps_out": buf.size / best / 1e9}
                print("%s threads %d: %.3f s" % (tag, threads, best), file=sys.stderr, flush=True)
    finally:
        for p in (path, opath):
            if os.path.exists(p):
                os.unlink(p)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
