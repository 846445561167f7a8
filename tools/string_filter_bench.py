#!/usr/bin/env python3
"""K6 with string leaves through the scan operator: SELECT count(*) FROM read_arrow(lineitem SF) WHERE l_shipmode IN ('MAIL','SHIP')
AND l_shipinstruct = 'DELIVER IN PERSON' AND l_quantity < 2400 (TPC-H Q12 / Q19 shapes), device-resident consumer, only the
filter columns are read.  Run under `rocprofv3 --kernel-trace --stats` for the filter_program time per record batch."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=10.0)
    ap.add_argument("--dir", default="/dev/shm")
    args = ap.parse_args()
    import duckdb_arrow_amd as da
    buf, info = da.synth_lineitem_stream(scale_factor=args.sf, seed=42)
    path = os.path.join(args.dir, "mi_strflt_sf%g.arrows" % args.sf)
    buf.tofile(path)
    del buf
    con = da.Connection(0)
    out = {"rows": info["n_rows"]}
    try:
        for tag, expr in (("shipmode_in_2", ("l_shipmode", "in", ["MAIL", "SHIP"])),
                          ("shipinstruct_eq_17_bytes", ("l_shipinstruct", "=", "DELIVER IN PERSON")),
                          ("q12_like", ("and", ("l_shipmode", "in", ["MAIL", "SHIP"]), ("l_shipinstruct", "=", "DELIVER IN PERSON"), ("l_quantity", "<", 2400))),
                          ("int_only_shipdate_range", ("and", ("l_shipdate", ">=", 8766), ("l_shipdate", "<", 9131)))):
            best = None
            for _ in range(2):
                first = expr[1][0] if expr[0] == "and" else expr[0]
                rel = con.read_arrow(path, device_resident=True, pipeline_depth=8).project([first]).filter(expr)   # count(*): no other column is read
                t0 = time.perf_counter()
                got = rel.count(detail=True)
                dt = time.perf_counter() - t0
                rel.close()
                best = dt if best is None else min(best, dt)
            out[tag] = {"seconds": best, "rows_per_s": info["n_rows"] / best, "selected": got["selected"]}
            print(tag, "%.3f s" % best, got["selected"], file=sys.stderr, flush=True)
    finally:
        os.unlink(path)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
