#!/usr/bin/env python3
"""Generates tests/golden/ (run once in the build container; outputs are committed).

Inputs are DATA FILES the reference's own tests hold (/root/reference/data/...) and pyarrow -- the oracle the
reference's python tests use (test/python/test_integration.py:32-61).  No reference source text is copied.

Outputs
  ref_data/...                      verbatim copies of the reference's test data files (fixtures)
  lineitem_sf0_01_q6.arrows         data/parquet-testing/lineitem_sf0_01.parquet, the four TPC-H Q6 columns, cast to
                                    the schema DuckDB exports (DECIMAL(15,2) -> decimal128, DATE -> date32), all 60175 rows
  lineitem_sf0_01_head.arrows       first 8192 rows, all 16 columns, DuckDB export schema (pyarrow omits the
                                    validity bitmaps of null-free columns; DuckDB-writer-style files with bitmaps
                                    present are produced by the build's own writer in the tests)
  edge_*.arrows                     pyarrow-written edge cases (nulls, every supported type, window boundaries,
                                    empty batches, dictionary encoding, int64 offsets)
  expected.json                     logical expected values computed with pyarrow: row counts, per-column sha256 of
                                    the canonical value list, and the known answers of the reference's tests
"""
import decimal
import hashlib
import json
import os
import shutil
import sys

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pyarrow.ipc as ipc
import pyarrow.parquet as pq

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/data"


def canon_value(v):
    """Canonical JSON-able logical value. Temporal values are reduced to the integer DuckDB stores."""
    if v is None or isinstance(v, (bool, int, str)):
        return v
    if isinstance(v, list):
        return [canon_value(x) for x in v]
    if isinstance(v, tuple):
        return [canon_value(x) for x in v]  # map entries (key, value)
    if isinstance(v, dict):
        return {k: canon_value(x) for k, x in v.items()}
    if isinstance(v, float):
        return "nan" if v != v else repr(v)
    if isinstance(v, bytes):
        return "b:" + v.hex()
    if isinstance(v, decimal.Decimal):
        return "d:" + str(v)
    raise TypeError(type(v))


def tdiv(x, d):
    """C++ integer division (truncation toward zero), which is what the scan's `/ 1000` does."""
    q = abs(x) // d
    return q if x >= 0 else -q


def canon_column(col: pa.ChunkedArray):
    """pyarrow column -> list of canonical logical values as DuckDB would hold them after the scan."""
    t = col.type
    if pa.types.is_dictionary(t):
        col = col.cast(t.value_type)
        t = col.type
    if pa.types.is_timestamp(t):
        ints = col.cast(pa.int64()).to_pylist()
        if t.tz is None:
            return ints  # TIMESTAMP_S/MS/US/NS keep the arrow unit (direct conversion)
        f = {"s": 1000000, "ms": 1000, "us": 1, "ns": None}[t.unit]
        return [None if x is None else (x * f if f else tdiv(x, 1000)) for x in ints]
    if pa.types.is_date32(t):
        return col.cast(pa.int32()).to_pylist()
    if pa.types.is_date64(t):
        return [None if x is None else tdiv(x, 86400000) for x in col.cast(pa.int64()).to_pylist()]
    if pa.types.is_time32(t):
        f = {"s": 1000000, "ms": 1000}[t.unit]
        return [None if x is None else x * f for x in col.cast(pa.int32()).to_pylist()]
    if pa.types.is_time64(t):
        ints = col.cast(pa.int64()).to_pylist()
        return ints if t.unit == "us" else [None if x is None else tdiv(x, 1000) for x in ints]
    if pa.types.is_duration(t):
        f = {"s": 1000000, "ms": 1000, "us": 1, "ns": None}[t.unit]
        return [None if x is None else (x * f if f else tdiv(x, 1000)) for x in col.cast(pa.int64()).to_pylist()]
    if pa.types.is_interval(t):
        return [None if x is None else [x.months, x.days, tdiv(x.nanoseconds, 1000)] for x in col.to_pylist()]
    if pa.types.is_decimal(t):
        scale = t.scale
        return [None if x is None else int(x.scaleb(scale)) for x in col.to_pylist()]
    return [canon_value(v) for v in col.to_pylist()]


def column_digest(values):
    return hashlib.sha256(json.dumps(values, separators=(",", ":")).encode()).hexdigest()


def table_expectation(t: pa.Table):
    return {"rows": t.num_rows, "columns": {name: column_digest(canon_column(t.column(name))) for name in t.column_names},
            "null_counts": {name: t.column(name).null_count for name in t.column_names}}


def write_stream(path, schema, batches, options=None):
    with ipc.new_stream(path, schema, options=options) as w:
        for b in batches:
            w.write_batch(b)


def read_any(path):
    try:
        return ipc.open_stream(path).read_all()
    except pa.ArrowInvalid:
        return ipc.open_file(path).read_all()


def duckdb_lineitem_schema(names):
    types = {}
    for n in ["l_orderkey", "l_partkey", "l_suppkey", "l_linenumber"]:
        types[n] = pa.int64()
    for n in ["l_quantity", "l_extendedprice", "l_discount", "l_tax"]:
        types[n] = pa.decimal128(15, 2)
    for n in ["l_returnflag", "l_linestatus", "l_shipinstruct", "l_shipmode", "l_comment"]:
        types[n] = pa.string()
    for n in ["l_shipdate", "l_commitdate", "l_receiptdate"]:
        types[n] = pa.date32()
    return pa.schema([pa.field(n, types[n]) for n in names])


def main():
    rng = np.random.default_rng(20250523)
    exp = {}

    # ---- 1. the reference's own data files -----------------------------------------------------------
    files = ["test.arrows", "fruit.arrow"]
    for root, _, fs in os.walk(os.path.join(REF, "multifile")):
        for f in fs:
            files.append(os.path.relpath(os.path.join(root, f), REF))
    for rel in sorted(files):
        dst = os.path.join(HERE, "ref_data", rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(REF, rel), dst)
        exp["ref_data/" + rel] = table_expectation(read_any(dst))

    # known answers of the reference's tests on data/test.arrows (test/sql/read_arrow.test:35-55)
    t = read_any(os.path.join(REF, "test.arrows"))
    msg = t.filter(pc.equal(t["commit"], "fa5f0299f046c46e1b2f671e5e3b4f1956522711"))["message"].to_pylist()
    micros = np.array(t["time"].cast(pa.int64()).to_pylist(), dtype=np.int64)
    days = np.floor_divide(micros, 86400000000)
    wednesday = int(np.sum((days + 3) % 7 == 2))  # 1970-01-01 was a Thursday; Monday = 0
    exp["kat"] = {
        "test_arrows_rows": t.num_rows,
        "test_arrows_commit_message": msg,
        "test_arrows_wednesday": wednesday,
        "test_arrows_batch0_buffers": [[0, 0], [0, 4100], [4104, 40960], [45064, 0], [45064, 8192], [53256, 0],
                                       [53256, 4096], [57352, 0], [57352, 128], [57480, 0], [57480, 4100],
                                       [61584, 76938]],
    }
    assert exp["kat"]["test_arrows_rows"] == 15487 and wednesday == 2927 and msg == ["ARROW-1: Initial Arrow Code Commit"]

    # ---- 2. lineitem SF0.01 in DuckDB's export schema ------------------------------------------------
    li = pq.read_table(os.path.join(REF, "parquet-testing", "lineitem_sf0_01.parquet"))
    q6_cols = ["l_quantity", "l_extendedprice", "l_discount", "l_shipdate"]
    q6 = li.select(q6_cols).cast(duckdb_lineitem_schema(q6_cols))
    p = os.path.join(HERE, "lineitem_sf0_01_q6.arrows")
    write_stream(p, q6.schema, q6.to_batches(max_chunksize=20480))
    exp["lineitem_sf0_01_q6.arrows"] = table_expectation(read_any(p))
    # TPC-H Q6 on unscaled integers: sum(extendedprice * discount), scale 4 (test/nodejs/arrow_test.js:423-424)
    qty = np.array(canon_column(q6["l_quantity"]), dtype=np.int64)
    price = np.array(canon_column(q6["l_extendedprice"]), dtype=np.int64)
    disc = np.array(canon_column(q6["l_discount"]), dtype=np.int64)
    ship = np.array(canon_column(q6["l_shipdate"]), dtype=np.int64)
    sel = (ship >= 8766) & (ship < 9131) & (disc >= 5) & (disc <= 7) & (qty < 2400)
    exp["kat"]["q6_sf0_01_revenue_scale4"] = int(np.sum(price[sel] * disc[sel]))
    exp["kat"]["q6_sf0_01_rows_passing"] = int(sel.sum())
    exp["kat"]["shipdate_1994_selected"] = int(((ship >= 8766) & (ship < 9131)).sum())
    assert exp["kat"]["q6_sf0_01_revenue_scale4"] == 11930532253, exp["kat"]

    names = li.column_names
    head = li.slice(0, 8192).cast(duckdb_lineitem_schema(names))
    p = os.path.join(HERE, "lineitem_sf0_01_head.arrows")
    write_stream(p, head.schema, head.to_batches(max_chunksize=4096))
    exp["lineitem_sf0_01_head.arrows"] = table_expectation(read_any(p))

    # ---- 3. edge cases --------------------------------------------------------------------------------
    def nullify(arr, frac=0.2):
        mask = rng.random(len(arr)) < frac
        return pa.array(arr.to_pylist(), type=arr.type, mask=mask)

    # 3a. the reference's own python test shape: ints/strings/bools with NULLs x 5 batches
    #     (test/python/test_arrow_ipc_scan.py:7-17)
    b = pa.record_batch([pa.array([1, 2, 3, 4]), pa.array(["foo", "bar", "baz", None]),
                         pa.array([True, None, False, True])], names=["f0", "f1", "f2"])
    p = os.path.join(HERE, "edge_reftest.arrows")
    write_stream(p, b.schema, [b] * 5)
    exp["edge_reftest.arrows"] = table_expectation(read_any(p))

    # 3b. every supported type, with nulls, batch sizes straddling the 64-bit word and 2048-row windows
    def types_batch(n):
        words = ["", "a", "hello", "twelve bytes", "thirteen byte", "a considerably longer string value éè",
                 "x" * 90]
        cols = {
            "i8": pa.array(rng.integers(-128, 127, n), pa.int8()), "u8": pa.array(rng.integers(0, 255, n), pa.uint8()),
            "i16": pa.array(rng.integers(-2**15, 2**15 - 1, n), pa.int16()),
            "u16": pa.array(rng.integers(0, 2**16 - 1, n), pa.uint16()),
            "i32": pa.array(rng.integers(-2**31, 2**31 - 1, n), pa.int32()),
            "u32": pa.array(rng.integers(0, 2**32 - 1, n), pa.uint32()),
            "i64": pa.array(rng.integers(-2**63, 2**63 - 1, n), pa.int64()),
            "u64": pa.array(rng.integers(0, 2**63 - 1, n).astype(np.uint64) * 2, pa.uint64()),
            "f32": pa.array(rng.standard_normal(n).astype(np.float32)), "f64": pa.array(rng.standard_normal(n)),
            "b": pa.array(rng.random(n) < 0.5),
            "dec4": pa.array([decimal.Decimal(int(x)).scaleb(-1) for x in rng.integers(-9999, 9999, n)], pa.decimal128(4, 1)),
            "dec9": pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in rng.integers(-10**9 + 1, 10**9 - 1, n)], pa.decimal128(9, 2)),
            "dec18": pa.array([decimal.Decimal(int(x)).scaleb(-3) for x in rng.integers(-10**18 + 1, 10**18 - 1, n)], pa.decimal128(18, 3)),
            "dec38": pa.array([decimal.Decimal(int(x) * 10**15 + 7).scaleb(-5) for x in rng.integers(-10**18, 10**18, n)], pa.decimal128(38, 5)),
            "d32": pa.array(rng.integers(-10000, 20000, n).astype(np.int32), pa.date32()),
            "d64": pa.array(rng.integers(-10000, 20000, n) * 86400000, pa.date64()),
            "t32s": pa.array(rng.integers(0, 86399, n).astype(np.int32), pa.time32("s")),
            "t32ms": pa.array(rng.integers(0, 86399999, n).astype(np.int32), pa.time32("ms")),
            "t64us": pa.array(rng.integers(0, 86399999999, n), pa.time64("us")),
            "t64ns": pa.array(rng.integers(0, 86399999999999, n), pa.time64("ns")),
            "ts_s": pa.array(rng.integers(-10**9, 2 * 10**9, n), pa.timestamp("s")),
            "ts_ms": pa.array(rng.integers(-10**12, 2 * 10**12, n), pa.timestamp("ms")),
            "ts_us": pa.array(rng.integers(-10**15, 2 * 10**15, n), pa.timestamp("us")),
            "ts_ns": pa.array(rng.integers(-10**18, 2 * 10**18, n), pa.timestamp("ns")),
            "tz_s": pa.array(rng.integers(-10**9, 2 * 10**9, n), pa.timestamp("s", tz="UTC")),
            "tz_ms": pa.array(rng.integers(-10**12, 2 * 10**12, n), pa.timestamp("ms", tz="UTC")),
            "tz_us": pa.array(rng.integers(-10**15, 2 * 10**15, n), pa.timestamp("us", tz="Europe/Amsterdam")),
            "tz_ns": pa.array(rng.integers(-10**18, 2 * 10**18, n), pa.timestamp("ns", tz="UTC")),
            "s": pa.array([words[i] for i in rng.integers(0, len(words), n)], pa.string()),
            "ls": pa.array([words[i] for i in rng.integers(0, len(words), n)], pa.large_string()),
            "bin": pa.array([words[i].encode() for i in rng.integers(0, len(words), n)], pa.binary()),
            "fsb": pa.array([bytes(rng.integers(0, 255, 20).astype(np.uint8)) for _ in range(n)], pa.binary(20)),
        }
        return pa.record_batch([nullify(a) for a in cols.values()], names=list(cols.keys()))

    sizes = [1, 63, 64, 65, 2047, 2048, 2049, 4097]
    batches = [types_batch(n) for n in sizes]
    p = os.path.join(HERE, "edge_types.arrows")
    write_stream(p, batches[0].schema, batches)
    exp["edge_types.arrows"] = table_expectation(read_any(p))
    exp["edge_types.arrows"]["batch_sizes"] = sizes

    # 3b'. types DuckDB >= 1.3 also reads: decimal32/64 inputs, half floats, month_day_nano intervals, durations, null
    def types2_batch(n):
        import datetime
        cols = {
            "d32_7": pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in rng.integers(-9999999, 9999999, n)], pa.decimal32(7, 2)),
            "d32_4": pa.array([decimal.Decimal(int(x)).scaleb(-1) for x in rng.integers(-9999, 9999, n)], pa.decimal32(4, 1)),
            "d64_12": pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in rng.integers(-10**12 + 1, 10**12 - 1, n)], pa.decimal64(12, 2)),
            "d64_9": pa.array([decimal.Decimal(int(x)).scaleb(-3) for x in rng.integers(-10**9 + 1, 10**9 - 1, n)], pa.decimal64(9, 3)),
            "d64_4": pa.array([decimal.Decimal(int(x)).scaleb(-1) for x in rng.integers(-9999, 9999, n)], pa.decimal64(4, 1)),
            "f16": pa.array(rng.integers(0, 65536, n).astype(np.uint16).view(np.float16)),
            "mdn": pa.array([pa.MonthDayNano([int(a), int(b), int(c)]) for a, b, c in
                             zip(rng.integers(-100, 100, n), rng.integers(-1000, 1000, n), rng.integers(-10**15, 10**15, n))],
                            pa.month_day_nano_interval()),
            "dur_s": pa.array(rng.integers(-10**9, 10**9, n), pa.duration("s")),
            "dur_ms": pa.array(rng.integers(-10**12, 10**12, n), pa.duration("ms")),
            "dur_us": pa.array(rng.integers(-10**15, 10**15, n), pa.duration("us")),
            "dur_ns": pa.array(rng.integers(-10**18, 10**18, n), pa.duration("ns")),
        }
        arrays = [nullify(a) for a in cols.values()] + [pa.nulls(n)]
        return pa.record_batch(arrays, names=list(cols.keys()) + ["nothing"])

    b2 = [types2_batch(n) for n in (1, 65, 2049, 3000)]
    p = os.path.join(HERE, "edge_types2.arrows")
    write_stream(p, b2[0].schema, b2)
    exp["edge_types2.arrows"] = table_expectation(read_any(p))

    # 3b''. nested types and string views: lists, lists of lists, structs, lists of structs,
    #       fixed-size lists, maps, large lists, utf8_view / binary_view -- with NULLs at every level
    def nested_batch(n):
        def maybe(v, p=0.15):
            return None if rng.random() < p else v
        words = ["", "a", "hello", "twelve bytes", "thirteen byte", "a considerably longer string value"]
        ints = lambda k: [maybe(int(x)) for x in rng.integers(-1000, 1000, k)]
        cols = {
            "l_i": pa.array([maybe(ints(int(rng.integers(0, 6)))) for _ in range(n)], pa.list_(pa.int32())),
            "l_s": pa.array([maybe([maybe(words[int(i)]) for i in rng.integers(0, len(words), int(rng.integers(0, 4)))]) for _ in range(n)],
                            pa.list_(pa.string())),
            "ll": pa.array([maybe([maybe(ints(int(rng.integers(0, 4)))) for _ in range(int(rng.integers(0, 4)))]) for _ in range(n)],
                           pa.list_(pa.list_(pa.int64()))),
            "st": pa.array([maybe({"a": maybe(int(rng.integers(0, 100))), "b": maybe(words[int(rng.integers(0, len(words)))])}) for _ in range(n)],
                           pa.struct([("a", pa.int32()), ("b", pa.string())])),
            "l_st": pa.array([maybe([maybe({"x": maybe(int(rng.integers(0, 10**9))), "y": maybe(words[int(rng.integers(0, len(words)))])})
                                     for _ in range(int(rng.integers(0, 4)))]) for _ in range(n)],
                             pa.list_(pa.struct([("x", pa.int64()), ("y", pa.string())]))),
            "fl": pa.array([maybe([maybe(int(x)) for x in rng.integers(-100, 100, 3)]) for _ in range(n)], pa.list_(pa.int16(), 3)),
            "mp": pa.array([maybe([(words[int(i)] + str(j), maybe(int(rng.integers(0, 50)))) for j, i in
                                   enumerate(rng.integers(0, len(words), int(rng.integers(0, 4))))]) for _ in range(n)],
                           pa.map_(pa.string(), pa.int32())),
            "lgl": pa.array([maybe([float(x) for x in rng.integers(0, 100, int(rng.integers(0, 5)))]) for _ in range(n)],
                            pa.large_list(pa.float64())),
            "sv": pa.array([maybe(words[int(i)]) for i in rng.integers(0, len(words), n)], pa.string_view()),
            "bv": pa.array([maybe(words[int(i)].encode() * 2) for i in rng.integers(0, len(words), n)], pa.binary_view()),
        }
        return pa.record_batch(list(cols.values()), names=list(cols.keys()))

    nb = [nested_batch(n) for n in (5, 2100, 4500)]
    p = os.path.join(HERE, "edge_nested.arrows")
    write_stream(p, nb[0].schema, nb)
    exp["edge_nested.arrows"] = table_expectation(read_any(p))

    # 3c. empty batches, all-null and all-valid columns, zero-length validity
    sch = pa.schema([("a", pa.int32()), ("s", pa.string()), ("n", pa.int64())])
    eb = [pa.record_batch([pa.array([], pa.int32()), pa.array([], pa.string()), pa.array([], pa.int64())], schema=sch),
          pa.record_batch([pa.array([1, 2, 3], pa.int32()), pa.array(["", "", "z"]), pa.array([None, None, None], pa.int64())], schema=sch),
          pa.record_batch([pa.array([], pa.int32()), pa.array([], pa.string()), pa.array([], pa.int64())], schema=sch),
          pa.record_batch([pa.array([None] * 70, pa.int32()), pa.array([None] * 70, pa.string()), pa.array(list(range(70)), pa.int64())], schema=sch)]
    p = os.path.join(HERE, "edge_empty.arrows")
    write_stream(p, sch, eb)
    exp["edge_empty.arrows"] = table_expectation(read_any(p))

    # 3d. dictionary encoded (beyond the reference: base_stream_reader.cpp:86-96 accepts RecordBatch only)
    cats = ["AIR", "MAIL", "SHIP", "TRUCK", "RAIL", "REG AIR", "FOB", "a string longer than twelve"]
    def dict_batch(n):
        idx = pa.array(rng.integers(0, len(cats), n).astype(np.int32), mask=rng.random(n) < 0.1)
        idx8 = pa.array(rng.integers(0, 4, n).astype(np.int8), mask=rng.random(n) < 0.1)
        return pa.record_batch([pa.DictionaryArray.from_arrays(idx, pa.array(cats)),
                                pa.DictionaryArray.from_arrays(idx8, pa.array([10, 20, 30, 40], pa.int64())),
                                pa.array(rng.integers(0, 100, n), pa.int32())], names=["mode", "code", "x"])
    db = [dict_batch(n) for n in (100, 3000)]
    p = os.path.join(HERE, "edge_dict.arrows")
    write_stream(p, db[0].schema, db)
    exp["edge_dict.arrows"] = table_expectation(read_any(p))

    # 3e. an IPC *file* (footer + magic), readable as an embedded stream (ipc_file_stream_reader.cpp:107-119)
    p = os.path.join(HERE, "edge_file_format.arrow")
    with ipc.new_file(p, batches[1].schema) as w:
        w.write_batch(batches[1])
        w.write_batch(batches[4])
    exp["edge_file_format.arrow"] = table_expectation(read_any(p))

    with open(os.path.join(HERE, "expected.json"), "w") as f:
        json.dump(exp, f, indent=1, sort_keys=True)
    total = sum(os.path.getsize(os.path.join(r, x)) for r, _, fs in os.walk(HERE) for x in fs)
    print("golden written: %d files, %.1f MB" % (sum(len(fs) for _, _, fs in os.walk(HERE)), total / 1e6))


if __name__ == "__main__":
    sys.exit(main())
