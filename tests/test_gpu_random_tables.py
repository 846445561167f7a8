"""Seeded differential test: random Arrow tables (random schemas up to three levels of nesting, NULLs at every level, empty
batches, zero-length strings, all-NULL columns) are written with pyarrow, scanned on the GPU through the operator path and
compared with pyarrow's own view of the same file; then copied back out with COPY and compared again."""
import decimal

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pytest

import duckdb_arrow_amd as da

pytestmark = pytest.mark.gpu

WORDS = ["", "a", "xy", "hello", "twelve bytes", "thirteen byte", "a considerably longer string value", "éè utf8 中文"]


def random_type(rng, depth):
    leaves = [pa.int8(), pa.int16(), pa.int32(), pa.int64(), pa.uint8(), pa.uint16(), pa.uint32(), pa.uint64(), pa.float32(),
              pa.float64(), pa.bool_(), pa.string(), pa.large_string(), pa.binary(), pa.string_view(), pa.date32(),
              pa.decimal128(9, 2), pa.decimal128(18, 4), pa.decimal128(30, 6)]
    if depth >= 3 or rng.random() < 0.55:
        return leaves[int(rng.integers(0, len(leaves)))]
    kind = int(rng.integers(0, 5))
    if kind == 0:
        return pa.list_(random_type(rng, depth + 1))
    if kind == 1:
        return pa.large_list(random_type(rng, depth + 1))
    if kind == 2:
        return pa.struct([("f%d" % i, random_type(rng, depth + 1)) for i in range(int(rng.integers(1, 4)))])
    if kind == 3:
        return pa.list_(random_type(rng, depth + 1), int(rng.integers(1, 4)))
    return pa.map_(pa.string(), random_type(rng, depth + 1))


def random_value(rng, t, null_p):
    if rng.random() < null_p:
        return None
    if pa.types.is_boolean(t):
        return bool(rng.integers(0, 2))
    if pa.types.is_integer(t):
        info = np.iinfo(t.to_pandas_dtype())
        return int(rng.integers(info.min, info.max, dtype=t.to_pandas_dtype(), endpoint=True))
    if pa.types.is_floating(t):
        return float(np.float32(rng.normal()) if t == pa.float32() else rng.normal())
    if pa.types.is_date32(t):
        return int(rng.integers(-20000, 40000))
    if pa.types.is_decimal(t):
        return decimal.Decimal(int(rng.integers(-10 ** min(t.precision, 18) + 1, 10 ** min(t.precision, 18)))).scaleb(-t.scale)
    if pa.types.is_binary(t):
        return WORDS[int(rng.integers(0, len(WORDS)))].encode() * int(rng.integers(0, 3))
    if pa.types.is_string(t) or pa.types.is_large_string(t) or pa.types.is_string_view(t):
        return WORDS[int(rng.integers(0, len(WORDS)))]
    if pa.types.is_map(t):
        return [("k%d" % j, random_value(rng, t.item_type, null_p)) for j in range(int(rng.integers(0, 4)))]
    if pa.types.is_fixed_size_list(t):
        return [random_value(rng, t.value_type, null_p) for _ in range(t.list_size)]
    if pa.types.is_list(t) or pa.types.is_large_list(t):
        return [random_value(rng, t.value_type, null_p) for _ in range(int(rng.integers(0, 5)))]
    if pa.types.is_struct(t):
        return {t.field(i).name: random_value(rng, t.field(i).type, null_p) for i in range(t.num_fields)}
    raise TypeError(t)


def canon(t, v):
    """pyarrow python value -> what the mirror returns (stored integers for DATE / DECIMAL, tuples for map entries)."""
    if v is None:
        return None
    if pa.types.is_date32(t):
        return (v - __import__("datetime").date(1970, 1, 1)).days if not isinstance(v, int) else v
    if pa.types.is_decimal(t):
        return int(v.scaleb(t.scale).to_integral_value())
    if pa.types.is_floating(t):
        return "nan" if v != v else float(v)
    if pa.types.is_map(t):
        return [(k, canon(t.item_type, x)) for k, x in v]
    if pa.types.is_fixed_size_list(t) or pa.types.is_list(t) or pa.types.is_large_list(t):
        return [canon(t.value_type, x) for x in v]
    if pa.types.is_struct(t):
        return {t.field(i).name: canon(t.field(i).type, v[t.field(i).name]) for i in range(t.num_fields)}
    return v


def fix_floats(v):
    if isinstance(v, float):
        return "nan" if v != v else v
    if isinstance(v, list):
        return [fix_floats(x) for x in v]
    if isinstance(v, tuple):
        return tuple(fix_floats(x) for x in v)
    if isinstance(v, dict):
        return {k: fix_floats(x) for k, x in v.items()}
    return v


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MI_RANDOM_TABLE_SEEDS", "24"))))
def test_random_tables_scan_and_copy(seed, tmp_path):
    rng = np.random.default_rng(1000 + seed)
    ncols = int(rng.integers(1, 7))
    fields = [pa.field("c%d" % i, random_type(rng, 0)) for i in range(ncols)]
    if seed % 4 == 0:
        fields.append(pa.field("allnull", pa.int32()))
    schema = pa.schema(fields)
    batches = []
    for bi, n in enumerate([int(x) for x in rng.choice([0, 1, 63, 700, 2048, 2500, 5000], size=int(rng.integers(1, 5)))]):
        cols = []
        for f in schema:
            null_p = 1.0 if f.name == "allnull" else (0.0 if rng.random() < 0.2 else 0.15)
            cols.append(pa.array([random_value(rng, f.type, null_p) for _ in range(n)], f.type))
        batches.append(pa.record_batch(cols, schema=schema))
    path = str(tmp_path / "r.arrows")
    with ipc.new_stream(path, schema) as w:
        for b in batches:
            w.write_batch(b)
    want_table = ipc.open_stream(path).read_all()
    con = da.Connection(0)
    rel = con.read_arrow(path)
    got = rel.fetch_columns()
    assert len(got) == len(schema)
    for f, g in zip(schema, got):
        want = [canon(f.type, v) for v in want_table.column(f.name).to_pylist()]
        assert fix_floats(g) == want, (seed, f.name, str(f.type))
    # COPY the scan back out (nested encode, validity always emitted) and let pyarrow compare the two files logically
    out = str(tmp_path / "copy.arrows")
    writable = [f.name for f in schema if "view" not in str(f.type)]   # views are scanned as VARCHAR / BLOB: types change
    if writable:
        con.copy_to(con.read_arrow(path).project(writable), out, row_group_size=2048)
        back = ipc.open_stream(out).read_all()
        assert back.num_rows == want_table.num_rows
        for name in writable:
            a, b = want_table.column(name).to_pylist(), back.column(name).to_pylist()
            assert fix_floats(a) == fix_floats(b), (seed, name)
    con.close()


# ---------------------------------------------------------------------------------------- flat types x container modes
def _flat_random_column(rng, n):
    """One random flat column of a temporal / decimal / dictionary / plain type (NULLs at 15 %)."""
    mask = rng.random(n) < 0.15
    kind = int(rng.integers(0, 14))
    i64 = lambda lo, hi: rng.integers(lo, hi, n)
    if kind == 0:
        return pa.array(i64(-10**12, 10**12), pa.timestamp(["s", "ms", "us", "ns"][int(rng.integers(0, 4))]), mask=mask)
    if kind == 1:
        return pa.array(i64(-10**11, 10**11), pa.timestamp(["s", "ms", "us", "ns"][int(rng.integers(0, 4))], "UTC"), mask=mask)
    if kind == 2:
        return pa.array(i64(0, 86400).astype(np.int32), pa.time32("s"), mask=mask)
    if kind == 3:
        return pa.array(i64(0, 86400000).astype(np.int32), pa.time32("ms"), mask=mask)
    if kind == 4:
        return pa.array(i64(0, 86400 * 10**6), pa.time64("us"), mask=mask)
    if kind == 5:
        return pa.array(i64(0, 86400 * 10**9), pa.time64("ns"), mask=mask)
    if kind == 6:
        return pa.array(i64(-10**6, 10**6) * 86400000, pa.date64(), mask=mask)
    if kind == 7:
        return pa.array(i64(-10**9, 10**9), pa.duration(["s", "ms", "us", "ns"][int(rng.integers(0, 4))]), mask=mask)
    if kind == 8:
        p = int(rng.integers(1, 39))
        s = int(rng.integers(0, p + 1))
        vals = [None if m else decimal.Decimal(int(rng.integers(-10**min(p, 18) + 1, 10**min(p, 18)))).scaleb(-s) for m in mask]
        return pa.array(vals, pa.decimal128(p, s))
    if kind == 9:
        cats = pa.array([WORDS[i] + str(i) for i in range(len(WORDS))])
        idx_t = [pa.int8(), pa.int16(), pa.int32(), pa.int64(), pa.uint8(), pa.uint16()][int(rng.integers(0, 6))]
        return pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, len(cats), n), idx_t, mask=mask), cats)
    if kind == 10:
        return pa.array(rng.normal(size=n).astype(np.float16), pa.float16(), mask=mask)
    if kind == 11:
        return pa.array([None if m else bytes(rng.integers(0, 256, 5, dtype=np.uint8)) for m in mask], pa.binary(5))
    if kind == 12:
        return pa.array([None] * n, pa.null())
    return pa.array([None if m else WORDS[int(rng.integers(0, len(WORDS)))] * int(rng.integers(0, 4)) for m in mask], pa.large_binary()
                    if rng.random() < 0.5 else pa.large_string())


@pytest.mark.parametrize("mode", ["stream", "file", "zstd", "lz4", "zero_copy", "projection"])
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MI_RANDOM_FLAT_SEEDS", "4"))))
def test_random_flat_types_in_every_container(seed, mode, tmp_path):
    """Temporal unit casts, decimals of every precision, dictionaries with every index type, half floats, fixed binary,
    null columns -- through the stream format, the IPC *file* format, ZSTD and LZ4 bodies, zero_copy_direct and a projected
    read; expected values are make_golden.py's canonical forms (what DuckDB stores), computed from pyarrow."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden import canon_column
    from helpers import canon_python
    rng = np.random.default_rng(5000 + seed)
    ncols = int(rng.integers(2, 8))
    sizes = [int(x) for x in rng.choice([0, 1, 64, 999, 2048, 4100], size=int(rng.integers(1, 4)))]
    first = [_flat_random_column(rng, sizes[0]) for _ in range(ncols)]
    schema = pa.schema([pa.field("c%d" % i, a.type) for i, a in enumerate(first)])
    batches = [pa.record_batch(first, schema=schema)]
    for n in sizes[1:]:
        cols = []
        for f in schema:
            while True:   # a column of the same type (the generator picks the type at random)
                a = _flat_random_column(rng, n)
                if a.type == f.type:
                    break
            cols.append(a)
        batches.append(pa.record_batch(cols, schema=schema))
    path = str(tmp_path / ("t." + ("arrow" if mode == "file" else "arrows")))
    opts = ipc.IpcWriteOptions(compression=mode) if mode in ("zstd", "lz4") else None
    if mode == "file":
        with ipc.new_file(path, schema) as w:
            for b in batches:
                w.write_batch(b)
        table = ipc.open_file(path).read_all()
    else:
        with ipc.new_stream(path, schema, options=opts) as w:
            for b in batches:
                w.write_batch(b)
        table = ipc.open_stream(path).read_all()
    con = da.Connection(0)
    names = table.column_names
    if mode == "projection":
        names = [n for i, n in enumerate(names) if (seed + i) % 2 == 0] or names[:1]
        names = names[::-1]
    rel = con.read_arrow(path, accept_dictionaries=True, zero_copy_direct=(mode == "zero_copy"))
    if mode == "projection":
        rel = rel.project(names)
    got = rel.fetch_columns()
    for name, g in zip(names, got):
        t = table.schema.field(name).type
        want = canon_column(table.column(name))
        if pa.types.is_duration(t):
            g = [None if v is None else v[2] for v in g]
        elif pa.types.is_float16(t) or pa.types.is_floating(t):
            g = [None if v is None else ("nan" if v != v else repr(v)) for v in g]
            want = [None if v is None else (v if isinstance(v, str) else repr(float(v))) for v in want]
        assert canon_python(g) == want, (seed, mode, name, str(t))
    con.close()


# ---------------------------------------------------------------------------------------- kernel level vs the oracle
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MI_RANDOM_ORACLE_SEEDS", "10"))))
def test_random_tables_bit_exact_vs_oracle(seed):
    """The same random nested tables at kernel level: every decoded array (data bytes, validity words, list entries,
    child windows) equals the CPU oracle's bit for bit -- layout parity, not just logical values."""
    from duckdb_arrow_amd.hbm import HbmStream
    from oracle import pyoracle as po
    from test_gpu_decode_parity import assert_streams_equal
    rng = np.random.default_rng(9000 + seed)
    schema = pa.schema([pa.field("c%d" % i, random_type(rng, 0)) for i in range(int(rng.integers(1, 6)))])
    sink = pa.BufferOutputStream()
    with ipc.new_stream(sink, schema) as w:
        for n in [int(x) for x in rng.choice([0, 1, 63, 700, 2048, 2500, 5000], size=int(rng.integers(1, 4)))]:
            w.write_batch(pa.record_batch([pa.array([random_value(rng, f.type, 0.15) for _ in range(n)], f.type) for f in schema],
                                          schema=schema))
    buf = np.frombuffer(sink.getvalue(), np.uint8)
    ctx = da.Context(0)
    hs = HbmStream(ctx, buf)
    hs.launch()
    assert hs.status() == 0
    _, want = po.decode_stream(buf)
    assert_streams_equal(hs.fetch(), want)
