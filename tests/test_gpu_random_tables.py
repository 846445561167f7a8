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
