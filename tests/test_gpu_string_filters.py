"""K6 on VARCHAR / BLOB columns: =, <>, <, <=, >, >=, IN and starts_with (LIKE 'abc%') against byte strings, alone and inside AND / OR trees with integer leaves, with
selection vectors and with late materialisation.  The reference pushes no filters (filter_pushdown = false,
src/scanner/read_arrow.cpp:47-48), DuckDB's filter above the scan keeps the same rows: the check is python's own evaluation
of the predicate over pyarrow's values, and the oracle's scalar CNF evaluator (SQL rules: a comparison with NULL is not
true)."""
import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pytest

import duckdb_arrow_amd as da
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

MODES = ["MAIL", "SHIP", "AIR", "REG AIR", "TRUCK", "RAIL", "FOB"]
INSTR = ["DELIVER IN PERSON", "COLLECT COD", "NONE", "TAKE BACK RETURN"]          # 17 / 11 / 4 / 16 bytes: inline and heap
LONG = ["a string that is much longer than the inline twelve bytes %d" % i for i in range(5)]


@pytest.fixture(scope="module")
def con():
    return da.Connection(0)


@pytest.fixture(scope="module")
def table():
    rng = np.random.default_rng(12)
    n = 30000
    pick = lambda xs, p_null: pa.array([None if rng.random() < p_null else xs[int(rng.integers(0, len(xs)))] for _ in range(n)])
    return pa.table({
        "k": pa.array(np.arange(n, dtype=np.int64)),
        "mode": pick(MODES, 0.05),
        "instr": pick(INSTR, 0.1),
        "flag": pa.array([None if rng.random() < 0.02 else "ANR"[int(rng.integers(0, 3))] for _ in range(n)]),
        "long": pick(LONG + ["a string that is much longer than the inline twelve bytes X", ""], 0.1).cast(pa.large_string()),
        "blob": pa.array([None if rng.random() < 0.1 else bytes(rng.integers(0, 3, 3, dtype=np.uint8)) for _ in range(n)], pa.binary(3)),
        "q": pa.array(rng.integers(0, 50, n).astype(np.int32)),
    })


def _eval(expr, cols, i):
    if expr[0] in ("and", "or") and isinstance(expr[1], tuple):
        rs = [_eval(e, cols, i) for e in expr[1:]]
        return all(rs) if expr[0] == "and" else any(rs)
    v, op = cols[expr[0]][i], expr[1]
    if op == "is null":
        return v is None
    if op == "is not null":
        return v is not None
    if v is None:
        return False
    norm = lambda c: c.encode() if isinstance(c, str) else c
    v = norm(v)
    if op == "in":
        return v in [norm(c) for c in expr[2]]
    c = norm(expr[2])
    if op == "starts_with":
        return v.startswith(c)
    return {"=": v == c, "<>": v != c, "<": v < c, "<=": v <= c, ">": v > c, ">=": v >= c}[op]   # bytes: byte-wise, a proper prefix first


EXPRS = [
    ("mode", "=", "MAIL"), ("mode", "<>", "MAIL"), ("mode", "in", ["MAIL", "SHIP"]), ("mode", "in", ["REG AIR", "nothing", ""]),
    ("instr", "=", "DELIVER IN PERSON"), ("instr", "<>", "DELIVER IN PERSON"), ("instr", "in", ["TAKE BACK RETURN", "NONE", "DELIVER IN PERSO"]),
    ("flag", "=", "R"), ("flag", "in", ["A", "N"]), ("flag", "=", ""), ("long", "=", LONG[3]), ("long", "in", [LONG[0], LONG[4], ""]),
    ("long", "<>", "a string that is much longer than the inline twelve bytes X"), ("long", "=", "a string that is much longer"),
    ("blob", "=", b"\x00\x01\x02"), ("blob", "in", [b"\x00\x00\x00", b"\x02\x02\x02"]),
    ("and", ("mode", "in", ["MAIL", "SHIP"]), ("q", "<", 24), ("instr", "=", "DELIVER IN PERSON")),          # TPC-H Q12 / Q19 shapes
    ("or", ("flag", "=", "R"), ("and", ("mode", "=", "AIR"), ("q", ">=", 40)), ("instr", "is null")),
    ("and", ("or", ("mode", "=", "FOB"), ("mode", "is null")), ("long", "<>", LONG[1])),
    # ordering and prefixes (DuckDB's default collation: byte-wise, a proper prefix sorts first)
    ("mode", "<", "REG AIR"), ("mode", ">=", "SHIP"), ("mode", ">", "RAIL"), ("mode", "<=", "AIR"), ("mode", "<", "REG"),
    ("instr", "<=", "DELIVER IN PERSON"), ("instr", ">", "DELIVER IN PERSO"), ("instr", "<", "DELIVER IN PERSON AND MORE"),
    ("long", ">", LONG[2]), ("long", "<", "a string that is much longer than the inline twelve bytes 3"), ("long", ">=", ""), ("long", "<=", ""),
    ("long", "starts_with", "a string that is much longer than the inline twelve bytes"), ("long", "starts_with", LONG[4][:-1] + "4"),
    ("mode", "starts_with", "R"), ("mode", "starts_with", ""), ("instr", "starts_with", "TAKE"), ("instr", "starts_with", "TAKE BACK RETURN!"),
    ("blob", ">=", b"\x01"), ("blob", "<", b"\x01\x00\x00"), ("blob", "starts_with", b"\x02\x02"), ("flag", "<", "N"),
    ("and", ("mode", ">", "AIR"), ("mode", "<", "SHIP"), ("q", "<", 25)),
    ("or", ("instr", "starts_with", "COLLECT"), ("and", ("long", ">", LONG[3]), ("flag", ">=", "N"))),
]


@pytest.mark.parametrize("expr", EXPRS, ids=[str(e)[:70] for e in EXPRS])
@pytest.mark.parametrize("compact", [False, True])
def test_string_predicates_equal_python_and_the_oracle(con, table, tmp_path_factory, expr, compact):
    path = str(tmp_path_factory.mktemp("sflt") / "t.arrows")
    with ipc.new_stream(path, table.schema) as w:
        w.write_table(table, max_chunksize=7000)
    cols = {name: table.column(name).to_pylist() for name in table.column_names}
    want = [i for i in range(table.num_rows) if _eval(expr, cols, i)]
    rel = con.read_arrow(path, filter_compact=compact).project(["k", "instr", "long"]).filter(expr)
    got_k, got_instr, got_long = rel.fetch_columns()
    assert got_k == want
    assert got_instr == [cols["instr"][i] for i in want] and got_long == [cols["long"][i] for i in want]
    # the oracle's evaluator over the same values, for trees that are already AND-of-ORs
    def cnf(e):
        if e[0] == "and" and isinstance(e[1], tuple):
            return [c for kid in e[1:] for c in cnf(kid)]
        if e[0] == "or" and isinstance(e[1], tuple):
            if all(not (k[0] in ("and", "or") and isinstance(k[1], tuple)) for k in e[1:]):
                return [list(e[1:])]
            raise ValueError
        return [[e]]
    try:
        clauses = cnf(expr)
    except ValueError:
        clauses = None
    if clauses is not None:
        assert po.filter_cnf(clauses, {name: cols[name] for name in table.column_names}, table.num_rows).tolist() == want
    # count without projecting the filter columns
    rel = con.read_arrow(path).filter(expr)
    assert rel.count(detail=True)["selected"] == len(want)


def test_string_filter_errors_and_other_consumers(con, table, tmp_path):
    path = str(tmp_path / "t.arrows")
    with ipc.new_stream(path, table.schema) as w:
        w.write_table(table, max_chunksize=9000)
    with pytest.raises(da.MiError) as e:      # a byte string against an integer column
        con.read_arrow(path).filter(("q", "=", "five")).count()
    assert e.value.code == da._ffi.MI_ENOTSUP
    with pytest.raises(da.MiError) as e:      # an integer against a VARCHAR column
        con.read_arrow(path).filter(("mode", "=", 5)).count()
    assert e.value.code == da._ffi.MI_ENOTSUP
    # device-resident consumer and an LZ4 file: the filter kernel reads the bytes the K8 kernels produced
    packed = str(tmp_path / "t_lz4.arrows")
    with ipc.new_stream(packed, table.schema, options=ipc.IpcWriteOptions(compression="lz4")) as w:
        w.write_table(table, max_chunksize=9000)
    cols = {name: table.column(name).to_pylist() for name in table.column_names}
    expr = ("and", ("instr", "=", "DELIVER IN PERSON"), ("long", "in", [LONG[2], LONG[3]]))
    want = sum(1 for i in range(table.num_rows) if _eval(expr, cols, i))
    for kw in ({}, {"device_resident": True}):
        rel = con.read_arrow(packed, **kw).filter(expr)
        assert rel.count(detail=True)["selected"] == want


DICT_EXPRS = [
    ("author", "=", "alice"), ("author", "<>", "alice"), ("author", "in", ["bob", "carol the third of her name", "nobody"]),
    ("author", "in", []), ("comp", "=", "C++"), ("comp", "<>", "Python"), ("author", "is null"), ("author", "is not null"),
    ("and", ("author", "in", ["alice", "bob"]), ("comp", "<>", "C++"), ("q", "<", 30)),
    ("or", ("author", "=", "dave"), ("and", ("comp", "=", "Rust"), ("mode", "=", "MAIL"))),
    ("author", "<", "carol the third of her name"), ("author", ">=", "bob"), ("author", "starts_with", "carol the"),
    ("and", ("comp", ">", "C++"), ("comp", "<=", "Python"), ("author", "starts_with", "")),
]


@pytest.mark.parametrize("expr", DICT_EXPRS, ids=[str(e)[:70] for e in DICT_EXPRS])
def test_string_predicates_on_dictionary_encoded_columns(con, tmp_path_factory, expr):
    """The arrow-commits shape (BASELINE config 5): low-cardinality VARCHAR columns arrive dictionary-encoded.  A string predicate
    on such a column is matched against the dictionary once and against the rows by index; dictionaries change from record
    batch to record batch (replacement), entries and rows can be NULL."""
    rng = np.random.default_rng(8)
    n = 6000
    authors = ["alice", "bob", "carol the third of her name", "dave", None]
    comps = ["C++", "Python", "Rust", "Go"]
    batches = []
    for bi in range(4):
        # every batch brings its own dictionary (different order / subset): the stream carries replacement dictionaries
        order = list(rng.permutation(len(authors)))
        a_vals = [authors[i] for i in order][: 3 + bi % 3]
        a_idx = pa.array(rng.integers(0, len(a_vals), n).astype(np.int32), mask=rng.random(n) < 0.1)
        a = pa.DictionaryArray.from_arrays(a_idx, pa.array(a_vals, pa.string()))
        c_vals = comps[bi % 2:] + comps[: bi % 2]
        c = pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, len(c_vals), n).astype(np.int8)), pa.array(c_vals, pa.large_string()))
        batches.append(pa.record_batch([pa.array(np.arange(bi * n, (bi + 1) * n, dtype=np.int64)), a, c,
                                        pa.array([MODES[int(x)] for x in rng.integers(0, len(MODES), n)]),
                                        pa.array(rng.integers(0, 50, n).astype(np.int32))], names=["k", "author", "comp", "mode", "q"]))
    path = str(tmp_path_factory.mktemp("dflt") / "d.arrows")
    with ipc.new_stream(path, batches[0].schema) as w:
        for b in batches:
            w.write_batch(b)
    t = pa.Table.from_batches(batches)
    cols = {name: t.column(name).to_pylist() for name in t.column_names}
    want = [i for i in range(t.num_rows) if _eval(expr, cols, i)]
    rel = con.read_arrow(path, accept_dictionaries=True).project(["k", "author"]).filter(expr)
    got_k, got_author = rel.fetch_columns()
    assert got_k == want and got_author == [cols["author"][i] for i in want]
    assert con.read_arrow(path, accept_dictionaries=True).filter(expr).count(detail=True)["selected"] == len(want)
