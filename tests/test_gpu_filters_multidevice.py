"""Pushed-down predicates (K6 generalised), late materialisation through the selection vector, and the in-library
multi-device scan -- together BASELINE config 3: lineitem over a list of >= 8 files with an l_shipdate filter, record
batches dealt over several device contexts (on the one-GPU test box: several contexts on device 0).

The reference pushes no filters (filter_pushdown = false, src/scanner/read_arrow.cpp:47-48): the expected rows are the
ones DuckDB's filter above the scan would keep, computed with numpy and with the oracle's scalar CNF evaluation
(oracle_transcode.c orc_filter_cnf) -- parity unpinned by the reference by construction."""
import os

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pytest

import duckdb_arrow_amd as da
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

SHIP_LO, SHIP_HI = 8766, 9131   # 1994-01-01 <= l_shipdate < 1995-01-01


@pytest.fixture(scope="module")
def con():
    return da.Connection(0)


def g(golden_dir, rel):
    return os.path.join(golden_dir, rel)


def _table(n=9000, seed=3):
    """Integer-like columns of every width / signedness + NULLs + a string and a decimal, in 3 record batches."""
    rng = np.random.default_rng(seed)
    nul = lambda p: rng.random(n) < p
    cols = {
        "i8": pa.array(rng.integers(-100, 100, n).astype(np.int8), mask=nul(0.1)),
        "u16": pa.array(rng.integers(0, 60000, n).astype(np.uint16), mask=nul(0.1)),
        "i32": pa.array(rng.integers(-1000, 1000, n).astype(np.int32), mask=nul(0.2)),
        "u64": pa.array(rng.integers(0, 2**63, n).astype(np.uint64) * 2 + rng.integers(0, 2, n).astype(np.uint64), mask=nul(0.05)),
        "d": pa.array(rng.integers(8000, 10600, n).astype(np.int32), pa.date32()),
        "dec": pa.array([None if x else __import__("decimal").Decimal(int(v)) for x, v in zip(nul(0.1), rng.integers(0, 11, n))], pa.decimal128(15, 0)),
        "flag": pa.array(rng.random(n) < 0.5, mask=nul(0.3)),
        "s": pa.array(["row %d %s" % (i, "x" * int(k)) for i, k in enumerate(rng.integers(0, 30, n))], mask=nul(0.25)),
        "k": pa.array(np.arange(n, dtype=np.int64)),
    }
    return pa.table(cols)


def _write(t, path, chunk=4000, **kw):
    with ipc.new_stream(path, t.schema, **kw) as w:
        w.write_table(t, max_chunksize=chunk)


def _stored(t, name):
    """(stored integers, valid mask) of a pyarrow column the way the scan stores it."""
    col = t.column(name).combine_chunks()
    ok = ~np.asarray(col.is_null())
    ty = col.type
    if pa.types.is_decimal(ty):
        vals = np.array([int(v.as_py().scaleb(ty.scale)) if v.is_valid else 0 for v in col], np.int64)
    elif pa.types.is_boolean(ty):
        vals = np.array([bool(v.as_py()) if v.is_valid else False for v in col], np.uint8)
    elif pa.types.is_date32(ty):
        vals = np.asarray(col.cast(pa.int32()).fill_null(0))
    elif pa.types.is_string(ty):
        vals = np.zeros(len(col), np.int64)
    else:
        vals = np.asarray(col.fill_null(0))
    return vals, ok


def _words(ok):
    n = len(ok)
    return np.packbits(np.concatenate([ok, np.ones((-n) % 64, bool)]), bitorder="little").view(np.uint64).copy()


def _cnf_of(expr):
    """the test's own expression -> AND-of-ORs of leaves (only shapes that are already in that form)"""
    if expr[0] == "and":
        out = []
        for e in expr[1:]:
            out.extend(_cnf_of(e))
        return out
    if expr[0] == "or":
        return [[leaf for e in expr[1:] for clause in _cnf_of(e) for leaf in clause]]
    return [[expr]]


def _numpy_eval(expr, t):
    if expr[0] in ("and", "or") and isinstance(expr[1], tuple):
        parts = [_numpy_eval(e, t) for e in expr[1:]]
        out = parts[0]
        for p in parts[1:]:
            out = (out & p) if expr[0] == "and" else (out | p)
        return out
    vals, ok = _stored(t, expr[0])
    op = expr[1].lower()
    if op == "is null":
        return ~ok
    if op == "is not null":
        return ok
    if vals.dtype == np.uint64:   # python ints: uint64 against possibly negative constants
        py = [int(v) for v in vals]
        c = expr[2]
        f = {"=": lambda v: v == c, "<>": lambda v: v != c, "<": lambda v: v < c, "<=": lambda v: v <= c, ">": lambda v: v > c,
             ">=": lambda v: v >= c, "in": lambda v: v in set(c)}[op]
        return np.array([f(v) for v in py]) & ok
    v = vals.astype(np.int64)
    c = expr[2]
    m = {"=": lambda: v == c, "<>": lambda: v != c, "<": lambda: v < c, "<=": lambda: v <= c, ">": lambda: v > c, ">=": lambda: v >= c,
         "in": lambda: np.isin(v, np.array(list(c), np.int64))}[op]()
    return m & ok


EXPRS = [
    ("i32", "=", 7), ("i32", "<>", 7), ("i32", "<", -500), ("i32", "<=", -500), ("i32", ">", 990), ("i32", ">=", 990),
    ("i8", "in", [1, 2, 3, -100, 99]), ("u16", ">", 59000), ("u16", "<", -1), ("u64", ">=", 2**62), ("u64", "<", 5), ("u64", ">", -5),
    ("u64", "in", [1, 2, 3]), ("i32", "is null"), ("s", "is null"), ("s", "is not null"), ("flag", "=", 1), ("flag", "<>", 1),
    ("d", ">=", SHIP_LO), ("dec", "in", [5, 6, 7]), ("i32", "<", -2**63), ("i32", ">", 2**63 - 1), ("i32", "in", []),
    ("and", ("d", ">=", SHIP_LO), ("d", "<", SHIP_HI)),
    ("and", ("d", ">=", SHIP_LO), ("d", "<", SHIP_HI), ("dec", ">=", 5), ("dec", "<=", 7), ("i32", "<", 240)),
    ("or", ("i32", "<", -900), ("i32", "is null"), ("u16", "=", 12345)),
    ("and", ("or", ("i8", "<", 0), ("flag", "is null")), ("or", ("d", "<", 8100), ("d", ">", 10500), ("dec", "=", 3)), ("k", "<>", 17)),
    ("or", ("and", ("i32", ">", 0), ("i8", "<", 0)), ("and", ("d", "<", 8500), ("u16", ">", 1000))),   # needs distribution into CNF
    ("and", ("i32", "<>", 5), ("i32", "<>", 6), ("s", "is not null")),
]


@pytest.mark.parametrize("expr", EXPRS, ids=[str(e)[:60] for e in EXPRS])
@pytest.mark.parametrize("compact", [False, True])
def test_predicate_forms_equal_numpy_and_the_oracle(con, tmp_path_factory, expr, compact):
    """Every predicate form of SURVEY.md Appendix C (= <> < <= > >=, IS [NOT] NULL, IN, AND / OR trees over several
    columns, NULLs in every column): the rows the scan keeps are numpy's, and for trees already in AND-of-ORs form also the
    oracle's scalar evaluation.  compact = late materialisation: chunks hold exactly those rows, densely packed."""
    t = _table()
    path = str(tmp_path_factory.mktemp("flt") / "t.arrows")
    _write(t, path)
    want = np.nonzero(_numpy_eval(expr, t))[0]
    rel = con.read_arrow(path, filter_compact=compact).project(["k", "i32", "s", "dec"]).filter(expr)
    got_k, got_i32, got_s, got_dec = rel.fetch_columns()
    assert got_k == want.tolist()
    assert got_i32 == [t.column("i32")[int(i)].as_py() for i in want]
    assert got_s == [t.column("s")[int(i)].as_py() for i in want]
    dec = _stored(t, "dec")
    assert got_dec == [int(dec[0][i]) if dec[1][i] else None for i in want]
    # the oracle on the same decoded vectors (trees already in conjunctive form)
    try:
        clauses = _cnf_of(expr)
    except Exception:
        clauses = None
    if clauses is not None and expr != ("or", ("and", ("i32", ">", 0), ("i8", "<", 0)), ("and", ("d", "<", 8500), ("u16", ">", 1000))):
        cols = {}
        for clause in clauses:
            for leaf in clause:
                v, ok = _stored(t, leaf[0])
                cols[leaf[0]] = (v, _words(ok))
        assert po.filter_cnf(clauses, cols, t.num_rows).tolist() == want.tolist()


def test_count_with_filter_needs_no_projection_of_the_filter_column(con, tmp_path):
    t = _table(20000, seed=8)
    path = str(tmp_path / "t.arrows")
    _write(t, path, chunk=7000)
    want = int(_numpy_eval(("and", ("d", ">=", SHIP_LO), ("d", "<", SHIP_HI)), t).sum())
    for compact in (False, True):
        rel = con.read_arrow(path, filter_compact=compact).project(["k"]).filter_range("d", SHIP_LO, SHIP_HI)
        got = rel.count(detail=True)
        assert (got["rows"], got["selected"]) == (20000, want)


def test_filter_errors(con, golden_dir, tmp_path):
    path = g(golden_dir, "ref_data/test.arrows")
    with pytest.raises(da.MiError, match="does not exist"):
        con.read_arrow(path).filter(("nope", "=", 1)).count()
    with pytest.raises(da.MiError, match="needs an integer"):
        con.read_arrow(path).filter(("message", "=", 1)).count()
    with pytest.raises(da.MiError, match="filter_compact needs flat projected columns"):
        first = con.read_arrow(g(golden_dir, "edge_nested.arrows")).columns[0]
        con.read_arrow(g(golden_dir, "edge_nested.arrows"), filter_compact=True).filter((first, "is null")).count()
    big = ("or",) + tuple(("and", ("files", ">", i), ("files", "<", i + 2), ("merge", "=", 1)) for i in range(6))
    with pytest.raises(da.MiError, match="too complex|leaves"):
        con.read_arrow(path).filter(big).count()


def test_compact_chunks_are_full_flat_vectors(con, tmp_path):
    """filter_compact: the chunks of a record batch are consecutive 2048-row slices of the surviving rows (no selection
    vector, source_rows tells how many rows the batch had), and columns with NULLs keep their validity bits in step."""
    t = _table(30000, seed=12)
    path = str(tmp_path / "t.arrows")
    _write(t, path, chunk=30000)
    rel = con.read_arrow(path, filter_compact=True, unset_all_valid=True).project(["k", "i32", "flag"]).filter(("i8", ">=", -50))
    keep = _numpy_eval(("i8", ">=", -50), t)
    total = int(keep.sum())
    sizes, scanned = [], 0
    for ch in rel.chunks():
        assert not ch.sel and ch.sel_count == ch.size
        assert ch.columns[0].validity is None   # `k` has no NULLs: the mask stays unset
        sizes.append(ch.size)
        scanned += ch.source_rows
    assert sizes == [2048] * (total // 2048) + ([total % 2048] if total % 2048 else [])
    assert scanned == 30000


# ---------------------------------------------------------------------------------------- multi-device + config 3
def _split_stream(src_path, out_dir, n_files):
    """One IPC stream -> n_files streams holding consecutive runs of its record batches."""
    rd = ipc.open_stream(src_path)
    batches = list(rd)
    per = (len(batches) + n_files - 1) // n_files
    paths = []
    for i in range(n_files):
        part = batches[i * per: (i + 1) * per]
        p = os.path.join(out_dir, "part_%02d.arrows" % i)
        with ipc.new_stream(p, rd.schema) as w:
            for b in part:
                w.write_batch(b)
        paths.append(p)
    return paths, batches


@pytest.mark.parametrize("n_ctx", [1, 2, 3])
def test_multi_device_scan_returns_chunks_in_batch_order(golden_dir, tmp_path, n_ctx):
    """mi_scan_open_files_multi over a file list: whatever context decoded a record batch, chunks come back in the order of
    the unsharded scan (batch_index ascending, then chunk_offset), with the same values."""
    src = g(golden_dir, "lineitem_sf0_01_q6.arrows")
    t = ipc.open_stream(src).read_all()
    big = str(tmp_path / "many.arrows")
    with ipc.new_stream(big, t.schema) as w:
        w.write_table(t, max_chunksize=3000)      # ~20 record batches
    paths, batches = _split_stream(big, str(tmp_path), 8)
    con = da.Connection(0)
    single = con.read_arrow(paths)
    want = single.fetch_columns()
    ctxs = [da.Context(0) for _ in range(n_ctx)]
    rel = con.read_arrow(paths, contexts=ctxs)
    order, got = [], [[] for _ in rel.columns]
    for ch in rel.chunks():
        order.append((ch.batch_index, ch.chunk_offset))
        for o, c in zip(got, da.chunk_to_columns(ch, rel.fields)):
            o.extend(c)
    assert order == sorted(order) and len(set(b for b, _ in order)) == len(batches)
    assert got == want
    rel.close()
    single.close()


@pytest.mark.parametrize("compact", [False, True])
def test_config3_file_list_with_shipdate_filter_sharded_over_contexts_and_ranks(golden_dir, tmp_path, expected, compact):
    """BASELINE config 3 at golden size: lineitem split into 8 files, l_shipdate range pushed down, record batches dealt
    over 2 ranks x 2 device contexts.  The union of the shards = the unsharded scan = numpy on pyarrow's values; the Q6
    revenue over the selected rows is the reference's known answer 1193053.2253 (test/nodejs/arrow_test.js:423-424)."""
    src = g(golden_dir, "lineitem_sf0_01_q6.arrows")
    t = ipc.open_stream(src).read_all()
    big = str(tmp_path / "many.arrows")
    with ipc.new_stream(big, t.schema) as w:
        w.write_table(t, max_chunksize=2500)
    paths, batches = _split_stream(big, str(tmp_path), 8)
    ship = np.asarray(t.column("l_shipdate").cast(pa.int32()))
    keep = (ship >= SHIP_LO) & (ship < SHIP_HI)
    cols = ["l_shipdate", "l_quantity", "l_extendedprice", "l_discount"]
    con = da.Connection(0)
    rows = {}
    scanned = selected = 0
    for rank in range(2):
        ctxs = [da.Context(0), da.Context(0)]
        rel = con.read_arrow(paths, contexts=ctxs, rank=rank, world=2, filter_compact=compact).project(cols).filter_range("l_shipdate", SHIP_LO, SHIP_HI)
        last = -1
        for ch in rel.chunks():
            assert ch.batch_index % 2 == rank and ch.batch_index >= last
            last = ch.batch_index
            vals = da.chunk_to_columns(ch, rel._out_fields)
            idx = [ch.sel[i] for i in range(ch.sel_count)] if ch.sel else range(ch.size)
            rows.setdefault(ch.batch_index, []).extend(zip(*[[c[i] for i in idx] for c in vals]))
            selected += ch.sel_count
        rel.close()
        cnt = con.read_arrow(paths, contexts=[da.Context(0), da.Context(0)], rank=rank, world=2, filter_compact=compact) \
            .project(["l_quantity"]).filter_range("l_shipdate", SHIP_LO, SHIP_HI).count(detail=True)
        scanned += cnt["rows"]
    assert scanned == t.num_rows and selected == int(keep.sum())
    got = [r for b in sorted(rows) for r in rows[b]]
    want_idx = np.nonzero(keep)[0]
    assert [r[0] for r in got] == ship[want_idx].tolist()
    revenue = sum(p * d for s, q, p, d in got if 5 <= d <= 7 and q < 2400)
    assert revenue == expected["kat"]["q6_sf0_01_revenue_scale4"] == 11930532253
    # the fused aggregate over the same file list, every context draining on its own thread
    rel = con.read_arrow(paths, contexts=[da.Context(0) for _ in range(3)])
    total, sel, n = rel.sum_product("l_extendedprice", "l_discount", [("l_shipdate", SHIP_LO, SHIP_HI), ("l_discount", 5, 8), ("l_quantity", -2**63, 2400)])
    assert (total, n) == (11930532253, t.num_rows) and sel == expected["kat"]["q6_sf0_01_rows_passing"]


def test_full_size_config3_selected_count_equals_torch(tmp_path_factory):
    """SF10 lineitem (59 986 052 rows) written as 8 files; the pushed-down l_shipdate range over 2 device contexts selects
    exactly the rows a torch filter over the decoded column selects (size-independent property at BASELINE's size)."""
    torch = pytest.importorskip("torch")
    from duckdb_arrow_amd.hbm import HbmStream
    buf, info = da.synth_lineitem_stream(scale_factor=10.0, seed=42)
    d = "/dev/shm/mi_cfg3_%d" % os.getpid()
    os.makedirs(d, exist_ok=True)
    paths = []
    try:
        offs = info["batch_offsets"]
        head = buf[: offs[0]]
        nb = info["n_batches"]
        per = (nb + 7) // 8
        for i in range(8):
            lo, hi = offs[i * per], offs[min(nb, (i + 1) * per)]
            p = os.path.join(d, "lineitem_%d.arrows" % i)
            with open(p, "wb") as f:
                f.write(head.tobytes())
                f.write(buf[lo:hi].tobytes())
                f.write(b"\xff\xff\xff\xff\x00\x00\x00\x00")
            paths.append(p)
        ctx = da.Context(0)
        hs = HbmStream(ctx, buf, columns=["l_shipdate"], memory="torch")
        hs.launch()
        assert hs.status() == 0
        want = 0
        for lay in hs.layout:
            e = lay["columns"][0]
            ship = hs.d_out[e["data_off"]: e["data_off"] + 4 * lay["nrows"]].view(torch.int32)
            want += int(((ship >= SHIP_LO) & (ship < SHIP_HI)).sum().item())
        hs.close()
        con = da.Connection(0)
        for compact in (False, True):
            rel = con.read_arrow(paths, contexts=[da.Context(0), da.Context(0)], filter_compact=compact).filter_range("l_shipdate", SHIP_LO, SHIP_HI)
            got = rel.count(detail=True)
            assert got["rows"] == info["n_rows"] == 59986052 and got["selected"] == want
            rel.close()
    finally:
        for p in paths:
            os.remove(p)
        os.rmdir(d)
