"""The scan operator (read_arrow / scan_arrow_ipc) end to end on the GPU, written after the reference's own tests:
test/sql/read_arrow.test, test/sql/read_arrow_file.test, test/sql/multifile_reading.test,
test/python/test_arrow_ipc_scan.py.  Expected values are the reference's known answers (tests/golden/expected.json)."""
import os

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pytest

import duckdb_arrow_amd as da
from duckdb_arrow_amd import _ffi

from helpers import canon_python, column_digest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def con():
    return da.Connection(0)


def g(golden_dir, rel):
    return os.path.join(golden_dir, rel)


# ---------------------------------------------------------------------------------------- read_arrow.test
def test_count_star_test_arrows(con, golden_dir):
    rel = con.read_arrow(g(golden_dir, "ref_data/test.arrows"))
    assert rel.columns == ["commit", "time", "files", "merge", "message"]
    assert rel.types == ["VARCHAR", "TIMESTAMP WITH TIME ZONE", "INTEGER", "BOOLEAN", "VARCHAR"]
    assert rel.count() == 15487


def test_unknown_named_parameter(con, golden_dir):
    with pytest.raises(da.MiError, match='Invalid named parameter "made_up_option" for function read_arrow'):
        con.read_arrow(g(golden_dir, "ref_data/test.arrows"), made_up_option=False)


def test_filter_and_projection(con, golden_dir):
    """SELECT message FROM read_arrow(...) WHERE "commit" = 'fa5f0299...'"""
    rel = con.read_arrow(g(golden_dir, "ref_data/test.arrows")).project(["message", "commit"])
    rows = [m for m, c in rel.fetchall() if c == "fa5f0299f046c46e1b2f671e5e3b4f1956522711"]
    assert rows == ["ARROW-1: Initial Arrow Code Commit"]


def test_filter_over_multiple_batches(con, golden_dir):
    """SELECT count(*) ... WHERE dayname(time::TIMESTAMP) = 'Wednesday' -> 2927"""
    (times,) = con.read_arrow(g(golden_dir, "ref_data/test.arrows")).project(["time"]).fetch_columns()
    days = np.floor_divide(np.array(times, dtype=np.int64), 86400000000)
    assert int(np.sum((days + 3) % 7 == 2)) == 2927


def test_all_columns_match_pyarrow(con, golden_dir, expected):
    for rel_path in ("ref_data/test.arrows", "edge_types.arrows", "edge_types2.arrows", "edge_empty.arrows",
                     "lineitem_sf0_01_head.arrows"):
        rel = con.read_arrow(g(golden_dir, rel_path))
        cols = rel.fetch_columns()
        exp = expected[rel_path]
        for name, dt, values in zip(rel.columns, rel.types, cols):
            assert len(values) == exp["rows"]
            canon = values
            if dt in ("FLOAT", "DOUBLE"):
                canon = [None if v is None else ("nan" if v != v else repr(v)) for v in values]
            elif dt == "INTERVAL" and name.startswith("dur_"):
                canon = [None if v is None else v[2] for v in values]  # durations: months = days = 0
            elif dt == "INTERVAL":
                canon = [None if v is None else list(v) for v in values]
            elif dt == "BLOB":
                canon = [None if v is None else "b:" + v.hex() for v in values]
            assert column_digest(canon) == exp["columns"][name], (rel_path, name)


# ---------------------------------------------------------------------------------------- read_arrow_file.test
def test_ipc_file_format(con, golden_dir):
    rows = con.read_arrow(g(golden_dir, "ref_data/fruit.arrow")).fetchall()
    assert len(rows) == 6 and sum(r[2] is None for r in rows) == 2


# ---------------------------------------------------------------------------------------- multifile_reading.test
GLOB_ROWS = [("apple", "gala", 134.2), ("orange", "navel", 142.1), ("apple", "honeycrisp", 158.6),
             ("orange", "valencia", 96.7), ("apple", "fuji", None), ("orange", "cara cara", None)]


def test_file_list(con, golden_dir):
    p = g(golden_dir, "ref_data/test.arrows")
    assert con.read_arrow([p, p]).count() == 30974


def test_glob(con, golden_dir):
    assert con.read_arrow(g(golden_dir, "ref_data/multifile/glob/*.arrow")).fetchall() == GLOB_ROWS


def test_glob_projection(con, golden_dir):
    rel = con.read_arrow(g(golden_dir, "ref_data/multifile/glob/*.arrow")).project(["weight", "variety"])
    assert rel.fetchall() == [(w, v) for _, v, w in GLOB_ROWS]


def test_mismatching_schemas(con, golden_dir):
    for a, b in (("ref_data/test.arrows", "ref_data/multifile/glob/f1.arrow"),
                 ("ref_data/multifile/fruit_extra.arrows", "ref_data/multifile/glob/f1.arrow")):
        with pytest.raises(da.MiError, match="If you are trying to read files with different schemas, try setting union_by_name=True"):
            con.read_arrow([g(golden_dir, a), g(golden_dir, b)]).fetchall()


def test_union_by_name(con, golden_dir):
    rel = con.read_arrow([g(golden_dir, "ref_data/multifile/fruit_extra.arrows"), g(golden_dir, "ref_data/multifile/glob/f1.arrow")],
                         union_by_name=True)
    assert rel.columns == ["fruit", "variety", "weight", "tasteness"]
    assert rel.fetchall() == [("apple", "pink lady", 2.2, 10.0), ("orange", "jiha", None, None),
                              ("apple", "gala", 134.2, None), ("orange", "navel", 142.1, None)]


def test_different_column_order(con, golden_dir):
    rel = con.read_arrow([g(golden_dir, "ref_data/multifile/different_order.arrows"), g(golden_dir, "ref_data/multifile/glob/f1.arrow")])
    assert rel.columns == ["fruit", "weight", "variety"]
    assert sorted(rel.fetchall(), key=lambda r: (r[0], r[1] is None, r[1] or 0)) == \
        [("apple", 2.2, "pink lady"), ("apple", 134.2, "gala"), ("orange", 142.1, "navel"), ("orange", None, "jiha")]


def test_different_types_are_left_to_duckdb(con, golden_dir):
    """Cross-file casts (VARCHAR '2.2' -> DOUBLE) are MultiFileReader::FinalizeChunk's job above the path; the scan
    reports them instead of guessing.  The first file still decides the bound type (typeof(#3) checks)."""
    rel = con.read_arrow([g(golden_dir, "ref_data/multifile/different_type.arrows"), g(golden_dir, "ref_data/multifile/glob/f1.arrow")])
    assert rel.types[2] == "VARCHAR"
    rel2 = con.read_arrow([g(golden_dir, "ref_data/multifile/glob/f1.arrow"), g(golden_dir, "ref_data/multifile/different_type.arrows")])
    assert rel2.types[2] == "DOUBLE"
    with pytest.raises(da.MiError, match="cross-file casts") as e:
        rel2.fetchall()
    assert e.value.code == _ffi.MI_ENOTSUP


def test_filename_option(con, golden_dir):
    rel = con.read_arrow(g(golden_dir, "ref_data/multifile/glob/*.arrow"), filename=True)
    assert rel.columns == ["fruit", "variety", "weight", "filename"]
    rows = rel.fetchall()
    assert [r[:3] for r in rows] == GLOB_ROWS
    assert [os.path.basename(r[3]) for r in rows] == ["f1.arrow", "f1.arrow", "f2.arrow", "f2.arrow", "f3.arrow", "f3.arrow"]
    assert all(r[3].endswith("ref_data/multifile/glob/" + os.path.basename(r[3])) for r in rows)


def test_hive_partitioning(con, golden_dir):
    rel = con.read_arrow(g(golden_dir, "ref_data/multifile/hive/*/*.arrow"), hive_partitioning=True)
    assert rel.columns == ["fruit", "variety", "weight", "part"]
    assert rel.fetchall() == [("apple", "gala", 134.2, "a"), ("orange", "navel", 142.1, "a"), ("apple", "honeycrisp", 158.6, "a"),
                              ("orange", "valencia", 96.7, "a"), ("apple", "gala", 134.2, "b"), ("orange", "navel", 142.1, "b"),
                              ("apple", "fuji", None, "b"), ("orange", "cara cara", None, "b")]


# ---------------------------------------------------------------------------------------- test_arrow_ipc_scan.py
def get_record_batch():
    data = [pa.array([1, 2, 3, 4]), pa.array(["foo", "bar", "baz", None]), pa.array([True, None, False, True])]
    return pa.record_batch(data, names=["f0", "f1", "f2"])


EXPECTED_5 = [(1, "foo", True), (2, "bar", None), (3, "baz", False), (4, None, True)] * 5


def stream_of(n):
    batch = get_record_batch()
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, batch.schema) as writer:
        for _ in range(n):
            writer.write_batch(batch)
    return sink.getvalue()


def test_single_buffer(con):
    buffer = stream_of(5)
    with pa.BufferReader(buffer) as buf_reader:
        msg_reader = ipc.MessageReader.open_stream(buf_reader)
        assert con.from_arrow(msg_reader).fetchall() == EXPECTED_5


def test_multi_buffers(con):
    """scan_arrow_ipc over several {ptr, size} buffers: schema in the first, batches spread over the rest."""
    buffer = stream_of(5).to_pybytes()
    with pa.BufferReader(buffer) as buf_reader:
        msg_reader = ipc.MessageReader.open_stream(buf_reader)
        parts = []
        while True:
            try:
                parts.append(msg_reader.read_next_message().serialize().to_pybytes())
            except StopIteration:
                break
    rel = con.scan_arrow_ipc([parts[0] + parts[1]] + parts[2:])
    assert rel.columns == ["f0", "f1", "f2"] and rel.types == ["BIGINT", "VARCHAR", "BOOLEAN"]
    assert rel.fetchall() == EXPECTED_5


def test_empty_schema_is_rejected(con):
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, pa.schema([])):
        pass
    with pytest.raises(da.MiError, match="Provided table/dataframe must have at least one column"):
        con.scan_arrow_ipc([sink.getvalue().to_pybytes()])


# ---------------------------------------------------------------------------------------- beyond the reference
def test_row_group_sharding_covers_every_batch_once(con, golden_dir):
    """record batches shard over ranks with no collective; the union of the shards is the table."""
    p = g(golden_dir, "ref_data/test.arrows")
    whole = con.read_arrow(p).project(["commit", "files"]).fetchall()
    shards = [con.read_arrow(p, rank=r, world=3).project(["commit", "files"]) for r in range(3)]
    per_rank = []
    for s in shards:
        rows, idx = [], set()
        for ch in s.chunks():
            idx.add(ch.batch_index)
            cols = da.chunk_to_columns(ch, s._out_fields)
            rows.extend(zip(*cols))
        per_rank.append((rows, idx))
    assert [sorted(i) for _, i in per_rank] == [[0, 3, 6, 9, 12, 15], [1, 4, 7, 10, 13], [2, 5, 8, 11, 14]]
    assert sorted(r for rows, _ in per_rank for r in rows) == sorted(whole)


def test_filter_pushdown_selection_vector(con, golden_dir, expected):
    """K6 inside the scan: 1994-01-01 <= l_shipdate < 1995-01-01 selects the rows DuckDB's filter would keep."""
    rel = con.read_arrow(g(golden_dir, "lineitem_sf0_01_q6.arrows")).filter_range("l_shipdate", 8766, 9131)
    qty, price, disc, ship = rel.fetch_columns()
    assert len(ship) == expected["kat"]["shipdate_1994_selected"] and min(ship) >= 8766 and max(ship) < 9131
    keep = [(p, d) for q, p, d in zip(qty, price, disc) if 5 <= d <= 7 and q < 2400]
    assert len(keep) == 1191 and sum(p * d for p, d in keep) == 11930532253  # Q6 = 1193053.2253


def test_dictionary_encoded_columns(con, golden_dir, expected):
    """BASELINE config 5 (the reference cannot read these: base_stream_reader.cpp:86-96)."""
    with pytest.raises(da.MiError, match="dictionary-encoded"):
        con.read_arrow(g(golden_dir, "edge_dict.arrows")).fetchall()
    rel = con.read_arrow(g(golden_dir, "edge_dict.arrows"), accept_dictionaries=True)
    assert rel.types == ["VARCHAR", "BIGINT", "INTEGER"]
    for name, values in zip(rel.columns, rel.fetch_columns()):
        assert column_digest(values) == expected["edge_dict.arrows"]["columns"][name], name


def test_delta_and_replacement_dictionaries(con, tmp_path):
    """DictionaryBatch.isDelta appends to the dictionary, a plain DictionaryBatch replaces it; batches already in flight
    keep the version they were enqueued with (beyond the reference)."""
    sch = pa.schema([("k", pa.dictionary(pa.int8(), pa.string())), ("v", pa.int32())])

    def batch(values, dictionary, n0):
        idx = pa.array([None if v is None else dictionary.index(v) for v in values], pa.int8())
        return pa.record_batch([pa.DictionaryArray.from_arrays(idx, pa.array(dictionary)), pa.array(range(n0, n0 + len(values)), pa.int32())],
                               schema=sch)

    rows = [(["a", "b", None, "a"], ["a", "b"]),
            (["c", "a", None, "a much longer dictionary value"], ["a", "b", "c", "a much longer dictionary value"]),   # delta
            (["x", None, "y", "x"], ["x", "y"]),                                                                       # replacement
            (["y", "z"], ["x", "y", "z"])]                                                                              # delta again
    p = str(tmp_path / "delta.arrows")
    with ipc.new_stream(p, sch, options=ipc.IpcWriteOptions(emit_dictionary_deltas=True)) as w:
        n0 = 0
        for values, dictionary in rows:
            w.write_batch(batch(values, dictionary, n0))
            n0 += len(values)
    kinds = [(e["type"]) for e in da.Reader(path=p).index()]
    assert kinds.count(2) == 4 and kinds.count(3) == 4
    got = con.read_arrow(p, accept_dictionaries=True).fetchall()
    want = [(v, i) for i, v in enumerate(v for values, _ in rows for v in values)]
    assert got == want
    assert got == [(r["k"], r["v"]) for r in ipc.open_stream(p).read_all().to_pylist()]


def test_hbm_resident_mode_refuses_a_dictionary_replaced_in_mid_stream(con):
    """mi_hbm_open keeps ONE dictionary per id for the whole resident stream; a replacement after a record batch has used the
    id would re-interpret the earlier batches' indices, so it is refused (the scan operator versions dictionaries instead:
    test_delta_and_replacement_dictionaries).  A stream whose dictionaries do not change is accepted."""
    from duckdb_arrow_amd.hbm import HbmStream
    sch = pa.schema([("k", pa.dictionary(pa.int8(), pa.string()))])

    def stream(dicts):
        sink = pa.BufferOutputStream()
        with ipc.new_stream(sink, sch) as w:
            for d in dicts:
                w.write_batch(pa.record_batch([pa.DictionaryArray.from_arrays(pa.array([0, 1, 0], pa.int8()), pa.array(d))], schema=sch))
        return np.frombuffer(sink.getvalue().to_pybytes(), np.uint8)

    with pytest.raises(da.MiError, match="replaced in mid-stream") as e:
        HbmStream(con.ctx, stream([["a", "b"], ["x", "y"]]), accept_dictionaries=True)
    assert e.value.code == _ffi.MI_ENOTSUP
    hs = HbmStream(con.ctx, stream([["a", "b"], ["a", "b"]]), accept_dictionaries=True)   # pyarrow re-sends nothing: one DictionaryBatch
    hs.launch()
    assert hs.status() == 0
    hs.close()


def test_device_resident_chunks(con, golden_dir):
    """device_resident = 1: vectors stay in HBM for a GPU consumer (no D2H); pointers are device addresses."""
    import torch
    rel = con.read_arrow(g(golden_dir, "ref_data/test.arrows"), device_resident=True).project(["files"])
    host = con.read_arrow(g(golden_dir, "ref_data/test.arrows")).project(["files"]).fetch_columns()[0]
    got = []
    for ch in rel.chunks():
        n = ch.size
        t = torch.empty(n, dtype=torch.int32, device="cuda")
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ch.columns[0].data), C.c_size_t(4 * n), 3) == 0  # D2D
        got.extend(t.cpu().tolist())
    assert got == host


def test_zstd_compressed_file(con, golden_dir, tmp_path):
    """arrow_testing.test:60-64 reads zstd IPC files: CPU decompression (like the reference), then the GPU path."""
    t = ipc.open_stream(g(golden_dir, "ref_data/test.arrows")).read_all()
    p = str(tmp_path / "zstd.arrows")
    with ipc.new_stream(p, t.schema, options=ipc.IpcWriteOptions(compression="zstd")) as w:
        w.write_table(t, max_chunksize=4000)
    assert os.path.getsize(p) < os.path.getsize(g(golden_dir, "ref_data/test.arrows"))
    a = con.read_arrow(p).fetch_columns()
    b = con.read_arrow(g(golden_dir, "ref_data/test.arrows")).fetch_columns()
    assert a == b


def test_progress_reaches_100(con, golden_dir):
    rel = con.read_arrow(g(golden_dir, "ref_data/test.arrows"))
    assert rel.progress() < 100
    rel.count()
    assert rel.progress() == pytest.approx(100.0, abs=0.5)


# ---------------------------------------------------------------------------------------- nested types + string views
def test_nested_types_and_string_views(con, golden_dir, expected):
    """Lists, lists of lists, structs, lists of structs, fixed-size lists, maps, large lists, utf8_view / binary_view with
    NULLs at every level (ArrowToDuckDB list / struct / map / array / view branches, arrow_conversion.cpp in DuckDB):
    every 2048-row chunk carries child vectors whose list entries are relative to the chunk's child window."""
    rel = con.read_arrow(g(golden_dir, "edge_nested.arrows"))
    assert rel.types == ["INTEGER[]", "VARCHAR[]", "BIGINT[][]", "STRUCT(a INTEGER, b VARCHAR)", "STRUCT(x BIGINT, y VARCHAR)[]",
                         "SMALLINT[3]", "MAP(VARCHAR, INTEGER)", "DOUBLE[]", "VARCHAR", "BLOB"]
    exp = expected["edge_nested.arrows"]
    cols = rel.fetch_columns()
    for name, values in zip(rel.columns, cols):
        assert len(values) == exp["rows"]
        assert sum(v is None for v in values) == exp["null_counts"][name], name
        assert column_digest(canon_python(values)) == exp["columns"][name], name


def test_nested_projection_and_buffers(con, golden_dir, expected):
    """scan_arrow_ipc over in-memory buffers with a projection that keeps only nested columns (in another order)."""
    data = np.fromfile(g(golden_dir, "edge_nested.arrows"), np.uint8)
    rel = con.scan_arrow_ipc([data]).project(["mp", "ll", "sv"])
    exp = expected["edge_nested.arrows"]
    for name, values in zip(rel.columns, rel.fetch_columns()):
        assert column_digest(canon_python(values)) == exp["columns"][name], name


def test_nested_chunks_are_window_relative(con, golden_dir):
    """Every chunk's list vector addresses its OWN child vector: offsets start at 0 and end at the child's count."""
    rel = con.read_arrow(g(golden_dir, "edge_nested.arrows")).project(["l_i"])
    seen = 0
    for chunk in rel.chunks():
        v = chunk.columns[0]
        assert v.kind == _ffi.K_LIST32 and v.n_children == 1
        n = chunk.size
        ent = np.ctypeslib.as_array(_ffi.C.cast(v.data, _ffi.C.POINTER(_ffi.C.c_uint64)), shape=(n * 2,)).reshape(-1, 2)
        assert int(ent[0, 0]) == 0
        assert int(ent[-1, 0] + ent[-1, 1]) == v.children[0].count
        assert np.all(ent[1:, 0] == ent[:-1, 0] + ent[:-1, 1])
        seen += n
    assert seen == 6605


# ---------------------------------------------------------------------------------------- zero-copy direct columns
def test_zero_copy_direct_columns(con, golden_dir, expected):
    """zero_copy_direct: plain fixed-width columns without NULLs alias the record-batch body (the reference's
    DirectConversion) and carry no validity mask; everything else still goes through the kernels.  Same values."""
    path = g(golden_dir, "lineitem_sf0_01_head.arrows")
    off = con.read_arrow(path, zero_copy_direct=False)      # every vector materialised by a kernel
    want = off.fetch_columns()
    assert off.stats()["aliased_bytes"] == 0 and off.stats()["d2h_bytes"] > 0
    auto = con.read_arrow(path)                              # the default: on, for host consumers too
    assert auto.fetch_columns() == want
    st = auto.stats()
    assert st["aliased_bytes"] > 0 and st["d2h_bytes"] < off.stats()["d2h_bytes"] and st["h2d_bytes"] < off.stats()["h2d_bytes"]
    rel = con.read_arrow(path, zero_copy_direct=True)
    aliased = set()
    got = [[] for _ in rel.columns]
    for ch in rel.chunks():
        for ci in range(ch.n_columns):
            if not ch.columns[ci].validity:
                aliased.add(rel.columns[ci])
        for o, c in zip(got, da.chunk_to_columns(ch, rel._out_fields)):
            o.extend(c)
    assert got == want
    assert {"l_orderkey", "l_partkey", "l_suppkey", "l_linenumber", "l_shipdate", "l_commitdate", "l_receiptdate"} <= aliased
    assert not ({"l_quantity", "l_comment", "l_returnflag"} & aliased)
    # NULL-bearing and nested files: nothing breaks, columns with NULLs are transcoded as before
    for rel_path in ("edge_types.arrows", "edge_nested.arrows", "ref_data/test.arrows"):
        a = con.read_arrow(g(golden_dir, rel_path), zero_copy_direct=False).fetch_columns()
        b = con.read_arrow(g(golden_dir, rel_path), zero_copy_direct=True).fetch_columns()
        assert [canon_python(c) for c in a] == [canon_python(c) for c in b], rel_path
    # with a pushed-down filter on a direct column the filter column itself is still materialised on the GPU
    rel = con.read_arrow(g(golden_dir, "lineitem_sf0_01_q6.arrows"), zero_copy_direct=True).project(["l_shipdate", "l_discount"])
    ship, _ = rel.filter_range("l_shipdate", 8766, 9131).fetch_columns()
    assert len(ship) == expected["kat"]["shipdate_1994_selected"]


# ---------------------------------------------------------------------------------------- fused consumer (TPC-H Q6)
Q6_FILTERS = [("l_shipdate", 8766, 9131), ("l_discount", 5, 8), ("l_quantity", -2**63, 2400)]


def test_q6_fused_on_the_gpu_known_answer(con, golden_dir, expected):
    """benchmark/lineitem.py:22-34 / test/nodejs/arrow_test.js:423-424: sum(l_extendedprice * l_discount) with the Q6
    predicates = 1193053.2253 on lineitem SF0.01, computed on the GPU; only 32 bytes come back."""
    rel = con.read_arrow(g(golden_dir, "lineitem_sf0_01_q6.arrows"))
    total, selected, scanned = rel.sum_product("l_extendedprice", "l_discount", Q6_FILTERS)
    assert total == 11930532253 and selected == 1191
    assert scanned == expected["lineitem_sf0_01_q6.arrows"]["rows"]


def test_q6_fused_matches_numpy_on_synthetic_lineitem(con, tmp_path):
    buf, info = da.synth_lineitem_stream(scale_factor=0.05, seed=9)
    path = str(tmp_path / "li.arrows")
    buf.tofile(path)
    cols = con.read_arrow(path).project(["l_extendedprice", "l_discount", "l_quantity", "l_shipdate"]).fetch_columns()
    price, disc, qty, ship = (np.array(c, dtype=object) for c in cols)
    keep = [(8766 <= s < 9131) and (5 <= d < 8) and q < 2400 for s, d, q in zip(ship, disc, qty)]
    want = sum(int(p) * int(d) for p, d, k in zip(price, disc, keep) if k)
    total, selected, scanned = con.read_arrow(path).sum_product("l_extendedprice", "l_discount", Q6_FILTERS)
    assert (total, selected, scanned) == (want, sum(keep), info["n_rows"])
    # sharded: the partial sums of two ranks add up
    parts = [con.read_arrow(path, rank=r, world=2).sum_product("l_extendedprice", "l_discount", Q6_FILTERS) for r in range(2)]
    assert sum(p[0] for p in parts) == want and sum(p[2] for p in parts) == info["n_rows"]


def test_fused_aggregate_semantics(con, tmp_path):
    """NULL in a filter column drops the row, NULL in a factor contributes nothing, negative products and sums beyond
    64 bits are exact (128-bit accumulation), non-integer columns are refused."""
    n = 5000
    rng = np.random.default_rng(4)
    a = pa.array(rng.integers(-2**62, 2**62, n), pa.int64(), mask=rng.random(n) < 0.1)
    b = pa.array(rng.integers(-2**31, 2**31, n).astype(np.int32), mask=rng.random(n) < 0.1)
    f = pa.array(rng.integers(0, 100, n).astype(np.int16), mask=rng.random(n) < 0.2)
    s = pa.array(["x"] * n)
    path = str(tmp_path / "agg.arrows")
    with ipc.new_stream(path, pa.schema([("a", a.type), ("b", b.type), ("f", f.type), ("s", s.type)])) as w:
        w.write_batch(pa.record_batch([a, b, f, s], names=["a", "b", "f", "s"]))
    al, bl, fl = a.to_pylist(), b.to_pylist(), f.to_pylist()
    keep = [x is not None and 10 <= x < 60 for x in fl]
    want = sum(x * y for x, y, k in zip(al, bl, keep) if k and x is not None and y is not None)
    total, selected, scanned = con.read_arrow(path).sum_product("a", "b", [("f", 10, 60)])
    assert total == want and abs(want) > 2**64 and selected == sum(keep) and scanned == n
    assert con.read_arrow(path).sum_product("b", "b")[1] == n        # no filter: every row selected
    with pytest.raises(da.MiError, match="not a fixed-width integer-like column"):
        con.read_arrow(path).sum_product("a", "s")


def test_two_worker_threads_with_their_own_contexts(golden_dir, tmp_path):
    """The boundary's threading contract: one context per (device, worker), entry points re-entrant across
    contexts -- DuckDB scans different files from different threads.  Two threads scan concurrently (ctypes drops the
    GIL inside the calls), sharing only the process-wide I/O pool."""
    import threading
    buf, info = da.synth_lineitem_stream(scale_factor=0.2, seed=12)
    paths = []
    for i in range(2):
        p = str(tmp_path / ("t%d.arrows" % i))
        buf.tofile(p)
        paths.append(p)
    results, errors = {}, []

    def worker(i):
        try:
            c = da.Connection(0)
            for rep in range(3):
                rows = c.read_arrow(paths[i]).count()
                total, sel, scanned = c.read_arrow(paths[i]).sum_product("l_extendedprice", "l_discount", Q6_FILTERS)
                results[(i, rep)] = (rows, total, sel, scanned)
            c.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    assert len(set(results.values())) == 1 and next(iter(results.values()))[0] == info["n_rows"]


def test_many_files_and_early_close_with_read_ahead(con, tmp_path):
    """The read-ahead thread leases a fixed set of pinned staging buffers: more files than buffers (every finished reader
    must give its last body back), consumers that stop early (close with batches still queued), and an error in a later
    file surfacing only after the rows before it."""
    t = pa.table({"a": list(range(5000)), "s": ["value %d" % i for i in range(5000)]})
    paths = []
    for i in range(12):
        p = str(tmp_path / ("f%02d.arrows" % i))
        with ipc.new_stream(p, t.schema) as w:
            w.write_table(t, max_chunksize=1000)
        paths.append(p)
    assert con.read_arrow(paths).count() == 12 * 5000
    a, s = con.read_arrow(paths).fetch_columns()
    assert a == list(range(5000)) * 12 and s[-1] == "value 4999"
    # early close: pull two chunks, drop the relation (the producer is blocked on a full queue or a lease)
    rel = con.read_arrow(paths)
    it = rel.chunks()
    next(it)
    next(it)
    rel.close()
    # a broken file in the middle: rows of the files before it arrive, then the error
    bad = str(tmp_path / "f05.arrows")
    with open(bad, "r+b") as f:
        f.seek(0)
        f.write(b"\x01\x02\x03\x04")
    rel = con.read_arrow(paths)
    seen = 0
    with pytest.raises(da.MiError):
        for ch in rel.chunks():
            seen += ch.size
    assert seen >= 4 * 5000 and seen <= 5 * 5000


def test_empty_record_batches_with_nested_columns(con, tmp_path):
    """Regression (found by the random-table test): a record batch of 0 rows whose columns are lists of lists / string
    views gets no transcode task, so nothing may refer to one (window / buffer tables were patched into a missing task)."""
    sch = pa.schema([("ll", pa.large_list(pa.large_list(pa.int16()))), ("k", pa.int64()), ("sv", pa.string_view()),
                     ("l", pa.list_(pa.string()))])
    def batch(rows):
        return pa.record_batch([pa.array([r[0] for r in rows], sch[0].type), pa.array([r[1] for r in rows], sch[1].type),
                                pa.array([r[2] for r in rows], sch[2].type), pa.array([r[3] for r in rows], sch[3].type)], schema=sch)
    rows = [([[1, 2], [], None, [3]], 7, "a string longer than twelve bytes", ["x", None]), (None, None, None, None), ([], 1, "", [])]
    path = str(tmp_path / "e.arrows")
    with ipc.new_stream(path, sch) as w:
        w.write_batch(batch([]))
        w.write_batch(batch(rows))
        w.write_batch(batch([]))
        w.write_batch(batch(rows * 1000))
    got = con.read_arrow(path).fetch_columns()
    want = ipc.open_stream(path).read_all()
    assert [len(c) for c in got] == [3003] * 4
    for name, g in zip(want.column_names, got):
        assert g == want.column(name).to_pylist(), name


# ---------------------------------------------------------------------------------------- device-resident chunks, every shape
def _mirror_device_vector(hip, v, ty, n, keep):
    """Deep copy of a device-resident mi_vector tree into host memory (pointers rewritten), so the ordinary host
    conversion can read it: data / validity arrays by D2H, long-string payloads fetched one by one through their
    device pointers."""
    import ctypes as C

    def d2h(ptr, nbytes):
        buf = np.zeros(max(nbytes, 1), np.uint8)
        if nbytes:
            assert hip.hipMemcpy(C.c_void_p(buf.ctypes.data), C.c_void_p(ptr), C.c_size_t(nbytes), 2) == 0
        keep.append(buf)
        return buf

    out = _ffi.Vector()
    out.kind, out.out_width, out.count, out.validity_shift = v.kind, v.out_width, v.count, v.validity_shift
    if v.validity:
        out.validity = d2h(v.validity, ((n + v.validity_shift + 63) // 64) * 8).ctypes.data
    if ty[0] == "leaf":
        if v.kind == _ffi.K_DICT:
            out.data = d2h(v.data, 4 * n).ctypes.data
            dn = v.dict_len + 1
            w = 16 if ty[1] in ("VARCHAR", "BLOB") else da._dict_width(ty[1])
            dd = d2h(v.dictionary, dn * w)
            if w == 16:
                _fix_strings(hip, dd, dn, keep)
            out.dictionary, out.dict_len = dd.ctypes.data, v.dict_len
            out.dictionary_validity = d2h(v.dictionary_validity, ((dn + 63) // 64) * 8).ctypes.data
            return out
        data = d2h(v.data, n * v.out_width)
        if ty[1] in ("VARCHAR", "BLOB"):
            _fix_strings(hip, data, n, keep, valid=(out.validity, v.validity_shift))
        out.data = data.ctypes.data
        return out
    if ty[0] in ("list", "map"):
        out.data = d2h(v.data, 16 * n).ctypes.data
        cty = ty[1] if ty[0] == "list" else ("struct", [("key", ty[1]), ("value", ty[2])])
        kids = (_ffi.Vector * 1)()
        kids[0] = _mirror_device_vector(hip, v.children[0], cty, v.children[0].count, keep)
    elif ty[0] == "array":
        kids = (_ffi.Vector * 1)()
        kids[0] = _mirror_device_vector(hip, v.children[0], ty[1], v.children[0].count, keep)
    else:
        kids = (_ffi.Vector * len(ty[1]))()
        for k, (_, kt) in enumerate(ty[1]):
            kids[k] = _mirror_device_vector(hip, v.children[k], kt, n, keep)
    keep.append(kids)
    out.children, out.n_children = kids, len(kids)
    return out


def _fix_strings(hip, data, n, keep, valid=None):
    import ctypes as C
    s = data.reshape(-1, 16)
    lens = s[:, :4].copy().view(np.uint32).reshape(-1)
    ptrs = s[:, 8:].copy().view(np.uint64).reshape(-1)
    ok = np.ones(n, bool)
    if valid is not None and valid[0]:
        words = np.ctypeslib.as_array(C.cast(valid[0], C.POINTER(C.c_uint64)), shape=((n + valid[1] + 63) // 64,))
        ok = np.unpackbits(words.view(np.uint8), bitorder="little")[valid[1]: valid[1] + n].astype(bool)
    for i in range(n):
        if ok[i] and lens[i] > 12:
            payload = np.zeros(int(lens[i]), np.uint8)
            assert hip.hipMemcpy(C.c_void_p(payload.ctypes.data), C.c_void_p(int(ptrs[i])), C.c_size_t(int(lens[i])), 2) == 0
            keep.append(payload)
            s[i, 8:] = np.frombuffer(np.uint64(payload.ctypes.data).tobytes(), np.uint8)


@pytest.mark.parametrize("zero_copy", [False, True])
@pytest.mark.parametrize("rel_path", ["edge_nested.arrows", "lineitem_sf0_01_head.arrows", "edge_dict.arrows", "edge_types.arrows"])
def test_device_resident_chunks_of_every_shape(con, golden_dir, rel_path, zero_copy):
    """device_resident = 1 for a GPU consumer: every chunk's vector tree (strings with payload pointers into the body in
    HBM, list entries + child vectors, struct children, dictionaries, zero-copy aliases of the body) copied back through
    its device pointers equals the host-consumer scan of the same file."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    path = g(golden_dir, rel_path)
    want = con.read_arrow(path, accept_dictionaries=True).fetch_columns()
    rel = con.read_arrow(path, accept_dictionaries=True, device_resident=True, zero_copy_direct=zero_copy)
    types = [da.parse_duck_type(t) for t in rel.types]
    got = [[] for _ in types]
    for ch in rel.chunks():
        keep = []
        for ci, ty in enumerate(types):
            hv = _mirror_device_vector(hip, ch.columns[ci], ty, ch.size, keep)
            got[ci].extend(da._vector_values(hv, ty, ch.size))
    assert [canon_python(c) for c in got] == [canon_python(c) for c in want]


def test_context_reports_the_gpus_numa_node(con):
    """mi_ctx_numa: the node of the GPU and its CPUs as /sys names them (-1 and nothing on a platform that does not say); the
    library's own host threads run there (Context::BindThisThread), the caller's thread only when it asks."""
    import os
    node, cpus = con.ctx.numa()
    assert node >= -1
    if node < 0:
        assert not cpus
        return
    assert cpus and all(0 <= c < 4096 for c in cpus)
    with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
        want = set()
        for part in f.read().strip().split(","):
            lo, _, hi = part.partition("-")
            want.update(range(int(lo), int(hi or lo) + 1))
    assert cpus <= want      # the device's local CPUs are CPUs of that node
    before = os.sched_getaffinity(0)
    try:
        assert con.ctx.bind_this_thread() == (node if cpus & before else -1)
        assert os.sched_getaffinity(0) == ((cpus & before) or before)
    finally:
        os.sched_setaffinity(0, before)
