"""Host half of the product path (IPC framing, flatbuffer metadata, projection, index, encoder) -- CPU only.
Checked against pyarrow (the reference's own test oracle) and against the CPU oracle's independent parser."""
import os

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pytest

import duckdb_arrow_amd as da
from duckdb_arrow_amd import _ffi
from oracle import pyoracle as po

from test_oracle_golden import STREAM_FILES, load


@pytest.mark.parametrize("rel", STREAM_FILES)
def test_reader_matches_oracle_and_pyarrow_metadata(golden_dir, rel):
    buf = load(golden_dir, rel)
    rd = da.Reader(buffers=[buf])
    fields = rd.schema()
    ofields, _, _ = po.decode_schema(buf[po.walk_stream(buf)[0]["meta_off"]:][: po.walk_stream(buf)[0]["meta_len"]])
    top, i = [], 0   # the oracle lists fields depth first; keep the top-level ones
    def skip(i):
        k = ofields[i]["n_children"]
        i += 1
        for _ in range(k):
            i = skip(i)
        return i
    while i < len(ofields):
        top.append(ofields[i]["name"])
        i = skip(i)
    assert [f["name"] for f in fields] == top
    try:
        t = ipc.open_stream(pa.py_buffer(buf)).read_all()
    except pa.ArrowInvalid:
        t = ipc.open_file(pa.py_buffer(buf)).read_all()
    assert [f["name"] for f in fields] == t.column_names
    msgs = [m for m in po.walk_stream(buf) if m["type"] != po.MSG_SCHEMA]
    n = 0
    for m in msgs:
        b = rd.next_batch(accept_dictionaries=True)
        assert b is not None
        rb = po.decode_record_batch(buf[m["meta_off"]: m["meta_off"] + m["meta_len"]])
        assert b["length"] == rb["length"] and b["body_file_offset"] == m["body_off"] and b["body_size"] == m["body_len"]
        assert bool(b["is_dictionary"]) == (m["type"] == po.MSG_DICTIONARY_BATCH)
        if not b["is_dictionary"]:
            n += b["length"]
            # every span the product slices equals the oracle's: field nodes depth first, each node's buffers in order
            assert len(b["nodes"]) == len(rb["nodes"])
            k = 0
            for ni, nd in enumerate(b["nodes"]):
                nb = len(nd["spans"])
                assert [tuple(x) for x in nd["spans"]] == [tuple(x) for x in rb["buffers"][k: k + nb]], (ni, nd["name"])
                assert (nd["length"], nd["null_count"]) == tuple(rb["nodes"][ni][:2])
                k += nb
            assert k == len(rb["buffers"])
    assert rd.next_batch(accept_dictionaries=True) is None
    assert n == t.num_rows


def test_duck_types_like_typeof(golden_dir):
    """multifile_reading.test:96-114 reads typeof(): VARCHAR / DOUBLE / BIGINT for the fruit files."""
    types = lambda rel: {f["name"]: f["duck_type"] for f in da.Reader(path=os.path.join(golden_dir, rel)).schema()}
    assert types("ref_data/multifile/glob/f1.arrow") == {"fruit": "VARCHAR", "variety": "VARCHAR", "weight": "DOUBLE"}
    assert types("ref_data/multifile/different_type.arrows")["weight"] == "VARCHAR"
    assert types("ref_data/multifile/different_type_int.arrows")["weight"] == "BIGINT"
    assert types("ref_data/test.arrows") == {"commit": "VARCHAR", "time": "TIMESTAMP WITH TIME ZONE", "files": "INTEGER",
                                             "merge": "BOOLEAN", "message": "VARCHAR"}
    t = types("edge_types.arrows")
    assert t["dec4"] == "DECIMAL(4,1)" and t["dec38"] == "DECIMAL(38,5)" and t["ts_s"] == "TIMESTAMP_S"
    assert t["tz_ns"] == "TIMESTAMP WITH TIME ZONE" and t["t32ms"] == "TIME" and t["fsb"] == "BLOB" and t["u16"] == "USMALLINT"


def test_file_reader_equals_buffer_reader(golden_dir):
    p = os.path.join(golden_dir, "ref_data/test.arrows")
    a, b = da.Reader(path=p), da.Reader(buffers=[load(golden_dir, "ref_data/test.arrows")])
    assert a.schema() == b.schema()
    while True:
        x, y = a.next_batch(), b.next_batch()
        assert (x is None) == (y is None)
        if x is None:
            break
        assert x["length"] == y["length"] and x["buffers"] == y["buffers"] and (x["body"] == y["body"]).all()
    assert a.progress() == pytest.approx(100.0, abs=0.01)


def test_ipc_file_format_magic_is_skipped(golden_dir):
    """ipc_file_stream_reader.cpp:107-119; read_arrow_file.test:9-17 (data/fruit.arrow has 6 rows)."""
    rd = da.Reader(path=os.path.join(golden_dir, "ref_data/fruit.arrow"))
    assert [f["name"] for f in rd.schema()] == ["fruit", "variety", "weight"]
    rows = 0
    while True:
        b = rd.next_batch()
        if b is None:
            break
        rows += b["length"]
    assert rows == 6


def test_projection_and_its_errors(golden_dir):
    """IPCStreamReader::SetColumnProjection (base_stream_reader.cpp:146-212)"""
    buf = load(golden_dir, "ref_data/test.arrows")
    rd = da.Reader(buffers=[buf])
    rd.set_projection(["message", "files"])
    b = rd.next_batch()
    assert b["column_field"] == [4, 2]
    full = da.Reader(buffers=[buf]).next_batch()
    assert b["buffers"][0:3] == full["buffers"][12:15] and b["buffers"][3:6] == full["buffers"][6:9]
    with pytest.raises(da.MiError, match="Can't request zero fields projected from IpcStreamReader"):
        da.Reader(buffers=[buf]).set_projection([])
    with pytest.raises(da.MiError, match="Field 'nope' does not exist in IPC file schema"):
        da.Reader(buffers=[buf]).set_projection(["nope"])


def test_duplicate_column_names_are_deduplicated():
    """QueryResult::DeduplicateColumns as used at base_stream_reader.cpp:177: a, a -> a, a_1"""
    b = pa.record_batch([pa.array([1]), pa.array([2]), pa.array([3])], names=["a", "A", "a"])
    sink = pa.BufferOutputStream()
    with ipc.new_stream(sink, b.schema) as w:
        w.write_batch(b)
    buf = sink.getvalue().to_pybytes()
    rd = da.Reader(buffers=[buf])
    rd.set_projection(["A_1", "a_2"])
    assert rd.next_batch()["column_field"] == [1, 2]


def test_framing_errors_match_the_reference(golden_dir):
    bad = np.zeros(64, np.uint8)
    bad[:4] = [1, 2, 3, 4]
    with pytest.raises(da.MiError, match=r"Expected continuation token \(0xFFFFFFFF\) but got 67305985") as e:
        da.Reader(buffers=[bad]).schema()
    assert e.value.code == _ffi.MI_EIO
    neg = np.zeros(64, np.uint8)
    neg[:4] = 0xFF
    neg[4:8] = np.frombuffer(np.int32(-5).tobytes(), np.uint8)
    with pytest.raises(da.MiError, match="Expected metadata size >= 0 but got -5"):
        da.Reader(buffers=[neg]).schema()
    # a stream whose first message is not a Schema (base_stream_reader.cpp:238-269)
    buf = load(golden_dir, "ref_data/test.arrows")
    msgs = po.walk_stream(buf)
    with pytest.raises(da.MiError, match="Expected Schema Arrow IPC message but got RecordBatch"):
        da.Reader(buffers=[buf[msgs[1]["prefix_off"]:]]).schema()
    # empty input: "Expected Schema Arrow IPC message but got end of stream"
    with pytest.raises(da.MiError, match="Expected Schema Arrow IPC message but got end of stream"):
        da.Reader(buffers=[]).schema()
    # a second Schema message where a RecordBatch is expected
    two = np.concatenate([buf[: msgs[1]["prefix_off"]], buf[: msgs[1]["prefix_off"]]])
    rd = da.Reader(buffers=[two])
    rd.schema()
    with pytest.raises(da.MiError, match="Expected RecordBatch Arrow IPC message but got Schema"):
        rd.next_batch()


def test_truncated_file_ends_the_stream(golden_dir, tmp_path):
    """ipc_file_stream_reader.cpp:126-129: a file that ends where a prefix should start is end-of-stream; a file cut
    inside a message is an error (BufferedFileReader throws while DecodeMessage runs outside the try block)."""
    buf = load(golden_dir, "ref_data/test.arrows")
    msgs = po.walk_stream(buf)
    p = tmp_path / "no_eos.arrows"
    p.write_bytes(buf[: msgs[3]["body_off"] + msgs[3]["body_len"]].tobytes())
    rd = da.Reader(path=str(p))
    n = 0
    while rd.next_batch() is not None:
        n += 1
    assert n == 3
    p2 = tmp_path / "cut.arrows"
    p2.write_bytes(buf[: msgs[3]["body_off"] + 100].tobytes())
    rd = da.Reader(path=str(p2))
    assert rd.next_batch() is not None and rd.next_batch() is not None
    with pytest.raises(da.MiError):
        rd.next_batch()


def test_batch_index_for_sharding(golden_dir):
    buf = load(golden_dir, "ref_data/test.arrows")
    idx = da.Reader(buffers=[buf]).index()
    msgs = [m for m in po.walk_stream(buf) if m["type"] == po.MSG_RECORD_BATCH]
    assert [(e["prefix_offset"], e["meta_len"], e["body_offset"], e["body_len"]) for e in idx] == \
        [(m["prefix_off"], m["meta_len"], m["body_off"], m["body_len"]) for m in msgs]
    assert sum(e["n_rows"] for e in idx) == 15487
    idx2 = da.Reader(path=os.path.join(golden_dir, "ref_data/test.arrows")).index()
    assert idx2 == idx


def test_file_footer_index_equals_header_walk(golden_dir, tmp_path):
    """IPC *file* format: the footer's Block list gives the batch index without walking the stream
    (noted as future work in the reference: ipc_file_stream_reader.cpp:113-115, arrow_file_scan.cpp:36-40)."""
    for rel in ("edge_file_format.arrow", "ref_data/fruit.arrow"):
        p = os.path.join(golden_dir, rel)
        from_footer = da.Reader(path=p).index()
        walked = da.Reader(buffers=[load(golden_dir, rel)]).index()  # the buffer reader always walks headers
        assert from_footer == walked and len(from_footer) >= 1
    # a file with a dictionary: dictionary blocks come first, in stream order
    t = pa.table({"d": pa.array(["x", "y", "x", None] * 50).dictionary_encode(), "v": list(range(200))})
    p = str(tmp_path / "dict.arrow")
    with ipc.new_file(p, t.schema) as w:
        w.write_table(t, max_chunksize=64)
    idx = da.Reader(path=p).index()
    assert [e["type"] for e in idx] == [2, 3, 3, 3, 3] and sum(e["n_rows"] for e in idx if e["type"] == 3) == 200


def test_size_validation_rejects_short_buffers(golden_dir):
    """NANOARROW_VALIDATION_LEVEL_FULL size checks: a RecordBatch whose body is shorter than its buffers claim."""
    buf = load(golden_dir, "ref_data/test.arrows").copy()
    msgs = po.walk_stream(buf)
    # shrink bodyLength of the first record batch by rewriting the stream with a truncated body
    m = msgs[1]
    cut = np.concatenate([buf[: m["body_off"] + 1000], buf[m["body_off"] + m["body_len"]:]])
    rd = da.Reader(buffers=[cut])
    with pytest.raises(da.MiError):
        while rd.next_batch() is not None:
            pass


def _stream(table, **opts):
    sink = pa.BufferOutputStream()
    with ipc.new_stream(sink, table.schema, options=ipc.IpcWriteOptions(**opts)) as w:
        for b in table.to_batches(max_chunksize=3000):
            w.write_batch(b)
    return sink.getvalue().to_pybytes()


def test_zstd_bodies_are_decompressed_like_the_reference():
    """base_stream_reader.cpp:11-50: per-buffer ZSTD (int64 uncompressed length, -1 = stored raw); the decompressed
    buffers equal the buffers of the same table written uncompressed."""
    rng = np.random.default_rng(2)
    t = pa.table({"a": rng.integers(0, 50, 7000), "s": ["row %d" % (i % 97) for i in range(7000)],
                  "n": pa.array([None if i % 5 == 0 else float(i) for i in range(7000)]),
                  "tiny": pa.array([1] * 7000, pa.int8())})
    plain, packed = _stream(t), _stream(t, compression="zstd")
    assert len(packed) < len(plain)
    ra, rb = da.Reader(buffers=[plain]), da.Reader(buffers=[packed])
    assert ra.schema() == rb.schema()
    while True:
        x, y = ra.next_batch(), rb.next_batch()
        assert (x is None) == (y is None)
        if x is None:
            break
        assert x["length"] == y["length"] and y["compression"] == -1
        for (xo, xl), (yo, yl) in zip(x["buffers"], y["buffers"]):
            assert xl == yl and (x["body"][xo: xo + xl] == y["body"][yo: yo + yl]).all() and yo % 64 == 0


def test_lz4_frame_bodies_are_decompressed():
    """Beyond the reference (it registers no LZ4 function, base_stream_reader.cpp:37-50, and rejects these bodies):
    LZ4_FRAME buffers -- Feather V2's default -- decode to the buffers of the uncompressed stream."""
    rng = np.random.default_rng(3)
    t = pa.table({"a": rng.integers(0, 50, 90000), "s": ["row %d" % (i % 97) for i in range(90000)],
                  "n": pa.array([None if i % 5 == 0 else float(i) for i in range(90000)])})
    plain, packed = _stream(t), _stream(t, compression="lz4")
    assert len(packed) < len(plain)
    ra, rb = da.Reader(buffers=[plain]), da.Reader(buffers=[packed])
    n = 0
    while True:
        x, y = ra.next_batch(), rb.next_batch()
        assert (x is None) == (y is None)
        if x is None:
            break
        n += x["length"]
        for (xo, xl), (yo, yl) in zip(x["buffers"], y["buffers"]):
            assert xl == yl and (x["body"][xo: xo + xl] == y["body"][yo: yo + yl]).all()
    assert n == 90000
    # a damaged frame is an IO error, not a crash
    buf = bytearray(packed)
    m = po.walk_stream(np.frombuffer(bytes(buf), np.uint8))[1]
    buf[m["body_off"] + 8: m["body_off"] + 12] = b"\x00\x00\x00\x00"      # frame magic of the first buffer
    with pytest.raises(da.MiError, match="LZ4F_decompress") as e:
        da.Reader(buffers=[bytes(buf)]).next_batch()
    assert e.value.code == _ffi.MI_EIO


@pytest.mark.parametrize("codec", ["lz4", "zstd"])
def test_raw_buffers_with_length_prefix_minus_one(codec):
    """A compressed body may store single buffers raw (length prefix -1: Arrow C++ with min_space_savings, arrow-rs, Arrow
    Java); nanoarrow copies them, so the reference reads such files (base_stream_reader.cpp:11-32 sees only real frames).
    pyarrow's writer never emits them: helpers.rewrite_buffers_raw makes the stream, pyarrow reads it back as the table."""
    from helpers import rewrite_buffers_raw
    rng = np.random.default_rng(4)
    t = pa.table({"a": rng.integers(0, 1 << 60, 30000), "s": ["row %d" % (i % 97) for i in range(30000)],
                  "n": pa.array([None if i % 5 == 0 else float(i) for i in range(30000)])})
    plain = _stream(t)
    packed = rewrite_buffers_raw(_stream(t, compression=codec), lambda bi, k, ln: k % 3 != 2)
    assert ipc.open_stream(pa.py_buffer(packed)).read_all().equals(t)
    ra, rb = da.Reader(buffers=[plain]), da.Reader(buffers=[packed])
    n = 0
    while True:
        x, y = ra.next_batch(), rb.next_batch()
        assert (x is None) == (y is None)
        if x is None:
            break
        n += x["length"]
        for (xo, xl), (yo, yl) in zip(x["buffers"], y["buffers"]):
            assert xl == yl and (x["body"][xo: xo + xl] == y["body"][yo: yo + yl]).all()
    assert n == 30000


def test_corrupt_zstd_frame_is_an_io_error():
    t = pa.table({"a": list(range(5000))})
    buf = bytearray(_stream(t, compression="zstd"))
    msgs = po.walk_stream(np.frombuffer(bytes(buf), np.uint8))
    m = msgs[1]
    buf[m["body_off"] + 12: m["body_off"] + 20] = b"\xff" * 8
    rd = da.Reader(buffers=[bytes(buf)])
    with pytest.raises(da.MiError, match="ZSTD_decompress") as e:
        rd.next_batch()
    assert e.value.code == _ffi.MI_EIO


def test_synthetic_lineitem_stream_is_valid_arrow():
    """The generator goes through the product's own flatbuffer encoder; pyarrow must accept it (its verifier runs on
    every message) and the data must be TPC-H shaped and reproducible."""
    buf, info = da.synth_lineitem_stream(scale_factor=0.01, seed=7, rows_per_batch=20000)
    buf2, _ = da.synth_lineitem_stream(scale_factor=0.01, seed=7, rows_per_batch=20000, n_threads=1)
    assert (buf == buf2).all()
    t = ipc.open_stream(pa.py_buffer(buf)).read_all()
    assert t.num_rows == info["n_rows"] == 60012 and info["n_batches"] == 4
    assert t.schema.field("l_extendedprice").type == pa.decimal128(15, 2) and t.schema.field("l_shipdate").type == pa.date32()
    ship = np.array(t["l_shipdate"].cast(pa.int32()))
    assert ship.min() >= 8036 and ship.max() <= 10561
    assert set(t["l_returnflag"].to_pylist()) <= {"A", "N", "R"} and set(t["l_linestatus"].to_pylist()) == {"F", "O"}
    ok = np.array(t["l_orderkey"])
    assert (np.diff(ok) >= 0).all()
    # DuckDB-writer style: validity bitmaps present, ~174.85 B/row
    assert 172 < info["stream_size"] / info["n_rows"] < 178
    nov, info2 = da.synth_lineitem_stream(scale_factor=0.01, seed=7, rows_per_batch=20000, with_validity=False)
    assert ipc.open_stream(pa.py_buffer(nov)).read_all().equals(t)
    assert info2["stream_size"] < info["stream_size"]


@pytest.mark.parametrize("compression", [None, "zstd"])
def test_projected_file_read_touches_only_the_projected_buffers(tmp_path, compression):
    """Projection pushdown reaches the file (the reference reads whole bodies, ipc_file_stream_reader.cpp:71-89): with a
    projection the file reader preads / decompresses only the buffers of the projected columns; what it returns for them
    equals the unprojected read byte for byte, nested columns and string views included."""
    rng = np.random.default_rng(8)
    n = 30000
    t = pa.table({
        "k": rng.integers(0, 10**9, n),
        "s": pa.array(["string number %d" % i for i in range(n)]),
        "l": pa.array([[int(x) for x in rng.integers(0, 9, int(rng.integers(0, 4)))] for _ in range(n)], pa.list_(pa.int32())),
        "d": pa.array(rng.integers(8000, 11000, n).astype(np.int32), pa.int32()),
        "v": pa.array(["view %d" % (i % 13) * (1 + i % 3) for i in range(n)], pa.string_view()),
        "st": pa.array([{"a": int(i), "b": "x" * (i % 20)} for i in range(n)], pa.struct([("a", pa.int64()), ("b", pa.string())])),
    })
    path = str(tmp_path / "t.arrows")
    opts = ipc.IpcWriteOptions(compression=compression) if compression else None
    with ipc.new_stream(path, t.schema, options=opts) as w:
        w.write_table(t, max_chunksize=7000)
    full = da.Reader(path=path)
    part = da.Reader(path=path)
    part.set_projection(["st", "d", "l"])
    nb = 0
    while True:
        x, y = full.next_batch(), part.next_batch()
        assert (x is None) == (y is None)
        if x is None:
            break
        nb += 1
        by_name = {}
        for i, nd in enumerate(x["nodes"]):
            if nd["depth"] == 0:
                top = nd["name"]
            by_name.setdefault(top, []).append(nd)
        got = {}
        for nd in y["nodes"]:
            if nd["depth"] == 0:
                top = nd["name"]
            got.setdefault(top, []).append(nd)
        assert sorted(got) == ["d", "l", "st"]          # nodes are listed in file order,
        assert [y["nodes"][i]["name"] for i in y["column_node"]] == ["st", "d", "l"]   # columns in projection order
        for name, nodes in got.items():
            for a, b in zip(by_name[name], nodes):
                assert (a["length"], a["null_count"], len(a["spans"])) == (b["length"], b["null_count"], len(b["spans"]))
                for (ao, al), (bo, bl) in zip(a["spans"], b["spans"]):
                    assert al == bl and np.array_equal(x["body"][ao: ao + al], y["body"][bo: bo + bl]), (name, a["name"])
    assert nb == 5 and part.next_batch() is None


def test_writer_schema_of_nested_duck_types():
    """mi_encode_schema (host only): nested DuckDB type strings become the Arrow types ArrowConverter::ToArrowSchema
    exports -- list child "l", fixed_size_list, struct field names, map<entries: struct<key not null, value>> -- and
    pyarrow's flatbuffer verifier accepts the message; malformed type strings are rejected."""
    names = ["l", "ll", "arr", "st", "mp", "lst", "d", "ts"]
    types = ["INTEGER[]", "VARCHAR[][]", "SMALLINT[3]", 'STRUCT(a BIGINT, "b c" VARCHAR)', "MAP(VARCHAR, DECIMAL(10,2))",
             "STRUCT(x DOUBLE, y INTEGER[])[]", "DECIMAL(15,2)", "TIMESTAMP WITH TIME ZONE"]
    sch = ipc.read_schema(pa.py_buffer(da.encode_schema(names, types)))
    assert sch.names == names
    assert sch.field("l").type == pa.list_(pa.field("l", pa.int32()))
    assert sch.field("ll").type == pa.list_(pa.field("l", pa.list_(pa.field("l", pa.string()))))
    assert sch.field("arr").type == pa.list_(pa.field("l", pa.int16()), 3)
    assert sch.field("st").type == pa.struct([("a", pa.int64()), ("b c", pa.string())])
    assert sch.field("mp").type == pa.map_(pa.string(), pa.decimal128(10, 2))
    assert sch.field("lst").type == pa.list_(pa.field("l", pa.struct([("x", pa.float64()), ("y", pa.list_(pa.field("l", pa.int32())))])))
    assert sch.field("d").type == pa.decimal128(15, 2) and sch.field("ts").type == pa.timestamp("us", "UTC")
    assert all(sch.field(n).nullable for n in names)
    for bad in ("STRUCT()", "MAP(VARCHAR)", "INTEGER[0]", "STRUCT(a)", "NOTATYPE[]"):
        with pytest.raises(da.MiError):
            da.encode_schema(["c"], [bad])
