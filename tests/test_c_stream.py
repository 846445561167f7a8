"""The reader exported as an Arrow C stream (mi_reader_export_stream) -- the reference's IpcArrayStream seam
(src/ipc/array_stream.cpp:11-26).  pyarrow imports the stream through the C data interface and must see exactly the
table its own IPC reader sees: zero-copy buffers, nested children, dictionaries, views, decompressed bodies, projections,
schema / field metadata, and the reference's error behaviour (get_next -> EIO + message).  CPU only."""
import os

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pytest

import duckdb_arrow_amd as da

from test_oracle_golden import STREAM_FILES


def tables_equal(a, b):
    """Table.equals, except that NaN equals NaN in floating-point columns"""
    if a.schema != b.schema or a.num_rows != b.num_rows:
        return False
    for name in a.column_names:
        x, y = a.column(name), b.column(name)
        if pa.types.is_floating(x.type):
            canon = lambda col: ["nan" if (v is not None and v != v) else v for v in col.to_pylist()]
            if canon(x) != canon(y):
                return False
        elif not x.equals(y):
            return False
    return True


def read_any(path):
    try:
        return ipc.open_stream(path).read_all()
    except pa.ArrowInvalid:
        return ipc.open_file(path).read_all()


@pytest.mark.parametrize("rel", STREAM_FILES)
def test_exported_stream_equals_pyarrow(golden_dir, rel):
    path = os.path.join(golden_dir, rel)
    want = read_any(path)
    got = da.Reader(path=path).export_stream(accept_dictionaries=True).read_all()
    assert got.schema.equals(want.schema, check_metadata=True), (got.schema, want.schema)
    assert tables_equal(got, want)
    got.validate(full=True)
    # and over caller-owned buffers (scan_arrow_ipc's source)
    buf = np.fromfile(path, np.uint8)
    assert tables_equal(da.Reader(buffers=[buf]).export_stream(accept_dictionaries=True).read_all(), want)


def test_exported_stream_projection_compression_and_batches(tmp_path):
    rng = np.random.default_rng(3)
    n = 20000
    t = pa.table({
        "k": rng.integers(0, 10**9, n),
        "s": pa.array([None if i % 7 == 0 else "string number %d" % i for i in range(n)]),
        "l": pa.array([[int(x) for x in rng.integers(0, 9, int(rng.integers(0, 4)))] for _ in range(n)], pa.list_(pa.int32())),
        "v": pa.array(["view %d" % (i % 13) * (1 + i % 3) for i in range(n)], pa.string_view()),
        "st": pa.array([{"a": int(i), "b": "x" * (i % 20)} for i in range(n)], pa.struct([("a", pa.int64()), ("b", pa.string())])),
        "d": pa.array(["cat%d" % (i % 5) for i in range(n)]).dictionary_encode(),
    })
    sch = t.schema.with_metadata({"origin": "test", "blob": b"\\x00\\x01"})
    sch = sch.set(0, sch.field(0).with_metadata({"unit": "id"}))
    t = t.cast(sch)
    for codec in (None, "zstd", "lz4"):
        path = str(tmp_path / ("t_%s.arrows" % codec))
        with ipc.new_stream(path, t.schema, options=ipc.IpcWriteOptions(compression=codec) if codec else None) as w:
            w.write_table(t, max_chunksize=6000)
        rd = da.Reader(path=path)
        rd.set_projection(["st", "k", "v", "d"])
        stream = rd.export_stream(accept_dictionaries=True)
        batches = list(stream)
        assert [b.num_rows for b in batches] == [6000, 6000, 6000, 2000]
        got = pa.Table.from_batches(batches)
        assert got.equals(t.select(["st", "k", "v", "d"]))
        assert got.schema.metadata == t.schema.metadata and got.schema.field("k").metadata == {b"unit": b"id"}


def test_exported_stream_errors_like_the_reference(golden_dir, tmp_path):
    """A stream that breaks off inside a message: the batches before it arrive, then get_next fails with the reader's
    message (IpcArrayStream::Wrap maps IOException to EIO + last_msg, array_stream.hpp:29-48)."""
    src = np.fromfile(os.path.join(golden_dir, "ref_data/test.arrows"), np.uint8)
    cut = str(tmp_path / "cut.arrows")
    src[: src.size // 2].tofile(cut)
    stream = da.Reader(path=cut).export_stream()
    seen = 0
    with pytest.raises(Exception, match="not enough data in file"):
        for b in stream:
            seen += b.num_rows
    assert 0 < seen < 15487
    # dictionary-encoded input without accept_dictionaries: the reference's "Expected RecordBatch ... but got DictionaryBatch"
    with pytest.raises(Exception, match="Expected RecordBatch Arrow IPC message but got DictionaryBatch"):
        da.Reader(path=os.path.join(golden_dir, "edge_dict.arrows")).export_stream().read_all()
    # an exported reader only accepts close
    rd = da.Reader(path=os.path.join(golden_dir, "ref_data/test.arrows"))
    rd.export_stream().read_all()
    with pytest.raises(da.MiError):
        rd.export_stream()
    with pytest.raises(da.MiError, match="only mi_reader_close is valid"):
        rd.schema()
    with pytest.raises(da.MiError, match="only mi_reader_close is valid"):
        rd.next_batch()
    rd.close()


@pytest.mark.parametrize("rel", ["edge_nested.arrows", "ref_data/test.arrows", "edge_dict.arrows"])
def test_exported_stream_of_damaged_bodies_is_rejected_or_valid(golden_dir, rel):
    """A consumer of the C stream trusts offsets, views and dictionary indices (DuckDB does), so the export walks them like
    nanoarrow's FULL validation: a damaged body either fails in get_next or imports as arrays pyarrow's own full validation
    accepts -- never as arrays that point outside their buffers."""
    from oracle import pyoracle as po
    src = np.fromfile(os.path.join(golden_dir, rel), np.uint8)
    bodies = [(m["body_off"], m["body_len"]) for m in po.walk_stream(src) if m["body_len"] > 0]
    rng = np.random.default_rng(abs(hash(rel)) % 2**32)
    rejected = accepted = 0
    for it in range(int(os.environ.get("MI_FUZZ_ITERS", "120"))):
        buf = src.copy()
        for _ in range(int(rng.integers(1, 4))):
            off, ln = bodies[int(rng.integers(0, len(bodies)))]
            p = off + int(rng.integers(0, ln))
            if it % 3 == 0:
                p = p // 8 * 8
                buf[p: p + 8] = np.frombuffer(np.int64(rng.choice([-1, 2**31 - 1, 2**40, -2**31, 2**62])).tobytes(), np.uint8)
            else:
                buf[p] = int(rng.integers(0, 256))
        try:
            t = da.Reader(buffers=[buf]).export_stream(accept_dictionaries=True).read_all()
        except Exception:  # noqa: BLE001  (pyarrow raises OSError / ArrowInvalid with the stream's message)
            rejected += 1
            continue
        try:
            t.validate(full=True)
        except pa.ArrowInvalid as e:
            # content-level complaints (a flipped payload byte is not UTF-8 any more, an inline view's pad bytes are not
            # zero) are fine; anything about offsets, sizes or indices would mean the export let a bad pointer through
            assert any(k in str(e) for k in ("UTF8", "utf8", "UTF-8", "padding bytes", "inlined prefix", "null_count", "Null count")), str(e)
        accepted += 1
    assert rejected + accepted > 0
