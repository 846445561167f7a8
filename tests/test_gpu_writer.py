"""The encode direction on the GPU: K7 kernels vs the oracle, COPY ... (FORMAT ARROWS) and to_arrow_ipc end to end.
Written after test/sql/write_arrow_stream.test, test/sql/test_copy_to.test, test/sql/to_arrow_ipc.test and
test/python/test_arrow_ipc_writer.py; pyarrow (the reference's own oracle) reads everything the writer produces."""
import ctypes as C
import glob
import os

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pytest

import duckdb_arrow_amd as da
from duckdb_arrow_amd import _ffi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def con():
    return da.Connection(0)


@pytest.fixture(scope="module")
def torch():
    import torch
    return torch


def read_any(path):
    return ipc.open_stream(path).read_all()


# ---------------------------------------------------------------------------------------- kernel level
def test_encode_is_the_inverse_of_decode_on_lineitem(con, torch):
    """encode(decode(body)) reproduces every Arrow buffer of a DuckDB-writer-style lineitem batch bit for bit
    (validity present, decimal128 sign extension, int32 offsets + payload), i.e. the K7 kernels invert K1-K4."""
    from duckdb_arrow_amd.hbm import HbmStream
    buf, info = da.synth_lineitem_stream(scale_factor=0.05, seed=9, rows_per_batch=70000)
    hs = HbmStream(con.ctx, buf)
    hs.launch()
    assert hs.status() == 0
    torch.cuda.synchronize()
    in_base, out_base = hs.in_ptr, hs.out_ptr
    tasks, outs = [], []
    enc_kind = {_ffi.K_COPY: _ffi.K_ENC_COPY, _ffi.K_DEC128: _ffi.K_ENC_DEC128, _ffi.K_STR32: _ffi.K_ENC_STR32}
    for lay in hs.layout:
        n = lay["nrows"]
        for e in lay["columns"]:
            spans = e["buffers"]
            nb = 3 if e["kind"] == _ffi.K_STR32 else 2
            o_valid = torch.zeros((n + 7) // 8 + 16, dtype=torch.uint8, device="cuda")
            o_b1 = torch.zeros(spans[1][1] + 16, dtype=torch.uint8, device="cuda")
            o_b2 = torch.zeros(spans[2][1] + 16, dtype=torch.uint8, device="cuda") if nb == 3 else None
            outs.append((lay, e, o_valid, o_b1, o_b2))
            tasks.append(da.make_task(enc_kind[e["kind"]], n, out_base + e["data_off"], o_b1.data_ptr(),
                                      validity=out_base + e["valid_off"], out_validity=o_valid.data_ptr(),
                                      out_aux=o_b2.data_ptr() if nb == 3 else 0, buf2=in_base, ptr_base=0,
                                      buf2_len=spans[2][1] if nb == 3 else 0, param=e["param"] if e["kind"] != _ffi.K_STR32 else 0))
    # string_t pointers produced by the decode hold absolute stream positions (ptr_base = position of the data buffer),
    # so the "heap" of the encode tasks is the resident stream itself
    plan = da.Plan(con.ctx, tasks)
    plan.launch(torch.cuda.current_stream().cuda_stream)
    assert plan.status() == 0
    assert plan.null_counts() == [0] * len(tasks)
    host = buf
    for lay, e, o_valid, o_b1, o_b2 in outs:
        body = lay["body_off"]
        spans = e["buffers"]
        n = lay["nrows"]
        want_valid = host[body + spans[0][0]: body + spans[0][0] + (n + 7) // 8].copy()
        if n & 7:
            want_valid[-1] |= (0xFF << (n & 7)) & 0xFF  # ArrowAppender pads with 1s (ResizeValidity 0xFF)
        assert np.array_equal(o_valid.cpu().numpy()[: (n + 7) // 8], want_valid), e["name"]
        assert np.array_equal(o_b1.cpu().numpy()[: spans[1][1]], host[body + spans[1][0]: body + spans[1][0] + spans[1][1]]), e["name"]
        if o_b2 is not None:
            assert np.array_equal(o_b2.cpu().numpy()[: spans[2][1]], host[body + spans[2][0]: body + spans[2][0] + spans[2][1]]), e["name"]


def test_encode_kernels_match_oracle_with_nulls(con, torch):
    rng = np.random.default_rng(11)
    n = 5000
    ok = rng.random(n) < 0.8
    valid = np.packbits(np.concatenate([ok, np.ones((-n) % 64, bool)]), bitorder="little").view(np.uint64).copy()
    d_valid = torch.from_numpy(valid.view(np.uint8).copy()).cuda()

    def run(kind, src, out_bytes, param=0, aux_bytes=0, heap=None, buf2_len=0):
        d_src = torch.from_numpy(np.ascontiguousarray(src).view(np.uint8).copy()).cuda()
        d_heap = torch.from_numpy(heap.copy()).cuda() if heap is not None else None
        o_valid = torch.zeros((n + 7) // 8 + 16, dtype=torch.uint8, device="cuda")
        o_data = torch.zeros(out_bytes + 16, dtype=torch.uint8, device="cuda")
        o_aux = torch.zeros(aux_bytes + 16, dtype=torch.uint8, device="cuda")
        t = da.make_task(kind, n, d_src.data_ptr(), o_data.data_ptr(), validity=d_valid.data_ptr(), out_validity=o_valid.data_ptr(),
                         out_aux=o_aux.data_ptr(), buf2=d_heap.data_ptr() if d_heap is not None else 0, ptr_base=0, param=param,
                         buf2_len=buf2_len)
        plan = da.Plan(con.ctx, [t])
        plan.launch(torch.cuda.current_stream().cuda_stream)
        assert plan.status() == 0
        return (o_valid.cpu().numpy()[: (n + 7) // 8], o_data.cpu().numpy()[:out_bytes], o_aux.cpu().numpy()[:aux_bytes],
                plan.null_counts()[0])

    want_bitmap = np.full((n + 7) // 8, 0xFF, np.uint8)
    nulls = C.c_int64(0)
    po.lib().orc_enc_validity(valid.ctypes.data, n, 0, want_bitmap.ctypes.data, C.byref(nulls))
    # decimal widen
    for w, dt in ((2, np.int16), (4, np.int32), (8, np.int64)):
        src = rng.integers(np.iinfo(dt).min, np.iinfo(dt).max, n).astype(dt)
        bitmap, data, _, nc = run(_ffi.K_ENC_DEC128, src, 16 * n, param=w)
        want = np.zeros(16 * n, np.uint8)
        po.lib().orc_enc_decimal_widen(src.ctypes.data, w, n, want.ctypes.data)
        assert np.array_equal(data, want) and np.array_equal(bitmap, want_bitmap) and nc == nulls.value
    # bool pack
    src = (rng.random(n) < 0.5).astype(np.uint8)
    bitmap, data, _, nc = run(_ffi.K_ENC_BOOL, src, (n + 7) // 8)
    want = np.full((n + 7) // 8, 0xFF, np.uint8)
    po.lib().orc_enc_bool(src.ctypes.data, valid.ctypes.data, n, 0, want.ctypes.data)
    assert np.array_equal(data, want)
    # varchar: inline + long strings, NULLs repeat the previous offset
    lens = rng.integers(0, 40, n)
    heap = rng.integers(32, 127, int(lens.sum()) + 16, dtype=np.uint8)
    str16 = np.zeros((n, 16), np.uint8)
    pos = 0
    for i in range(n):
        ln = int(lens[i])
        str16[i, :4] = np.frombuffer(np.uint32(ln).tobytes(), np.uint8)
        if ln <= 12:
            str16[i, 4: 4 + ln] = heap[pos: pos + ln]
        else:
            str16[i, 4:8] = heap[pos: pos + 4]
            str16[i, 8:16] = np.frombuffer(np.uint64(pos).tobytes(), np.uint8)
        pos += ln
    payload = int(lens[ok].sum())
    bitmap, off, data, nc = run(_ffi.K_ENC_STR32, str16.reshape(-1), 4 * (n + 1), aux_bytes=payload, heap=heap, buf2_len=payload)
    want_off = np.zeros(n + 1, np.int32)
    want_data = np.zeros(payload + 1, np.uint8)
    rc = po.lib().orc_enc_varchar32(str16.ctypes.data, valid.ctypes.data, n, 0, 0, heap.ctypes.data, want_off.ctypes.data,
                                    want_data.ctypes.data)
    assert rc == 0 and np.array_equal(off.view(np.int32), want_off) and np.array_equal(data, want_data[:payload])
    assert nc == nulls.value


def _string_vectors(lens, heap_bytes, rng):
    """DuckDB string_t rows (n x 16 bytes) whose long payloads live in `heap` at arbitrary (unaligned) positions."""
    n = len(lens)
    heap = rng.integers(32, 127, heap_bytes, dtype=np.uint8)
    ln = np.asarray(lens, np.int64)
    starts = np.concatenate([[0], np.cumsum(ln)[:-1]])   # payload of row i = heap[starts[i] : starts[i] + ln[i]]
    str16 = np.zeros((n, 16), np.uint8)
    str16[:, :4] = ln.astype(np.uint32).view(np.uint8).reshape(n, 4)
    for i in np.nonzero(ln <= 12)[0]:
        str16[i, 4: 4 + ln[i]] = heap[starts[i]: starts[i] + ln[i]]
    big = np.nonzero(ln > 12)[0]
    for i in big:
        str16[i, 4:8] = heap[starts[i]: starts[i] + 4]
    str16[big, 8:16] = starts[big].astype(np.uint64).view(np.uint8).reshape(-1, 8)
    return str16, heap


@pytest.mark.parametrize("shape", ["long_mean_90", "one_9MiB_string", "mixed_with_empty_tiles", "all_inline"])
def test_encode_string_windows_and_huge_strings_match_oracle(con, torch, shape):
    """K7d beyond lineitem's shapes: sub-blocks whose payload exceeds one LDS window (mean 90 B), a string of 9 MiB
    (the 32-bit in-tile positions hand that sub-block to the 64-bit formulation), tiles without payload, NULLs."""
    rng = np.random.default_rng({"long_mean_90": 1, "one_9MiB_string": 2, "mixed_with_empty_tiles": 3, "all_inline": 4}[shape])
    n = 7000
    if shape == "long_mean_90":
        lens = rng.integers(0, 181, n)
    elif shape == "one_9MiB_string":
        lens = rng.integers(0, 30, n)
        lens[2500] = 9 * 2**20 + 3
    elif shape == "mixed_with_empty_tiles":
        lens = rng.integers(0, 600, n)
        lens[2048:4096] = 0
    else:
        lens = rng.integers(0, 13, n)
    ok = rng.random(n) < 0.85
    valid = np.packbits(np.concatenate([ok, np.ones((-n) % 64, bool)]), bitorder="little").view(np.uint64).copy()
    str16, heap = _string_vectors(lens, int(np.sum(lens)) + 64, rng)
    payload = int(np.asarray(lens)[ok].sum())
    d_valid = torch.from_numpy(valid.view(np.uint8).copy()).cuda()
    d_src = torch.from_numpy(str16.reshape(-1).copy()).cuda()
    d_heap = torch.from_numpy(heap).cuda()
    o_valid = torch.zeros((n + 7) // 8 + 16, dtype=torch.uint8, device="cuda")
    o_off = torch.zeros(4 * (n + 1) + 16, dtype=torch.uint8, device="cuda")
    o_data = torch.full((payload + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    t = da.make_task(_ffi.K_ENC_STR32, n, d_src.data_ptr(), o_off.data_ptr(), validity=d_valid.data_ptr(),
                     out_validity=o_valid.data_ptr(), out_aux=o_data.data_ptr(), buf2=d_heap.data_ptr(), ptr_base=0,
                     buf2_len=heap.size)
    plan = da.Plan(con.ctx, [t])
    plan.launch(torch.cuda.current_stream().cuda_stream)
    assert plan.status() == 0
    want_off = np.zeros(n + 1, np.int32)
    want_data = np.zeros(payload + 1, np.uint8)
    rc = po.lib().orc_enc_varchar32(str16.ctypes.data, valid.ctypes.data, n, 0, 0, heap.ctypes.data, want_off.ctypes.data,
                                    want_data.ctypes.data)
    assert rc == 0
    assert np.array_equal(o_off.cpu().numpy()[: 4 * (n + 1)].view(np.int32), want_off)
    got = o_data.cpu().numpy()
    assert np.array_equal(got[:payload], want_data[:payload])
    assert np.all(got[payload:] == 0xEE)          # nothing written past the payload
    assert plan.null_counts()[0] == int(n - ok.sum())


@pytest.mark.parametrize("layout", ["arrow_like", "nulls_take_no_heap", "long_strings_only", "every_other_wave_shuffled",
                                    "long_mean_90_arrow_like", "long_mean_90_shuffled"])
def test_encode_string_heap_layouts_match_oracle(con, torch, layout):
    """K7d picks, per wave of 64 rows, between one coalesced copy of the heap bytes between the first and the last long
    string (when the long strings lie in the heap as they will lie in the data buffer) and the per-row path.  Layouts:
    an Arrow-like heap (every row's bytes in row order; the slots of inline strings hold garbage here, the inline bytes
    must win), NULL rows that own no heap bytes, a heap of long strings only (DuckDB's own), waves with shuffled
    pointers next to contiguous ones, sub-blocks whose contiguous run crosses several LDS windows, and the same long strings
    (up to 180 bytes: more than the four 16-byte pieces the per-row path keeps in flight, and rows that a window cuts
    anywhere) with shuffled pointers."""
    rng = np.random.default_rng({"arrow_like": 21, "nulls_take_no_heap": 22, "long_strings_only": 23,
                                 "every_other_wave_shuffled": 24, "long_mean_90_arrow_like": 25, "long_mean_90_shuffled": 26}[layout])
    n = 9000
    lens = rng.integers(0, 181, n) if layout.startswith("long_mean_90") else rng.integers(0, 61, n)
    ok = np.ones(n, bool) if layout in ("arrow_like", "long_mean_90_arrow_like") else rng.random(n) < 0.9
    if layout == "long_strings_only":
        lens[rng.random(n) < 0.5] = 20    # runs of long strings back to back: contiguous waves exist in this layout too
        lens[2048:2048 + 640] = 33
    text = [rng.integers(97, 123, int(l), dtype=np.uint8) for l in lens]
    heap_len = {"arrow_like": lambda i: lens[i], "long_mean_90_arrow_like": lambda i: lens[i],
                "nulls_take_no_heap": lambda i: lens[i] if ok[i] else 0,
                "long_strings_only": lambda i: lens[i] if lens[i] > 12 else 0,
                "every_other_wave_shuffled": lambda i: lens[i], "long_mean_90_shuffled": lambda i: lens[i]}[layout]
    order = np.arange(n)
    if layout in ("every_other_wave_shuffled", "long_mean_90_shuffled"):
        for w in range(0, n // 64, 2):
            order[64 * w: 64 * w + 64] = rng.permutation(order[64 * w: 64 * w + 64])
    starts = np.zeros(n, np.int64)
    pos = 7                                   # an odd heap start: source and destination phases differ
    for i in order:
        starts[i] = pos
        pos += int(heap_len(i))
    heap = np.full(pos + 64, 0xAA, np.uint8)  # 0xAA wherever no long string lives (incl. the slots of inline strings)
    str16 = np.zeros((n, 16), np.uint8)
    str16[:, :4] = lens.astype(np.uint32).view(np.uint8).reshape(n, 4)
    for i in range(n):
        l = int(lens[i])
        if l <= 12:
            str16[i, 4: 4 + l] = text[i]
        else:
            heap[starts[i]: starts[i] + l] = text[i]
            str16[i, 4:8] = text[i][:4]
            str16[i, 8:16] = np.frombuffer(np.uint64(starts[i]).tobytes(), np.uint8)
    valid = np.packbits(np.concatenate([ok, np.ones((-n) % 64, bool)]), bitorder="little").view(np.uint64).copy()
    payload = int(lens[ok].sum())
    d_valid = torch.from_numpy(valid.view(np.uint8).copy()).cuda()
    d_src = torch.from_numpy(str16.reshape(-1).copy()).cuda()
    d_heap = torch.from_numpy(heap).cuda()
    o_valid = torch.zeros((n + 7) // 8 + 16, dtype=torch.uint8, device="cuda")
    o_off = torch.zeros(4 * (n + 1) + 16, dtype=torch.uint8, device="cuda")
    o_data = torch.full((payload + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    t = da.make_task(_ffi.K_ENC_STR32, n, d_src.data_ptr(), o_off.data_ptr(), validity=d_valid.data_ptr(),
                     out_validity=o_valid.data_ptr(), out_aux=o_data.data_ptr(), buf2=d_heap.data_ptr(), ptr_base=0,
                     buf2_len=heap.size)
    plan = da.Plan(con.ctx, [t])
    plan.launch(torch.cuda.current_stream().cuda_stream)
    assert plan.status() == 0
    want_off = np.zeros(n + 1, np.int32)
    want_data = np.zeros(payload + 1, np.uint8)
    rc = po.lib().orc_enc_varchar32(str16.ctypes.data, valid.ctypes.data, n, 0, 0, heap.ctypes.data, want_off.ctypes.data,
                                    want_data.ctypes.data)
    assert rc == 0
    assert np.array_equal(want_data[:payload], np.concatenate([text[i] for i in range(n) if ok[i]] + [np.zeros(0, np.uint8)]))
    assert np.array_equal(o_off.cpu().numpy()[: 4 * (n + 1)].view(np.int32), want_off)
    got = o_data.cpu().numpy()
    assert np.array_equal(got[:payload], want_data[:payload])
    assert np.all(got[payload:] == 0xEE)          # nothing written past the payload
    assert plan.null_counts()[0] == int(n - ok.sum())


# ---------------------------------------------------------------------------------------- test_arrow_ipc_writer.py
def create_table():
    return da.Table(["f0", "f1", "f2"], ["INTEGER", "VARCHAR", "BOOLEAN"],
                    [[1, 2, 3, 4], ["foo", "bar", "baz", None], [True, None, False, True]])


ROWS = [(1, "foo", True), (2, "bar", None), (3, "baz", False), (4, None, True)]


def test_round_trip(con):
    buffers = con.to_arrow_ipc(create_table())
    buffer = pa.py_buffer(buffers[0][0] + buffers[1][0])
    with pa.BufferReader(buffer) as buf_reader:
        msg_reader = ipc.MessageReader.open_stream(buf_reader)
        assert con.from_arrow(msg_reader).fetchall() == ROWS


def test_arrow_read_duck_buffers(con):
    buffers = con.to_arrow_ipc(create_table())
    assert buffers[0][1] is True and buffers[1][1] is False and len(buffers) == 2
    with pa.BufferReader(pa.py_buffer(buffers[0][0] + buffers[1][0])) as reader:
        stream_reader = ipc.RecordBatchStreamReader(reader)
        schema = stream_reader.schema
        batches = list(stream_reader)
    t = pa.Table.from_batches(batches, schema=schema)
    assert t.schema.types == [pa.int32(), pa.string(), pa.bool_()]
    assert t.to_pylist() == [dict(f0=a, f1=b, f2=c) for a, b, c in ROWS]
    # ArrowAppender always emits the validity bitmap and counts NULLs
    assert [c.null_count for c in t.columns] == [0, 1, 1]


def test_to_arrow_ipc_chunks_of_120_vectors(con):
    """to_arrow_ipc.test: one blob per 120 x 2048 rows (+ the schema blob)."""
    n = 2 * 120 * 2048 + 5
    t = da.Table(["a"], ["BIGINT"], [list(range(n))])
    blobs = con.to_arrow_ipc(t)
    assert [h for _, h in blobs] == [True, False, False, False]
    table = ipc.open_stream(pa.py_buffer(b"".join(b for b, _ in blobs))).read_all()
    assert table.num_rows == n and table["a"].to_pylist() == list(range(n))
    assert [len(b) for b in table["a"].chunks] == [245760, 245760, 5]


# ---------------------------------------------------------------------------------------- write_arrow_stream.test / test_copy_to.test
def commits_table(con, golden_dir):
    rel = con.read_arrow(os.path.join(golden_dir, "ref_data/test.arrows"))
    cols = rel.fetch_columns()
    return da.Table(rel.columns, rel.types, cols)


def test_copy_roundtrip_basic(con, tmp_path):
    p = str(tmp_path / "test.arrows")
    con.copy_to(da.Table(["foofy", "stringy"], ["INTEGER", "VARCHAR"], [[42], ["string"]]), p)
    assert con.read_arrow(p).fetchall() == [(42, "string")]
    assert con.read_arrow(p).project(["stringy"]).fetchall() == [("string",)]
    assert read_any(p).to_pylist() == [{"foofy": 42, "stringy": "string"}]


def test_write_then_read_equals_source(con, golden_dir, tmp_path):
    """write_arrow_stream.test:11-25: every row of the rewritten file equals the original."""
    t = commits_table(con, golden_dir)
    p = str(tmp_path / "rewritten.arrows")
    assert con.copy_to(t, p) == [p]
    back = con.read_arrow(p)
    assert back.types == t.types and back.fetch_columns() == t.columns
    orig = read_any(os.path.join(golden_dir, "ref_data/test.arrows"))
    mine = read_any(p)
    assert mine.num_rows == 15487 and mine.equals(orig.cast(mine.schema))
    assert mine.schema.field("time").type == pa.timestamp("us", tz="UTC")


def test_copy_options(con, golden_dir, tmp_path):
    t = commits_table(con, golden_dir)
    for opts in ({"row_group_size": 10}, {"chunk_size": 10}):
        p = str(tmp_path / ("rg_%s.arrow" % list(opts)[0]))
        con.copy_to(t, p, **opts)
        assert con.read_arrow(p).count() == 15487
        # "This actually has a minimum of 2048": one record batch per sunk DataChunk
        assert [b.num_rows for b in ipc.open_stream(p)] == [2048] * 7 + [1151]
    with pytest.raises(da.MiError, match="ROW_GROUP_SIZE and ROW_GROUP_SIZE_BYTES are mutually exclusive"):
        con.copy_to(t, str(tmp_path / "x.arrow"), row_group_size=100, chunk_size=10)
    with pytest.raises(da.MiError, match="ROW_GROUP_SIZE_BYTES does not work while preserving insertion order"):
        con.copy_to(t, str(tmp_path / "x.arrow"), row_group_size_bytes=100)
    p = str(tmp_path / "rgb.arrow")
    con.copy_to(t, p, preserve_insertion_order=False, row_group_size_bytes=100)
    assert con.read_arrow(p).count() == 15487


def test_row_groups_per_file_rotation(con, golden_dir, tmp_path):
    t = commits_table(con, golden_dir)
    d = str(tmp_path / "folder")
    files = con.copy_to(t, d, chunk_size=10, row_groups_per_file=1, format="ARROW")
    assert sorted(files) == sorted(glob.glob(os.path.join(d, "*")))
    assert len(files) == 9  # 8 data-bearing files + the one opened by the last rotation (test_copy_to.test:66-75 counts 9)
    assert con.read_arrow(os.path.join(d, "*")).count() == 15487


def test_kv_metadata(con, golden_dir, tmp_path):
    t = commits_table(con, golden_dir)
    p = str(tmp_path / "data_kv.arrow")
    con.copy_to(t, p, kv_metadata={"test": "works", "blob": b"\x00\x01"})
    assert con.read_arrow(p).count() == 15487
    assert read_any(p).schema.metadata == {b"test": b"works", b"blob": b"\x00\x01"}
    assert dict(da.Reader(path=p).schema_metadata()) == {"test": b"works", "blob": b"\x00\x01"}


def test_every_writable_type_roundtrips_through_pyarrow(con, tmp_path):
    n = 3000
    rng = np.random.default_rng(4)

    def nullable(vals):
        return [None if rng.random() < 0.15 else v for v in vals]

    cols = {
        ("b", "BOOLEAN"): nullable([bool(x) for x in rng.integers(0, 2, n)]),
        ("i8", "TINYINT"): nullable(rng.integers(-128, 127, n).tolist()),
        ("u16", "USMALLINT"): nullable(rng.integers(0, 65535, n).tolist()),
        ("i32", "INTEGER"): nullable(rng.integers(-2**31, 2**31 - 1, n).tolist()),
        ("i64", "BIGINT"): nullable(rng.integers(-2**63, 2**63 - 1, n).tolist()),
        ("u64", "UBIGINT"): nullable((rng.integers(0, 2**63 - 1, n).astype(np.uint64) * np.uint64(2)).tolist()),
        ("f64", "DOUBLE"): nullable(rng.standard_normal(n).tolist()),
        ("d4", "DECIMAL(4,1)"): nullable(rng.integers(-9999, 9999, n).tolist()),
        ("d15", "DECIMAL(15,2)"): nullable(rng.integers(-10**15 + 1, 10**15 - 1, n).tolist()),
        ("d38", "DECIMAL(38,5)"): nullable([int(x) * 10**15 + 7 for x in rng.integers(-10**18, 10**18, n)]),
        ("dt", "DATE"): nullable(rng.integers(-10000, 20000, n).tolist()),
        ("ts", "TIMESTAMP"): nullable(rng.integers(-10**15, 2 * 10**15, n).tolist()),
        ("s", "VARCHAR"): nullable([("x" * int(k)) + str(i) for i, k in enumerate(rng.integers(0, 30, n))]),
        ("bl", "BLOB"): nullable([bytes(rng.integers(0, 255, int(k)).astype(np.uint8)) for k in rng.integers(0, 30, n)]),
    }
    t = da.Table([k[0] for k in cols], [k[1] for k in cols], list(cols.values()))
    p = str(tmp_path / "types.arrows")
    con.copy_to(t, p, row_group_size=1000)
    back = con.read_arrow(p)
    assert back.types == t.types
    assert back.fetch_columns() == t.columns
    pt = read_any(p)
    assert pt.num_rows == n
    import decimal
    for (name, dt), vals in cols.items():
        got = pt[name].to_pylist()
        if dt.startswith("DECIMAL"):
            scale = int(dt[dt.index(",") + 1: -1])
            def unscaled(v):  # exact: Decimal.scaleb would round to the context's 28 digits
                sign, digits, exp = v.as_tuple()
                m = int("".join(map(str, digits))) * 10 ** (exp + scale)
                return -m if sign else m
            got = [None if v is None else unscaled(v) for v in got]
        elif dt == "DATE":
            got = pt[name].cast(pa.int32()).to_pylist()
        elif dt == "TIMESTAMP":
            got = pt[name].cast(pa.int64()).to_pylist()
        assert got == vals, name


# ---------------------------------------------------------------------------------------- nested types
NESTED_NAMES = ["l", "ll", "st", "lst", "arr", "mp", "s"]
NESTED_TYPES = ["INTEGER[]", "VARCHAR[][]", "STRUCT(a BIGINT, b VARCHAR)", "STRUCT(x DOUBLE, y INTEGER[])[]", "SMALLINT[2]",
                "MAP(VARCHAR, INTEGER)", "VARCHAR"]


def nested_rows(n, seed=3):
    rng = np.random.default_rng(seed)
    words = ["", "a", "hello", "twelve bytes", "thirteen byte", "a considerably longer string value"]
    maybe = lambda v, p=0.15: None if rng.random() < p else v
    w = lambda: words[int(rng.integers(0, len(words)))]
    cols = [[], [], [], [], [], [], []]
    for _ in range(n):
        cols[0].append(maybe([maybe(int(x)) for x in rng.integers(-9, 9, int(rng.integers(0, 5)))]))
        cols[1].append(maybe([maybe([maybe(w()) for _ in range(int(rng.integers(0, 3)))]) for _ in range(int(rng.integers(0, 3)))]))
        cols[2].append(maybe({"a": maybe(int(rng.integers(0, 10**12))), "b": maybe(w())}))
        cols[3].append(maybe([maybe({"x": maybe(float(rng.integers(0, 100)) / 4), "y": maybe([int(rng.integers(0, 5))])})
                              for _ in range(int(rng.integers(0, 3)))]))
        cols[4].append(maybe([maybe(int(rng.integers(-5, 5))), maybe(int(rng.integers(-5, 5)))]))
        cols[5].append(maybe([(w() + str(j), maybe(int(rng.integers(0, 50)))) for j in range(int(rng.integers(0, 3)))]))
        cols[6].append(maybe(w()))
    return cols


def test_copy_nested_types_roundtrip_through_pyarrow(con, tmp_path):
    """COPY a table with LIST / LIST of LIST / STRUCT / LIST of STRUCT / ARRAY / MAP columns (NULLs at every level) to
    .arrows: pyarrow reads back the same logical values and the Arrow types ArrowConverter::ToArrowSchema exports
    (list child "l", map entries / key / value); several record batches, child rows gathered in list order."""
    cols = nested_rows(5000)
    path = str(tmp_path / "nested.arrows")
    con.copy_to(da.Table(NESTED_NAMES, NESTED_TYPES, cols), path, row_group_size=2048)
    t = ipc.open_stream(path).read_all()
    assert t.num_rows == 5000
    assert t.schema.field("l").type == pa.list_(pa.field("l", pa.int32()))
    assert t.schema.field("arr").type == pa.list_(pa.field("l", pa.int16()), 2)
    assert t.schema.field("mp").type == pa.map_(pa.string(), pa.int32())
    assert t.schema.field("st").type == pa.struct([("a", pa.int64()), ("b", pa.string())])
    got = {name: t.column(name).to_pylist() for name in NESTED_NAMES}
    for name, want in zip(NESTED_NAMES, cols):
        if name == "st":   # pyarrow materialises the children of a NULL struct; compare valid structs only
            assert [g for g in got[name]] == want
        else:
            assert got[name] == want, name
    # validity bitmaps are always present and the NULL counts are exact
    rb = ipc.open_stream(path).read_next_batch()
    assert rb.column("l").null_count == sum(v is None for v in cols[0][:2048])


def test_nested_scan_then_copy_equals_source(con, golden_dir, tmp_path):
    """COPY (FROM read_arrow('edge_nested.arrows')) TO 'out.arrows': chunked nested vectors (window-relative list entries,
    shifted child validity) go straight from the scan into the sink; pyarrow sees the same table."""
    src = os.path.join(golden_dir, "edge_nested.arrows")
    out = str(tmp_path / "nested_copy.arrows")
    rel = con.read_arrow(src).project(["l_i", "l_s", "ll", "st", "l_st", "fl", "mp", "lgl"])
    con.copy_to(rel, out, row_group_size=3000)
    a = ipc.open_stream(src).read_all().select(rel.columns)
    b = ipc.open_stream(out).read_all()
    assert b.num_rows == a.num_rows
    for name in rel.columns:
        assert a.column(name).to_pylist() == b.column(name).to_pylist(), name


def test_arrow_large_buffer_size_exports_int64_offsets(con, tmp_path):
    """SET arrow_large_buffer_size=true (ClientProperties.arrow_offset_size, arrow_stream_writer.cpp:11-13): VARCHAR /
    BLOB / LIST columns are exported as LargeUtf8 / LargeBinary / LargeList with int64 offsets, MAP keeps int32."""
    cols = nested_rows(3000, seed=8)
    blobs = [None if i % 9 == 0 else bytes([i % 251]) * (i % 40) for i in range(3000)]
    path = str(tmp_path / "large.arrows")
    con.copy_to(da.Table(NESTED_NAMES + ["b"], NESTED_TYPES + ["BLOB"], cols + [blobs]), path, row_group_size=2048,
                arrow_large_buffer_size=True)
    t = ipc.open_stream(path).read_all()
    assert t.schema.field("s").type == pa.large_string() and t.schema.field("b").type == pa.large_binary()
    assert t.schema.field("l").type == pa.large_list(pa.field("l", pa.int32()))
    assert t.schema.field("ll").type == pa.large_list(pa.field("l", pa.large_list(pa.field("l", pa.large_string()))))
    assert t.schema.field("mp").type == pa.map_(pa.large_string(), pa.int32())
    for name, want in zip(NESTED_NAMES + ["b"], cols + [blobs]):
        assert t.column(name).to_pylist() == want, name
    # and the scan reads its own large output back (int64 offsets -> K4b / LIST64)
    back = con.read_arrow(path).project(["s", "l", "b"]).fetch_columns()
    assert back[0] == cols[6] and back[1] == cols[0] and back[2] == blobs


def test_sink_string_staging_modes_mix(con, tmp_path):
    """The sink stages long-string payloads as one growing run while chunks lay them out back to back, and gathers them
    string by string when they do not (shuffled pointers, a second heap); a row group that mixes both must still come out
    right.  Chunks are built by hand: string_t rows pointing into numpy heaps."""
    import ctypes as C
    rng = np.random.default_rng(21)
    L = _ffi.lib()
    o = _ffi.WriteOptions()
    _ffi.check(L.mi_write_options_init(C.byref(o)))
    _ffi.check(L.mi_write_options_set(C.byref(o), b"row_group_size", b"100000"))
    _ffi.check(L.mi_write_options_finalize(C.byref(o)))
    fields = (_ffi.Field * 1)()
    fields[0].name, fields[0].duck_type = b"s", b"VARCHAR"
    path = str(tmp_path / "mix.arrows")
    w = C.c_void_p()
    _ffi.check(L.mi_writer_open(con.ctx._h, path.encode(), fields, 1, C.byref(o), C.byref(w)))
    keep, want = [], []

    def sink(strings, order):
        """strings laid out in one heap in `order`; rows keep their original order"""
        n = len(strings)
        enc = [None if s is None else s.encode() for s in strings]
        heap = np.zeros(sum(len(e) for e in enc if e is not None) + 64, np.uint8)
        at, pos = {}, 0
        for i in order:
            if enc[i] is not None:
                heap[pos: pos + len(enc[i])] = np.frombuffer(enc[i], np.uint8)
                at[i] = pos
                pos += len(enc[i])
        data = np.zeros((n, 16), np.uint8)
        ok = np.ones(n, bool)
        for i, e in enumerate(enc):
            if e is None:
                ok[i] = False
                data[i, :] = 0xAB          # garbage in a NULL row must not matter
                continue
            data[i, :4] = np.frombuffer(np.uint32(len(e)).tobytes(), np.uint8)
            if len(e) <= 12:
                data[i, 4: 4 + len(e)] = np.frombuffer(e, np.uint8)
            else:
                data[i, 4:8] = np.frombuffer(e[:4], np.uint8)
                data[i, 8:] = np.frombuffer(np.uint64(heap.ctypes.data + at[i]).tobytes(), np.uint8)
        valid = np.packbits(np.concatenate([ok, np.ones((-n) % 64, bool)]), bitorder="little").view(np.uint64).copy()
        vec = (_ffi.Vector * 1)()
        vec[0].data, vec[0].validity, vec[0].count = data.ctypes.data, valid.ctypes.data, n
        ch = _ffi.DataChunk(size=n, n_columns=1, columns=vec)
        keep.extend([heap, data, valid, vec])
        _ffi.check(L.mi_writer_sink(w, C.byref(ch)))
        want.extend(strings)

    def strings(n):
        return [None if rng.random() < 0.1 else "x" * int(rng.integers(0, 40)) + str(i) for i in range(n)]

    for rep in range(3):
        s1 = strings(2048)
        sink(s1, range(2048))                              # back to back: starts / continues a run
        s2 = strings(1500)
        sink(s2, list(rng.permutation(1500)))              # shuffled inside its heap: gathered
        s3 = strings(2048)
        sink(s3, range(2048))                              # ordered again, but the run is closed for this row group
        sink(["short"] * 100, range(100))                  # inline only
    _ffi.check(L.mi_writer_finalize(w))
    L.mi_writer_close(w)
    got = ipc.open_stream(path).read_all().column("s").to_pylist()
    assert got == want


# ---------------------------------------------------------------------------------------- several sink threads
def test_parallel_copy_pump_writes_the_one_thread_file(con, tmp_path, monkeypatch):
    """mi_writer_sink_scan with several sink threads (MI_WRITER_THREADS): row groups are cut where the one-thread sink cuts
    them and claim their file ranges in input order, so the file is byte-identical to the one-thread file -- for record
    batches larger than, equal to and smaller than row_group_size, rows that are no multiple of 2048, NULLs and long strings."""
    import duckdb_arrow_amd as da
    rng = np.random.default_rng(21)
    n = 70000
    t = pa.table({"k": pa.array(np.arange(n, dtype=np.int64)),
                  "d": pa.array([__import__("decimal").Decimal(int(v)) for v in rng.integers(0, 1000, n)], pa.decimal128(15, 0)),
                  "s": pa.array(["str %d %s" % (i, "y" * int(k)) for i, k in enumerate(rng.integers(0, 40, n))], mask=rng.random(n) < 0.1),
                  "f": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.2),
                  "ls": pa.array(["large %d" % (i % 1001) if i % 7 else None for i in range(n)], pa.large_string()),
                  "dt": pa.array(rng.integers(8000, 11000, n).astype(np.int32), pa.date32()),
                  "ts": pa.array(rng.integers(0, 2**40, n), pa.timestamp("s", tz="UTC"))})
    for chunk, rgs in ((9000, 9000), (25000, 8192), (3000, 10000), (7001, 5000), (70000, 20000)):
        src = str(tmp_path / ("src_%d.arrows" % chunk))
        with ipc.new_stream(src, t.schema) as w:
            w.write_table(t, max_chunksize=chunk)
        outs = []
        # the one-thread sink; four sink threads; the fused pump (record batches encoded where they lie in HBM)
        for threads, fused in (("1", False), ("4", False), ("4", True)):
            monkeypatch.setenv("MI_WRITER_THREADS", threads)
            if fused:
                monkeypatch.delenv("MI_WRITER_NO_FUSED", raising=False)
            else:
                monkeypatch.setenv("MI_WRITER_NO_FUSED", "1")
            out = str(tmp_path / ("out_%d_%s_%d.arrows" % (chunk, threads, fused)))
            con.copy_to(con.read_arrow(src), out, row_group_size=rgs)
            outs.append(open(out, "rb").read())
        assert outs[0] == outs[1] and outs[0] == outs[2], (chunk, rgs)
        got = ipc.open_stream(pa.BufferReader(outs[1])).read_all()
        # DuckDB's types on the way out: VARCHAR is utf8 whatever the input offsets were, TIMESTAMP WITH TIME ZONE is microseconds
        want = t.set_column(t.schema.get_field_index("ls"), "ls", t.column("ls").cast(pa.string())) \
                .set_column(t.schema.get_field_index("ts"), "ts", t.column("ts").cast(pa.timestamp("us", tz="UTC")))
        assert got.equals(want), (chunk, rgs)
        sizes = [b.num_rows for b in ipc.open_stream(pa.BufferReader(outs[1]))]
        assert sum(sizes) == n and all(x >= rgs for x in sizes[:-1]) and all(x < rgs + 2048 for x in sizes)


@pytest.mark.timeout(180, method="thread")   # a hang inside the C call must end the process, not the box's time limit
def test_copy_pump_row_group_spans_more_batches_than_slots(con, tmp_path, monkeypatch):
    """One row group made of far more record batches than the scan has pipeline slots (pyarrow max_chunksize=1000 against
    row_group_size 40960 = 41 batches per group, 6 sink threads -> 10 slots): the pump stages the rows of the unfinished row
    group itself and gives the slots back instead of waiting for a release that cannot come.  Same file as the one-thread sink."""
    rng = np.random.default_rng(5)
    n = 130000
    t = pa.table({"k": pa.array(np.arange(n, dtype=np.int64)),
                  "s": pa.array(["row %d %s" % (i, "z" * int(k)) for i, k in enumerate(rng.integers(0, 30, n))], mask=rng.random(n) < 0.1),
                  "d": pa.array(rng.integers(8000, 11000, n).astype(np.int32), pa.date32())})
    src = str(tmp_path / "small_batches.arrows")
    with ipc.new_stream(src, t.schema) as w:
        w.write_table(t, max_chunksize=1000)
    outs = []
    for threads in ("1", "6", "2"):
        monkeypatch.setenv("MI_WRITER_THREADS", threads)
        out = str(tmp_path / ("out_%s.arrows" % threads))
        con.copy_to(con.read_arrow(src), out, row_group_size=40960)
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] == outs[2]
    got = ipc.open_stream(pa.BufferReader(outs[1]))
    sizes = [b.num_rows for b in got]
    assert sum(sizes) == n and all(x >= 40960 for x in sizes[:-1])
    assert ipc.open_stream(pa.BufferReader(outs[1])).read_all().equals(t)


def test_local_sink_states_from_several_threads(con, tmp_path):
    """mi_writer_local_*: every thread buffers, encodes and writes its own row groups (ArrowWriteSink with per-thread local
    state, write_arrow_stream.cpp:141-159); row groups land in completion order, every row exactly once."""
    import ctypes as C
    import threading
    import duckdb_arrow_amd as da
    from duckdb_arrow_amd import _ffi
    L = _ffi.lib()
    names, types = ["t", "i", "s"], ["INTEGER", "BIGINT", "VARCHAR"]
    path = str(tmp_path / "threads.arrows")
    o = _ffi.WriteOptions()
    _ffi.check(L.mi_write_options_init(C.byref(o)))
    _ffi.check(L.mi_write_options_set(C.byref(o), b"row_group_size", b"5000"))
    _ffi.check(L.mi_write_options_finalize(C.byref(o)))
    w = C.c_void_p()
    _ffi.check(L.mi_writer_open(con.ctx._h, path.encode(), da._c_fields(names, types), 3, C.byref(o), C.byref(w)))
    errors = []

    def work(tid):
        try:
            keep = []
            tab = da.Table(names, types, [[tid] * 12000, list(range(12000)), ["thread %d row %d long enough to leave the struct" % (tid, i) for i in range(12000)]])
            loc = C.c_void_p()
            _ffi.check(L.mi_writer_local_create(w, C.byref(loc)))
            for ch in da._chunks_from_table(tab, keep):
                _ffi.check(L.mi_writer_local_sink(loc, C.byref(ch)))
            _ffi.check(L.mi_writer_local_combine(loc))
            L.mi_writer_local_destroy(loc)
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [x.start() for x in ts]
    [x.join() for x in ts]
    assert not errors, errors
    _ffi.check(L.mi_writer_finalize(w))
    assert L.mi_writer_row_groups(w) == 4 * 2   # per thread: 6144 rows (3 chunks reach 5000), then the 5856-row tail
    L.mi_writer_close(w)
    got = ipc.open_stream(path).read_all()
    assert got.num_rows == 48000
    rows = sorted(zip(got.column("t").to_pylist(), got.column("i").to_pylist(), got.column("s").to_pylist()))
    assert rows == [(t_, i, "thread %d row %d long enough to leave the struct" % (t_, i)) for t_ in range(4) for i in range(12000)]
