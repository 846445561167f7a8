"""Pins the CPU oracle against the golden fixtures: the reference's own data files and known answers
(test/sql/read_arrow.test:35-55, test/sql/read_arrow_file.test:9-17, test/sql/multifile_reading.test:8-25,
test/nodejs/arrow_test.js:423-424, test/python/test_arrow_ipc_scan.py:7-17) with pyarrow-computed logical values
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from helpers import canon_stream, column_digest

STREAM_FILES = [
    "ref_data/test.arrows", "ref_data/fruit.arrow", "ref_data/multifile/glob/f1.arrow",
    "ref_data/multifile/glob/f2.arrow", "ref_data/multifile/glob/f3.arrow",
    "ref_data/multifile/different_order.arrows", "ref_data/multifile/different_type.arrows",
    "ref_data/multifile/different_type_int.arrows", "ref_data/multifile/different_type_order.arrows",
    "ref_data/multifile/fruit_extra.arrows", "ref_data/multifile/hive/part=a/f1.arrow",
    "ref_data/multifile/hive/part=a/f2.arrow", "ref_data/multifile/hive/part=b/f1.arrow",
    "ref_data/multifile/hive/part=b/f3.arrow", "lineitem_sf0_01_q6.arrows", "lineitem_sf0_01_head.arrows",
    "edge_reftest.arrows", "edge_types.arrows", "edge_types2.arrows", "edge_nested.arrows", "edge_empty.arrows", "edge_dict.arrows", "edge_file_format.arrow",
]


def load(golden_dir, rel):
    return np.fromfile(os.path.join(golden_dir, rel), dtype=np.uint8)


@pytest.mark.parametrize("rel", STREAM_FILES)
def test_oracle_matches_pyarrow_logical_values(golden_dir, expected, rel):
    buf = load(golden_dir, rel)
    fields, batches = po.decode_stream(buf)
    assert all(c["rc"] == 0 for b in batches for c in b["columns"])
    exp = expected[rel]
    assert sum(b["nrows"] for b in batches) == exp["rows"]
    got = canon_stream(fields, batches, buf)
    assert sorted(got.keys()) == sorted(exp["columns"].keys())
    for name, values in got.items():
        assert len(values) == exp["rows"]
        assert sum(v is None for v in values) == exp["null_counts"][name], name
        assert column_digest(values) == exp["columns"][name], name


def test_kat_test_arrows(golden_dir, expected):
    """read_arrow.test:35-55: 15487 rows, the ARROW-1 commit message, 2927 Wednesdays."""
    kat = expected["kat"]
    buf = load(golden_dir, "ref_data/test.arrows")
    msgs = po.walk_stream(buf)
    assert [m["type"] for m in msgs] == [po.MSG_SCHEMA] + [po.MSG_RECORD_BATCH] * 16
    rb = po.decode_record_batch(buf[msgs[1]["meta_off"]: msgs[1]["meta_off"] + msgs[1]["meta_len"]])
    assert [list(b) for b in rb["buffers"]] == kat["test_arrows_batch0_buffers"]
    fields, batches = po.decode_stream(buf)
    got = canon_stream(fields, batches, buf)
    assert len(got["commit"]) == kat["test_arrows_rows"] == 15487
    i = got["commit"].index("fa5f0299f046c46e1b2f671e5e3b4f1956522711")
    assert [got["message"][i]] == kat["test_arrows_commit_message"]
    micros = np.array(got["time"], dtype=np.int64)
    days = np.floor_divide(micros, 86400000000)
    assert int(np.sum((days + 3) % 7 == 2)) == kat["test_arrows_wednesday"] == 2927


def test_kat_tpch_q6(golden_dir, expected):
    """arrow_test.js:423-424: Q6 revenue on lineitem SF0.01 = 1193053.2253, computed from the oracle's
    decimal128 -> int64 narrowing and the K6 range filter."""
    kat = expected["kat"]
    buf = load(golden_dir, "lineitem_sf0_01_q6.arrows")
    fields, batches = po.decode_stream(buf)
    revenue, passing, shipsel = 0, 0, 0
    for b in batches:
        cols = {c["name"]: c for c in b["columns"]}
        n = b["nrows"]
        ship = cols["l_shipdate"]["data"].view(np.int32)
        sel = np.zeros(n, dtype=np.uint32)
        cnt = po.lib().orc_filter_range_i32(ship.ctypes.data, cols["l_shipdate"]["validity"].ctypes.data,
                                            po.C.c_int64(n), 8766, 9131, sel.ctypes.data)
        rows = sel[:cnt]
        shipsel += cnt
        qty = cols["l_quantity"]["data"].view(np.int64)[rows]
        price = cols["l_extendedprice"]["data"].view(np.int64)[rows]
        disc = cols["l_discount"]["data"].view(np.int64)[rows]
        keep = (disc >= 5) & (disc <= 7) & (qty < 2400)
        revenue += int(np.sum(price[keep] * disc[keep]))
        passing += int(keep.sum())
    assert shipsel == kat["shipdate_1994_selected"]
    assert passing == kat["q6_sf0_01_rows_passing"] == 1191
    assert revenue == kat["q6_sf0_01_revenue_scale4"] == 11930532253


def test_two_file_list_count(golden_dir):
    """multifile_reading.test:8-14: two copies of test.arrows = 30974 rows."""
    buf = load(golden_dir, "ref_data/test.arrows")
    total = 0
    for _ in range(2):
        _, batches = po.decode_stream(buf)
        total += sum(b["nrows"] for b in batches)
    assert total == 30974


def test_file_footer(golden_dir):
    """data/fruit.arrow is a true IPC file (read_arrow_file.test:9-17): footer blocks point at the same
    record batches the embedded-stream walk finds."""
    for rel in ("ref_data/fruit.arrow", "edge_file_format.arrow"):
        buf = load(golden_dir, rel)
        blocks, ndict = po.decode_footer(buf)
        msgs = [m for m in po.walk_stream(buf) if m["type"] == po.MSG_RECORD_BATCH]
        assert ndict == 0 and len(blocks) == len(msgs)
        for (off, meta_len, body_len), m in zip(blocks, msgs):
            assert off == m["prefix_off"] and meta_len == m["meta_len"] + 8 and body_len == m["body_len"]


# ------------------------------------------------------------------------------------ framing error cases
def test_bad_continuation_token():
    """ipc_file_stream_reader.cpp:121-124"""
    bad = np.zeros(64, np.uint8)
    bad[:4] = [1, 2, 3, 4]
    with pytest.raises(IOError, match=r"Expected continuation token \(0xFFFFFFFF\) but got 67305985"):
        po.walk_stream(bad)


def test_negative_metadata_size():
    """base_stream_reader.cpp:222-225"""
    bad = np.zeros(64, np.uint8)
    bad[:4] = 0xFF
    bad[4:8] = np.frombuffer(np.int32(-5).tobytes(), np.uint8)
    with pytest.raises(IOError, match="Expected metadata size >= 0 but got -5"):
        po.walk_stream(bad)


def test_truncated_stream_is_end_of_stream(golden_dir):
    """ipc_file_stream_reader.cpp:126-131: an input that ends where a prefix should start (no EOS marker) just ends;
    an input cut inside a message is an error (DecodeMessage runs outside the try block)."""
    buf = load(golden_dir, "ref_data/test.arrows")
    full = po.walk_stream(buf)
    no_eos = po.walk_stream(buf[: full[-1]["body_off"] + full[-1]["body_len"]])
    assert len(no_eos) == len(full)
    three = po.walk_stream(buf[: full[2]["body_off"] + full[2]["body_len"]])
    assert len(three) == 3
    with pytest.raises(IOError, match="not enough data in file to deserialize result"):
        po.walk_stream(buf[: full[3]["body_off"] + 100])
    with pytest.raises(IOError, match="not enough data in file to deserialize result"):
        po.walk_stream(buf[: full[3]["meta_off"] + 10])


def test_offset_validation():
    """NANOARROW_VALIDATION_LEVEL_FULL (base_stream_reader.cpp:117): offsets must be non-decreasing and inside data."""
    good = np.array([0, 3, 3, 7], np.int32)
    assert po.lib().orc_validate_offsets32(good.ctypes.data, po.C.c_int64(3), po.C.c_int64(7)) == 0
    assert po.lib().orc_validate_offsets32(good.ctypes.data, po.C.c_int64(3), po.C.c_int64(6)) != 0
    bad = np.array([0, 5, 3, 7], np.int32)
    assert po.lib().orc_validate_offsets32(bad.ctypes.data, po.C.c_int64(3), po.C.c_int64(7)) != 0
    neg = np.array([-1, 5, 6, 7], np.int32)
    assert po.lib().orc_validate_offsets32(neg.ctypes.data, po.C.c_int64(3), po.C.c_int64(7)) != 0


# ------------------------------------------------------------------------------------ kernel-level semantics
def test_validity_shift_matches_bitwise():
    """K1 at arbitrary bit offsets equals the bit-by-bit definition; pad bits canonical = 1."""
    rng = np.random.default_rng(1)
    bitmap = rng.integers(0, 256, 600, dtype=np.uint8)
    bits = np.unpackbits(bitmap, bitorder="little")
    for o in (0, 1, 7, 8, 13, 64, 2048, 2051):
        for n in (1, 5, 63, 64, 65, 700, 2048):
            out = np.zeros((n + 63) // 64, np.uint64)
            po.lib().orc_validity(bitmap.ctypes.data, po.C.c_int64(-1), po.C.c_int64(o), po.C.c_int64(n),
                                  out.ctypes.data)
            got = np.unpackbits(out.view(np.uint8), bitorder="little")
            assert (got[:n] == bits[o: o + n]).all(), (o, n)
            assert got[n:].all(), (o, n)
    out = np.zeros(2, np.uint64)
    po.lib().orc_validity(bitmap.ctypes.data, po.C.c_int64(0), po.C.c_int64(3), po.C.c_int64(100), out.ctypes.data)
    assert (out == np.uint64(0xFFFFFFFFFFFFFFFF)).all()  # null_count == 0 => all valid, bitmap ignored


def test_string_t_layout():
    """string_t: <=12 bytes inline zero padded; else 4-byte prefix + pointer = ptr_base + offset."""
    strings = [b"", b"a", b"twelve bytes", b"thirteen byte", b"x" * 40]
    off = np.cumsum([0] + [len(s) for s in strings]).astype(np.int32)
    data = np.frombuffer(b"".join(strings), np.uint8)
    d, v, rc = po.decode_column(po.K_STR32, 0, len(strings), None, off, data, 0, ptr_base=0x7000_0000_0000)
    s = d.reshape(-1, 16)
    assert rc == 0
    assert s[0].tolist() == [0] * 16
    assert s[1].tolist() == [1, 0, 0, 0, ord("a")] + [0] * 11
    assert s[2].tolist() == [12, 0, 0, 0] + list(b"twelve bytes")
    assert s[3][:8].tolist() == [13, 0, 0, 0] + list(b"thir")
    assert int(s[3][8:].view(np.uint64)[0]) == 0x7000_0000_0000 + int(off[3])
    assert int(s[4][8:].view(np.uint64)[0]) == 0x7000_0000_0000 + int(off[4])


def test_large_string_over_4gb_errors():
    off = np.array([0, 5, 2**32 + 10], np.int64)
    data = np.zeros(16, np.uint8)
    out = np.zeros(32, np.uint8)
    rc = po.lib().orc_string64(off.ctypes.data, data.ctypes.data, None, po.C.c_int64(0), po.C.c_int64(1),
                               po.C.c_uint64(0), out.ctypes.data)
    assert rc == 0
    rc = po.lib().orc_string64(off.ctypes.data, data.ctypes.data, None, po.C.c_int64(0), po.C.c_int64(2),
                               po.C.c_uint64(0), out.ctypes.data)
    assert rc != 0  # "DuckDB does not support Strings over 4GB"


def test_timestamp_multiply_overflow():
    src = np.array([1, 2**62], np.int64)
    out = np.zeros(2, np.int64)
    rc = po.lib().orc_mul_i64(src.ctypes.data, None, po.C.c_int64(0), po.C.c_int64(2), po.C.c_int64(1000000),
                              out.ctypes.data)
    assert rc != 0 and out[0] == 1000000


def test_scan_stream_driver_counts(golden_dir):
    buf = load(golden_dir, "ref_data/test.arrows")
    rc, st = po.scan_stream(buf, want_checksum=True)
    assert rc == 0 and st["rows"] == 15487 and st["batches"] == 16
    rc2, st2 = po.scan_stream(buf, want_checksum=True)
    assert st2["checksum"] == st["checksum"]
    rc3, st3 = po.scan_stream(buf, max_batches=3)
    assert st3["batches"] == 3 and st3["rows"] == 3072


def test_scan_stream_from_several_threads_equals_the_single_thread_scan():
    """bench.py's cpu_baseline_multifile runs one oracle scan per thread (one per file): the scan's scratch tables are
    thread-local, so concurrent scans of different streams give what each gives alone (rows, bytes, checksum)."""
    from concurrent.futures import ThreadPoolExecutor
    import duckdb_arrow_amd as da
    streams = [da.synth_lineitem_stream(scale_factor=0.01 * (i + 1), seed=7 + i, rows_per_batch=5000 + 700 * i)[0] for i in range(4)]
    alone = [po.scan_stream(b, want_checksum=True) for b in streams]
    assert all(rc == 0 and st["rows"] > 0 for rc, st in alone)
    with ThreadPoolExecutor(4) as ex:
        for _ in range(3):
            together = list(ex.map(lambda b: po.scan_stream(b, want_checksum=True), streams * 2))
    assert together == alone * 2
