"""Seeded mutation fuzzing of the host IPC reader (CPU only): corrupted or truncated streams must end in a clean MiError
(or decode to something), never in a crash -- the flatbuffer reader is bounds checked and every buffer span is validated
against the body before anything is sliced (the reference gets this from nanoarrow's verifier + FULL validation)."""
import os

import numpy as np
import pytest

import duckdb_arrow_amd as da
from oracle import pyoracle as po

FILES = ["ref_data/test.arrows", "edge_nested.arrows", "edge_types.arrows", "edge_dict.arrows", "edge_file_format.arrow"]


def _drain(buf):
    rd = da.Reader(buffers=[buf])
    try:
        rd.schema()
        try:
            rd.index()
        except da.MiError:
            pass
        n = 0
        while n < 64:
            b = rd.next_batch(accept_dictionaries=True)
            if b is None:
                break
            n += 1
            for nd in b["nodes"]:   # every span the reader hands out lies inside the body
                for off, ln in nd["spans"]:
                    assert 0 <= off and off + ln <= b["body_size"]
    except da.MiError:
        pass
    finally:
        rd.close()


@pytest.mark.parametrize("rel", FILES)
def test_mutated_streams_never_crash_the_reader(golden_dir, rel):
    src = np.fromfile(os.path.join(golden_dir, rel), np.uint8)
    msgs = po.walk_stream(src)
    rng = np.random.default_rng(abs(hash(rel)) % 2**32)
    meta_ranges = [(m["meta_off"], m["meta_len"]) for m in msgs if m["meta_len"] > 0][:6]
    for it in range(int(os.environ.get("MI_FUZZ_ITERS", "250"))):
        buf = src.copy()
        how = it % 5
        if how == 0:     # flip bytes inside a metadata flatbuffer
            off, ln = meta_ranges[int(rng.integers(0, len(meta_ranges)))]
            for _ in range(int(rng.integers(1, 6))):
                buf[off + int(rng.integers(0, ln))] = int(rng.integers(0, 256))
        elif how == 1:   # overwrite a 4-byte word of a metadata flatbuffer with an extreme value
            off, ln = meta_ranges[int(rng.integers(0, len(meta_ranges)))]
            p = off + int(rng.integers(0, max(ln - 4, 1)))
            buf[p: p + 4] = np.frombuffer(np.int32(rng.choice([-1, 0, 2**31 - 1, -2**31, 0x7fffff00])).tobytes(), np.uint8)
        elif how == 2:   # truncate anywhere
            buf = buf[: int(rng.integers(0, buf.size))]
        elif how == 3:   # corrupt a message prefix (continuation token / metadata length)
            m = msgs[int(rng.integers(0, min(len(msgs), 6)))]
            p = m["prefix_off"] + int(rng.integers(0, 8))
            buf[p] = int(rng.integers(0, 256))
        else:            # random bytes anywhere in the first 64 KB
            for _ in range(8):
                buf[int(rng.integers(0, min(buf.size, 65536)))] = int(rng.integers(0, 256))
        _drain(buf)


@pytest.mark.parametrize("codec", ["zstd", "lz4"])
def test_mutated_compressed_bodies_never_crash_the_reader(codec):
    import pyarrow as pa
    import pyarrow.ipc as ipc
    rng = np.random.default_rng(77)
    t = pa.table({"a": rng.integers(0, 50, 20000), "s": ["row %d" % (i % 97) for i in range(20000)],
                  "l": pa.array([[int(x) for x in rng.integers(0, 9, int(rng.integers(0, 4)))] for _ in range(20000)], pa.list_(pa.int32()))})
    sink = pa.BufferOutputStream()
    with ipc.new_stream(sink, t.schema, options=ipc.IpcWriteOptions(compression=codec)) as w:
        w.write_table(t, max_chunksize=6000)
    src = np.frombuffer(sink.getvalue(), np.uint8)
    msgs = po.walk_stream(src)
    bodies = [(m["body_off"], m["body_len"]) for m in msgs if m["body_len"] > 0]
    for it in range(int(os.environ.get("MI_FUZZ_ITERS", "250"))):
        buf = src.copy()
        off, ln = bodies[int(rng.integers(0, len(bodies)))]
        if it % 2 == 0:   # damage compressed frames / the per-buffer length prefixes
            for _ in range(int(rng.integers(1, 8))):
                buf[off + int(rng.integers(0, ln))] = int(rng.integers(0, 256))
        else:             # an absurd uncompressed length in front of a frame
            p = off + int(rng.integers(0, max(ln - 8, 1))) // 8 * 8
            buf[p: p + 8] = np.frombuffer(np.int64(rng.choice([-2, 2**40, 2**62, 0, 7])).tobytes(), np.uint8)
        _drain(buf)


def _utf8_stream():
    import pyarrow as pa
    import pyarrow.ipc as ipc
    t = pa.table({"s": ["value %d" % i for i in range(100)], "k": pa.array(range(100), pa.int64())})
    sink = pa.BufferOutputStream()
    with ipc.new_stream(sink, t.schema) as w:
        w.write_table(t)
    return np.frombuffer(sink.getvalue(), np.uint8).copy()


def test_buffer_span_near_int64_max_is_rejected_not_wrapped():
    """`offset + length > size` wraps for offset = 0x7FFFFFFFFFFFFFF8 and passed the bounds check; the
    span then reached pointer arithmetic.  Random mutation never produces this value, so it is seeded."""
    src = _utf8_stream()
    rd = da.Reader(buffers=[src])
    rd.schema()
    b = rd.next_batch()
    off, ln = b["nodes"][0]["spans"][1]   # the offsets buffer of `s`
    rd.close()
    msgs = [m for m in po.walk_stream(src) if m["type"] == po.MSG_RECORD_BATCH]
    meta = src[msgs[0]["meta_off"]: msgs[0]["meta_off"] + msgs[0]["meta_len"]]
    needle = np.array([off, ln], np.int64).view(np.uint8).tobytes()
    at = meta.tobytes().find(needle)
    assert at >= 0
    for bad_off, bad_len in [(0x7FFFFFFFFFFFFFF8, 420), (0x7FFFFFFFFFFFFFF8, 8), (8, 0x7FFFFFFFFFFFFFF8), (-8, 16), (0, -1)]:
        buf = src.copy()
        p = msgs[0]["meta_off"] + at
        buf[p: p + 16] = np.array([bad_off, bad_len], np.int64).view(np.uint8)
        rd = da.Reader(buffers=[buf])
        rd.schema()
        with pytest.raises(da.MiError, match="Buffer requires body offsets|size >="):
            rd.next_batch()
        rd.close()
        with pytest.raises(Exception):   # the Arrow C stream export walks the offsets: it must refuse first
            da.Reader(buffers=[buf]).export_stream().read_all()


def _patch_every_int32(src, lo, hi, value, bad):
    """Yields copies of `src` with one 4-byte aligned int32 == value inside [lo, hi) replaced by `bad`."""
    words = src[lo: lo + (hi - lo) // 4 * 4].view(np.int32)
    for i in np.nonzero(words == value)[0]:
        buf = src.copy()
        buf[lo + 4 * int(i): lo + 4 * int(i) + 4] = np.array([bad], np.int32).view(np.uint8)
        yield buf


@pytest.mark.parametrize("what,value,bads,expect", [
    ("fixed_size_binary", 0x1234, [-1, 0, -2**31], "FixedSizeBinary byteWidth"),
    ("fixed_size_list", 0x1234, [-1, -2**31], "FixedSizeList listSize"),
    ("dictionary", 16, [0, 24, -8, 7], "dictionary index bit width"),
])
def test_schema_scalars_used_as_widths_are_validated(what, value, bads, expect):
    """byteWidth / listSize / index bitWidth from the file were used as widths unchecked (negative
    byteWidth => negative size products pass every check; index width 24 => 8-byte reads of 3-byte slots)."""
    import pyarrow as pa
    import pyarrow.ipc as ipc
    if what == "fixed_size_binary":
        t = pa.table({"x": pa.array([b"\0" * 0x1234] * 3, pa.binary(0x1234))})
    elif what == "fixed_size_list":
        t = pa.table({"x": pa.array([[1] * 0x1234] * 2, pa.list_(pa.int8(), 0x1234))})
    else:
        t = pa.table({"x": pa.array(["a", "b", "a", None]).dictionary_encode().cast(pa.dictionary(pa.int16(), pa.utf8()))})
    sink = pa.BufferOutputStream()
    with ipc.new_stream(sink, t.schema) as w:
        w.write_table(t)
    src = np.frombuffer(sink.getvalue(), np.uint8).copy()
    m = po.walk_stream(src)[0]
    seen = []
    for bad in bads:
        for buf in _patch_every_int32(src, m["meta_off"], m["meta_off"] + m["meta_len"], value, bad):
            rd = da.Reader(buffers=[buf])
            try:
                rd.schema()
                while rd.next_batch(accept_dictionaries=True) is not None:
                    pass
            except da.MiError as e:
                seen.append(str(e))
            finally:
                rd.close()
    assert any(expect in s for s in seen), seen
