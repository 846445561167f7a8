"""Shared test helpers: DuckDB-layout vectors -> canonical logical values (the same canonical form
tests/golden/make_golden.py computes with pyarrow), used for both the oracle and the HIP path."""
import hashlib
import json

import numpy as np

from oracle import pyoracle as po

# Arrow type ids (Schema.fbs Type union)
T_INT, T_FLOAT, T_BINARY, T_UTF8, T_BOOL, T_DECIMAL, T_DATE, T_TIME, T_TIMESTAMP = 2, 3, 4, 5, 6, 7, 8, 9, 10
T_FIXED_BINARY, T_DURATION, T_LARGE_BINARY, T_LARGE_UTF8 = 15, 18, 19, 20


def column_digest(values):
    return hashlib.sha256(json.dumps(values, separators=(",", ":")).encode()).hexdigest()


def canon_python(v):
    """Python values as the host mirror returns them (lists / dicts / (key, value) tuples / bytes / floats) -> the
    canonical JSON-able form of tests/golden/expected.json."""
    if v is None or isinstance(v, (bool, int, str)):
        return v
    if isinstance(v, (list, tuple)):
        return [canon_python(x) for x in v]
    if isinstance(v, dict):
        return {k: canon_python(x) for k, x in v.items()}
    if isinstance(v, float):
        return "nan" if v != v else repr(v)
    if isinstance(v, (bytes, bytearray)):
        return "b:" + bytes(v).hex()
    raise TypeError(type(v))


def _fixed(data, ok, dtype):
    vals = data.view(dtype)
    return [vals[i].item() if ok[i] else None for i in range(len(ok))]


def canon_flat(field, kind, param, width, data, validity, n, heap, heap_base=0):
    """One decoded flat vector -> canonical logical list.  `heap_base` = pointer value of heap[0]; with
    pyoracle.decode_stream's default pointer bases the heap is the whole stream and heap_base is 0."""
    ok = po.valid_bits(validity, n) if n else np.zeros(0, bool)
    t = field["type"]
    if kind in (po.K_STR32, po.K_STR64, po.K_FIXED_BINARY):
        as_bytes = t in (T_BINARY, T_LARGE_BINARY, T_FIXED_BINARY)
        vals = po.strings_to_pylist(data, validity, n, heap, heap_base, as_bytes=as_bytes)
        return [("b:" + v.hex()) if (as_bytes and v is not None) else v for v in vals]
    if kind == po.K_BOOL:
        return [bool(data[i]) if ok[i] else None for i in range(n)]
    if kind == po.K_NULL:
        return [None] * n
    if t == T_FLOAT:
        vals = data.view(np.float32 if width == 4 else np.float64)
        return [("nan" if vals[i] != vals[i] else repr(float(vals[i]))) if ok[i] else None for i in range(n)]
    if t == T_INT:
        dt = np.dtype("%s%d" % ("i" if field["is_signed"] else "u", width))
        return _fixed(data, ok, dt)
    if t == T_DECIMAL and width == 16:
        lo = data.view(np.uint64)[0::2]
        hi = data.view(np.int64)[1::2]
        return [(int(hi[i]) << 64) + int(lo[i]) if ok[i] else None for i in range(n)]
    if kind == po.K_DURATION:
        return [int(data.view(np.int64)[2 * i + 1]) if ok[i] else None for i in range(n)]
    if kind in (po.K_INTERVAL_MONTHS, po.K_INTERVAL_MDN):
        md, us = data.view(np.int32), data.view(np.int64)
        return [[int(md[4 * i]), int(md[4 * i + 1]), int(us[2 * i + 1])] if ok[i] else None for i in range(n)]
    dt = {2: np.int16, 4: np.int32, 8: np.int64}[width]
    return _fixed(data, ok, dt)


def canon_node(node, heap, heap_base=0):
    """A decoded node (flat or nested) -> canonical logical values for ALL its rows (batch level)."""
    f = node["field"] if "field" in node else None
    n = node["nrows"]
    kind = node["kind"]
    ok = po.valid_bits(node["validity"], n) if n else np.zeros(0, bool)
    if kind in (po.K_LIST32, po.K_LIST64):
        child = canon_node(node["children"][0], heap, heap_base)
        ent = node["data"].view(np.uint64).reshape(-1, 2) if n else np.zeros((0, 2), np.uint64)
        win, cwin = node["win"], node["children"][0]["win"]
        out = []
        k = 0
        is_map = f is not None and f["type"] == 17
        for r in range(n):
            while r >= win[k + 1]:
                k += 1
            if not ok[r]:
                out.append(None)
                continue
            start = cwin[k] + int(ent[r, 0])
            vals = child[start: start + int(ent[r, 1])]
            out.append([[v["key"], v["value"]] for v in vals] if is_map else vals)
        return out
    if kind == po.K_STRUCT:
        kids = [canon_node(c, heap, heap_base) for c in node["children"]]
        if f is not None and f["type"] == 16:  # fixed_size_list
            size = int(node["param"])
            return [kids[0][r * size: (r + 1) * size] if ok[r] else None for r in range(n)]
        names = [c["name"] for c in node["children"]]
        return [{nm: kid[r] for nm, kid in zip(names, kids)} if ok[r] else None for r in range(n)]
    if kind == po.K_STRVIEW:
        as_bytes = f["type"] == 23
        vals = po.strings_to_pylist(node["data"], node["validity"], n, heap, heap_base, as_bytes=as_bytes)
        return [("b:" + v.hex()) if (as_bytes and v is not None) else v for v in vals]
    if kind == po.K_DICT:
        d = node["dictionary"]
        vf = dict(f, has_dict=0)
        base = canon_flat(vf, d["kind"], d["param"], d["width"], d["data"], d["validity"], d["nrows"], heap, heap_base) + [None]
        sel = node["data"].view(np.uint32)
        return [base[int(sel[i])] for i in range(n)]
    return canon_flat(f, kind, node["param"], node["width"], node["data"], node["validity"], n, heap, heap_base)


def canon_oracle_column(field, col, n, heap, heap_base=0):
    """A column node from pyoracle.decode_stream (or the GPU equivalent) -> canonical logical list."""
    if "field" not in col:
        col = dict(col, field=field)
    if "nrows" not in col:
        col = dict(col, nrows=n)
    return canon_node(col, heap, heap_base)


def canon_stream(fields, batches, heap, heap_base=0):
    """-> {column name: canonical list over all batches}"""
    out = {f["name"]: [] for f in fields}
    by_name = {f["name"]: f for f in fields}
    for b in batches:
        for c in b["columns"]:
            out[c["name"]].extend(canon_oracle_column(by_name[c["name"]], c, b["nrows"], heap, heap_base))
    return out


# ------------------------------------------------------------------------------------------ pyarrow as the value oracle
def pyarrow_value(t, v):
    """A pyarrow python value in the form the package's host mirror returns (stored integers for DATE / DECIMAL, tuples for
    map entries): pyarrow is the oracle of the reference's own python tests (test/python/test_integration.py:32-61)."""
    import datetime
    import pyarrow as pa
    if v is None:
        return None
    if pa.types.is_dictionary(t):
        return pyarrow_value(t.value_type, v)
    if pa.types.is_date32(t):
        return (v - datetime.date(1970, 1, 1)).days if not isinstance(v, int) else v
    if pa.types.is_decimal(t):
        return int(v.scaleb(t.scale).to_integral_value())
    if pa.types.is_floating(t):
        return "nan" if v != v else float(v)
    if pa.types.is_map(t):
        return [(k, pyarrow_value(t.item_type, x)) for k, x in v]
    if pa.types.is_fixed_size_list(t) or pa.types.is_list(t) or pa.types.is_large_list(t):
        return [pyarrow_value(t.value_type, x) for x in v]
    if pa.types.is_struct(t):
        return {t.field(i).name: pyarrow_value(t.field(i).type, v[t.field(i).name]) for i in range(t.num_fields)}
    return v


def pyarrow_columns(table):
    """Every column of a pyarrow table as canonical values (canon_python form)."""
    import pyarrow as pa
    out = []
    for i, f in enumerate(table.schema):
        col, t = table.column(i), f.type
        if pa.types.is_timestamp(t):   # stored int64 in DuckDB's unit: microseconds (nanoseconds stay TIMESTAMP_NS)
            if t.unit in ("s", "ms"):
                col = col.cast(pa.timestamp("us", tz=t.tz))
            col, t = col.cast(pa.int64()), pa.int64()
        out.append(canon_python([pyarrow_value(t, v) for v in col.to_pylist()]))
    return out


# ------------------------------------------------------------------------------------------ raw (-1) compressed buffers
def rewrite_buffers_raw(stream_bytes, pick):
    """A compressed IPC stream with some of its buffers stored RAW: length prefix -1 followed by the uncompressed bytes, what
    Arrow C++ (IpcWriteOptions::min_space_savings), arrow-rs and Arrow Java write for incompressible buffers.  pyarrow's
    Python writer never emits them, so the stream is rewritten here: bodies re-laid out, RecordBatch.buffers and
    Message.bodyLength patched in place in the flatbuffer (same metadata size).  pick(batch_index, buffer_index, length)
    chooses the buffers.  Returns the new stream as bytes."""
    import struct
    import pyarrow as pa
    a = np.frombuffer(stream_bytes, dtype=np.uint8)
    out = bytearray()
    at = 0
    bi = 0
    for m in po.walk_stream(a):
        out += a[at: m["prefix_off"]].tobytes()
        head = bytearray(a[m["prefix_off"]: m["body_off"]].tobytes())
        body = a[m["body_off"]: m["body_off"] + m["body_len"]].tobytes()
        at = m["body_off"] + m["body_len"]
        if m["type"] != po.MSG_RECORD_BATCH or m["body_len"] == 0:
            out += head + body
            continue
        rb = po.decode_record_batch(a[m["meta_off"]: m["meta_off"] + m["meta_len"]])
        assert rb["compression"] in (0, 1)
        codec = pa.Codec("lz4" if rb["compression"] == 0 else "zstd")
        new_body, new_bufs = bytearray(), []
        for k, (off, ln) in enumerate(rb["buffers"]):
            piece = body[off: off + ln]
            if ln > 8 and pick(bi, k, ln):
                (ulen,) = struct.unpack("<q", piece[:8])
                if ulen != -1:
                    piece = struct.pack("<q", -1) + codec.decompress(piece[8:], decompressed_size=ulen).to_pybytes()
            new_bufs.append((len(new_body), len(piece)))
            new_body += piece + b"\0" * ((-len(piece)) % 8)
        old_vec = b"".join(struct.pack("<qq", o, l) for o, l in rb["buffers"])
        new_vec = b"".join(struct.pack("<qq", o, l) for o, l in new_bufs)
        where = bytes(head).find(old_vec)
        assert where >= 0 and bytes(head).find(old_vec, where + 1) < 0, "RecordBatch.buffers not found exactly once"
        head[where: where + len(old_vec)] = new_vec
        old_len = struct.pack("<q", m["body_len"])
        hits = [i for i in range(0, len(head) - 7) if bytes(head[i: i + 8]) == old_len and not (where <= i < where + len(old_vec))]
        assert len(hits) == 1, "Message.bodyLength not found exactly once"
        head[hits[0]: hits[0] + 8] = struct.pack("<q", len(new_body))
        out += head + new_body
        bi += 1
    out += a[at:].tobytes()
    return bytes(out)
