"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY (see oracle/oracle.h): imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

VECTOR_SIZE = 2048

# kinds (oracle.h)
K_COPY, K_BOOL, K_DEC128, K_DATE64, K_MUL_I32, K_MUL_I64, K_DIV_I64, K_STR32, K_STR64, K_DICT, K_FIXED_BINARY, \
    K_DURATION, K_INTERVAL_MONTHS, K_INTERVAL_MDN, K_NARROW, K_HALF_FLOAT, K_NULL = range(1, 18)
MSG_SCHEMA, MSG_DICTIONARY_BATCH, MSG_RECORD_BATCH = 1, 2, 3


def build():
    """(Re)build liboracle.so with gcc; cheap, so always delegated to make."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


class Msg(C.Structure):
    _fields_ = [("type", C.c_int32), ("meta_len", C.c_int32), ("prefix_off", C.c_int64), ("meta_off", C.c_int64),
                ("body_off", C.c_int64), ("body_len", C.c_int64)]


class Field(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("tz", C.c_char * 64), ("type", C.c_int32), ("bit_width", C.c_int32),
                ("is_signed", C.c_int32), ("precision", C.c_int32), ("scale", C.c_int32), ("unit", C.c_int32),
                ("byte_width", C.c_int32), ("nullable", C.c_int32), ("n_children", C.c_int32),
                ("has_dict", C.c_int32), ("dict_id", C.c_int64), ("dict_index_bit_width", C.c_int32),
                ("dict_index_signed", C.c_int32)]


class Node(C.Structure):
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64)]


class Buf(C.Structure):
    _fields_ = [("offset", C.c_int64), ("length", C.c_int64)]


class ColTask(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("param", C.c_int64), ("param2", C.c_int64),
                ("validity", C.c_void_p), ("buf1", C.c_void_p), ("buf2", C.c_void_p), ("buf2_len", C.c_int64),
                ("null_count", C.c_int64), ("nrows", C.c_int64), ("ptr_base", C.c_uint64),
                ("out_data", C.c_void_p), ("out_validity", C.c_void_p)]


class ScanStats(C.Structure):
    _fields_ = [("rows", C.c_int64), ("batches", C.c_int64), ("bytes_in", C.c_int64), ("bytes_out", C.c_int64),
                ("checksum", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_out_width.restype = C.c_int32
        _lib.orc_out_width.argtypes = [C.c_int32, C.c_int64]
        P, I64, I32, U64 = C.c_void_p, C.c_int64, C.c_int32, C.c_uint64
        _lib.orc_filter_range_i32.restype = C.c_int64
        _lib.orc_filter_range_i32.argtypes = [P, P, I64, I32, I32, P]
        _lib.orc_filter_range_i64.restype = C.c_int64
        _lib.orc_filter_range_i64.argtypes = [P, P, I64, I64, I64, P]
        _lib.orc_validity.restype = None
        _lib.orc_validity.argtypes = [P, I64, I64, I64, P]
        _lib.orc_validate_offsets32.argtypes = [P, I64, I64]
        _lib.orc_validate_offsets64.argtypes = [P, I64, I64]
        _lib.orc_string64.argtypes = [P, P, P, I64, I64, U64, P]
        _lib.orc_string32.argtypes = [P, P, P, I64, I64, U64, P]
        _lib.orc_mul_i64.argtypes = [P, P, I64, I64, I64, P]
        _lib.orc_mul_i32_to_i64.argtypes = [P, P, I64, I64, I64, P]
        _lib.orc_enc_validity.restype = None
        _lib.orc_enc_validity.argtypes = [P, I64, I64, P, P]
        _lib.orc_enc_decimal_widen.restype = None
        _lib.orc_enc_decimal_widen.argtypes = [P, I32, I64, P]
        _lib.orc_enc_bool.restype = None
        _lib.orc_enc_bool.argtypes = [P, P, I64, I64, P]
        _lib.orc_enc_varchar32.argtypes = [P, P, I64, I64, U64, P, P, P]
    return _lib


def _u8(buf):
    """bytes / memoryview / ndarray -> contiguous uint8 ndarray view (no copy when possible)."""
    if isinstance(buf, np.ndarray):
        return np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    return np.frombuffer(buf, dtype=np.uint8)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ------------------------------------------------------------------------------------------ framing
def walk_stream(buf, max_msgs=1 << 16):
    a = _u8(buf)
    msgs = (Msg * max_msgs)()
    n = C.c_int32(0)
    err = C.create_string_buffer(256)
    rc = lib().orc_walk_stream(_ptr(a), C.c_int64(a.size), msgs, C.c_int32(max_msgs), C.byref(n), err, 256)
    if rc:
        raise IOError(err.value.decode())
    return [dict(type=m.type, meta_len=m.meta_len, prefix_off=m.prefix_off, meta_off=m.meta_off,
                 body_off=m.body_off, body_len=m.body_len) for m in msgs[: n.value]]


def decode_schema(meta):
    a = _u8(meta)
    fields = (Field * 512)()
    n, ntop, endian = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    rc = lib().orc_decode_schema(_ptr(a), C.c_int32(a.size), fields, 512, C.byref(n), C.byref(ntop), C.byref(endian))
    if rc:
        raise ValueError("orc_decode_schema rc=%d" % rc)
    out = []
    for f in fields[: n.value]:
        out.append(dict(name=f.name.decode(), tz=f.tz.decode(), type=f.type, bit_width=f.bit_width,
                        is_signed=f.is_signed, precision=f.precision, scale=f.scale, unit=f.unit,
                        byte_width=f.byte_width, nullable=f.nullable, n_children=f.n_children,
                        has_dict=f.has_dict, dict_id=f.dict_id, dict_index_bit_width=f.dict_index_bit_width,
                        dict_index_signed=f.dict_index_signed, _c=f))
    return out, ntop.value, endian.value


def decode_record_batch(meta):
    a = _u8(meta)
    nodes = (Node * 512)()
    bufs = (Buf * 2048)()
    length, dict_id = C.c_int64(0), C.c_int64(0)
    nn, nb, comp, delta = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    rc = lib().orc_decode_record_batch(_ptr(a), C.c_int32(a.size), C.byref(length), nodes, 512, C.byref(nn), bufs,
                                       2048, C.byref(nb), C.byref(comp), C.byref(dict_id), C.byref(delta))
    if rc:
        raise ValueError("orc_decode_record_batch rc=%d" % rc)
    return dict(length=length.value, nodes=[(x.length, x.null_count) for x in nodes[: nn.value]],
                buffers=[(x.offset, x.length) for x in bufs[: nb.value]], compression=comp.value,
                dict_id=dict_id.value, is_delta=delta.value)


def decode_footer(file_bytes):
    a = _u8(file_bytes)
    blocks = (C.c_int64 * (3 * 65536))()
    n, nd = C.c_int32(0), C.c_int32(0)
    rc = lib().orc_decode_footer(_ptr(a), C.c_int64(a.size), blocks, 65536, C.byref(n), C.byref(nd))
    if rc:
        raise ValueError("orc_decode_footer rc=%d" % rc)
    return [(blocks[3 * i], blocks[3 * i + 1], blocks[3 * i + 2]) for i in range(n.value)], nd.value


def plan_column(field):
    kind, nbuf, param = C.c_int32(0), C.c_int32(0), C.c_int64(0)
    rc = lib().orc_plan_column(C.byref(field["_c"]), C.byref(kind), C.byref(param), C.byref(nbuf))
    if rc:
        raise NotImplementedError("column %r: arrow type %d not supported (rc=%d)" % (field["name"], field["type"], rc))
    return kind.value, param.value, nbuf.value


def out_width(kind, param):
    return lib().orc_out_width(kind, C.c_int64(param))


# ------------------------------------------------------------------------------------------ decode
def decode_column(kind, param, nrows, validity, buf1, buf2=None, null_count=-1, ptr_base=0, param2=0,
                  copy_direct=True):
    """Runs the 2048-row window loop over one column of one batch.
    Returns (data uint8[nrows*w], validity uint64[ceil(nrows/64)], rc)."""
    w = out_width(kind, param)
    out = np.zeros(max(nrows * w, 1), dtype=np.uint8)
    val = np.zeros(max((nrows + 63) // 64, 1), dtype=np.uint64)
    v = _u8(validity) if validity is not None and len(validity) else None
    b1 = _u8(buf1) if buf1 is not None and len(buf1) else np.zeros(16, np.uint8)
    b2 = _u8(buf2) if buf2 is not None and len(buf2) else (np.zeros(16, np.uint8) if buf2 is not None else None)
    t = ColTask(kind=kind, param=param, param2=param2, validity=_ptr(v), buf1=_ptr(b1), buf2=_ptr(b2),
                buf2_len=(len(buf2) if buf2 is not None else 0), null_count=null_count, nrows=nrows,
                ptr_base=ptr_base, out_data=_ptr(out), out_validity=_ptr(val))
    rc = lib().orc_decode_column(C.byref(t), C.c_int32(1 if copy_direct else 0))
    return out[: nrows * w], val[: (nrows + 63) // 64], rc


def _slice_column(body, bl, nbuf):
    if nbuf == 0:
        return None, np.zeros(16, np.uint8), None
    v = body[bl[0][0]: bl[0][0] + bl[0][1]] if bl[0][1] else None
    b1 = body[bl[1][0]: bl[1][0] + bl[1][1]]
    b2 = body[bl[2][0]: bl[2][0] + bl[2][1]] if nbuf > 2 else None
    return v, b1, b2


def decode_stream(buf, ptr_base_of=None, columns=None):
    """Decode every RecordBatch of a flat-schema IPC stream with the oracle.

    Returns (fields, batches); batches[i] = dict(nrows, columns=[dict(kind,param,width,data,validity,rc,...)]).
    String pointers are `ptr_base + offset`, ptr_base = ptr_base_of(batch_index, body_off, buf_off) when given,
    else the absolute position of the data buffer inside `buf` (so `buf` itself is the heap).
    DictionaryBatch messages (which the reference rejects, base_stream_reader.cpp:86-96) are decoded with the value
    type's plan and attached to the columns that use them as `dictionary`."""
    a = _u8(buf)
    msgs = walk_stream(a)
    if not msgs or msgs[0]["type"] != MSG_SCHEMA:
        raise IOError("Expected Schema Arrow IPC message but got end of stream")
    fields, ntop, _ = decode_schema(a[msgs[0]["meta_off"]: msgs[0]["meta_off"] + msgs[0]["meta_len"]])
    if len(fields) != ntop:
        raise NotImplementedError("nested schema")
    plans = [plan_column(f) for f in fields]
    dicts = {}
    batches = []
    bi = 0
    for m in msgs[1:]:
        rb = decode_record_batch(a[m["meta_off"]: m["meta_off"] + m["meta_len"]])
        body = a[m["body_off"]: m["body_off"] + m["body_len"]]
        if m["type"] == MSG_DICTIONARY_BATCH:
            f = [x for x in fields if x["has_dict"] and x["dict_id"] == rb["dict_id"]][0]
            vf = Field.from_buffer_copy(f["_c"])
            vf.has_dict = 0
            kind, param, nbuf = plan_column(dict(f, _c=vf))
            bl = rb["buffers"][:nbuf]
            v, b1, b2 = _slice_column(body, bl, nbuf)
            nrows, null_count = rb["nodes"][0]
            base = m["body_off"] + (bl[2][0] if nbuf > 2 else bl[1][0])
            d, val, rc = decode_column(kind, param, nrows, v, b1, b2, null_count, base)
            dicts[rb["dict_id"]] = dict(kind=kind, param=param, width=out_width(kind, param), data=d, validity=val,
                                        rc=rc, nrows=nrows, ptr_base=base, is_delta=rb["is_delta"])
            continue
        if m["type"] != MSG_RECORD_BATCH:
            raise IOError("Expected RecordBatch Arrow IPC message but got type %d" % m["type"])
        cols = []
        k = 0
        for ci, (f, (kind, param, nbuf)) in enumerate(zip(fields, plans)):
            bl = rb["buffers"][k: k + nbuf]
            k += nbuf
            if columns is not None and f["name"] not in columns:
                continue
            nrows, null_count = rb["nodes"][ci]
            v, b1, b2 = _slice_column(body, bl, nbuf)
            boff = bl[2][0] if nbuf > 2 else (bl[1][0] if nbuf else 0)
            base = ptr_base_of(bi, m["body_off"], boff) if ptr_base_of else m["body_off"] + boff
            dictionary = dicts.get(f["dict_id"]) if f["has_dict"] else None
            d, val, rc = decode_column(kind, param, nrows, v, b1, b2, null_count, base,
                                       param2=(dictionary["nrows"] if dictionary else 0))
            cols.append(dict(name=f["name"], kind=kind, param=param, width=out_width(kind, param), data=d,
                             validity=val, rc=rc, buffers=bl, ptr_base=base, null_count=null_count,
                             dictionary=dictionary))
        batches.append(dict(nrows=rb["length"], columns=cols, body_off=m["body_off"], body_len=m["body_len"]))
        bi += 1
    return fields, batches


def scan_stream(buf, max_batches=1 << 30, want_checksum=False):
    """The timed CPU baseline: body copy + FULL validation + 2048-row pull loop (oracle_scan.c)."""
    a = _u8(buf)
    st = ScanStats()
    rc = lib().orc_scan_stream(_ptr(a), C.c_int64(a.size), C.c_int32(min(max_batches, (1 << 31) - 1)),
                               C.c_int32(1 if want_checksum else 0), C.byref(st))
    return rc, dict(rows=st.rows, batches=st.batches, bytes_in=st.bytes_in, bytes_out=st.bytes_out,
                    checksum=st.checksum)


# ------------------------------------------------------------------------------------------ logical views
def valid_bits(validity_words, n):
    bits = np.unpackbits(validity_words.view(np.uint8), bitorder="little")[:n]
    return bits.astype(bool)


def strings_to_pylist(data16, validity_words, n, heap, ptr_base, as_bytes=False):
    """DuckDB string_t[n] -> python list (None for NULL); long strings are resolved through `heap`."""
    s = data16.reshape(-1, 16)
    lens = s[:, 0:4].copy().view(np.uint32).reshape(-1)
    ptrs = s[:, 8:16].copy().view(np.uint64).reshape(-1)
    ok = valid_bits(validity_words, n)
    heap = _u8(heap)
    out = []
    for i in range(n):
        if not ok[i]:
            out.append(None)
            continue
        ln = int(lens[i])
        if ln <= 12:
            b = s[i, 4: 4 + ln].tobytes()
        else:
            o = int(ptrs[i]) - ptr_base
            b = heap[o: o + ln].tobytes()
            assert b[:4] == s[i, 4:8].tobytes(), "string_t prefix mismatch"
        out.append(b if as_bytes else b.decode("utf-8"))
    return out


def fixed_to_pylist(data, validity_words, n, dtype):
    vals = data.view(dtype)[:n]
    ok = valid_bits(validity_words, n)
    return [vals[i].item() if ok[i] else None for i in range(n)]
