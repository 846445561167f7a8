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
    K_DURATION, K_INTERVAL_MONTHS, K_INTERVAL_MDN, K_NARROW, K_HALF_FLOAT, K_NULL, K_STRVIEW, K_LIST32, K_LIST64, \
    K_STRUCT = range(1, 22)
T_STRUCT, T_FIXED_LIST, T_LIST, T_LARGE_LIST, T_MAP, T_UTF8_VIEW, T_BINARY_VIEW, T_NULL, T_UNION = 13, 16, 12, 21, 17, 24, 23, 1, 14
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


class EncodeStats(C.Structure):
    _fields_ = [("rows", C.c_int64), ("batches", C.c_int64), ("bytes_in", C.c_int64), ("bytes_out", C.c_int64),
                ("checksum", C.c_uint64), ("seconds", C.c_double), ("mismatches", C.c_int64)]


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
        _lib.orc_filter_cnf.restype = C.c_int64
        _lib.orc_filter_cnf.argtypes = [P, I32, I64, P]
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
        _lib.orc_list_entries.argtypes = [P, I32, I64, I64, I64, I64, P]
        _lib.orc_convert_column.argtypes = [P, P]
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
    var = (C.c_int64 * 512)()
    nv = C.c_int32(0)
    lib().orc_decode_variadic_counts(_ptr(a), C.c_int32(a.size), var, 512, C.byref(nv))
    return dict(length=length.value, nodes=[(x.length, x.null_count) for x in nodes[: nn.value]],
                buffers=[(x.offset, x.length) for x in bufs[: nb.value]], compression=comp.value,
                dict_id=dict_id.value, is_delta=delta.value, variadic=list(var[: nv.value]))


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


def _and_parent(words, n, parent_words, div):
    """NULLs of a struct (div 1) / fixed_size_list (div = list size) parent propagate into the child."""
    bits = np.unpackbits(words.view(np.uint8), bitorder="little")
    pbits = np.unpackbits(parent_words.view(np.uint8), bitorder="little")
    idx = np.arange(n) // max(div, 1)
    bits[:n] &= pbits[idx]
    return np.packbits(bits, bitorder="little").view(np.uint64).copy()


def _decode_node(a, fields, cur, rb, m, ptr_base_fn, win, parent_valid=None, parent_div=1, value_only=False, keep=True):
    """Depth-first decode of field `cur['field']` (index into the flattened schema) and its descendants.
    win = row numbers (in this node's row space) where the top-level 2048-row chunk windows start, plus the end."""
    f = fields[cur["field"]]
    cur["field"] += 1
    nrows, null_count = rb["nodes"][cur["node"]]
    cur["node"] += 1
    dict_encoded = f["has_dict"] and not value_only
    t = f["type"]
    if dict_encoded:
        own = 2
    elif t == T_NULL:
        own = 0
    elif t in (T_STRUCT, T_FIXED_LIST):
        own = 1
    elif t in (5, 4, 20, 19):
        own = 3
    elif t in (T_UTF8_VIEW, T_BINARY_VIEW):
        own = 2 + rb["variadic"][cur["variadic"]]
        cur["variadic"] += 1
    else:
        own = 2
    spans = rb["buffers"][cur["buf"]: cur["buf"] + own]
    cur["buf"] += own
    body = a[m["body_off"]: m["body_off"] + m["body_len"]]
    sl = lambda sp: body[sp[0]: sp[0] + sp[1]]
    if value_only:
        vf = Field.from_buffer_copy(f["_c"])
        vf.has_dict = 0
        kind, param, _ = plan_column(dict(f, _c=vf))
    else:
        kind, param, _ = plan_column(f)
    w = out_width(kind, param)
    node = dict(name=f["name"], kind=kind, param=param, width=w, nrows=nrows, null_count=null_count, rc=0, win=list(win),
                buffers=spans, children=[], field=f)
    # validity: own bitmap (null_count == 0 => all valid), then what propagates from the parent
    words = np.zeros(max((nrows + 63) // 64, 1), np.uint64)
    bm = _u8(sl(spans[0])) if own > 0 and spans[0][1] else None
    if kind == K_NULL:
        words[:] = 0
    else:
        lib().orc_validity(_ptr(bm), null_count, 0, nrows, _ptr(words))
    if parent_valid is not None and nrows:
        words = _and_parent(words, nrows, parent_valid, parent_div)
        if nrows & 63:
            words[(nrows - 1) >> 6] |= np.uint64(0xFFFFFFFFFFFFFFFF) << np.uint64(nrows & 63)
    words = words[: (nrows + 63) // 64]
    node["validity"] = words
    out = np.zeros(max(nrows * w, 1), np.uint8)
    vptr = _ptr(words) if len(words) else None
    if kind in (K_LIST32, K_LIST64):
        offw = 4 if kind == K_LIST32 else 8
        offs = _u8(sl(spans[1])).view(np.int32 if offw == 4 else np.int64)
        child_len = rb["nodes"][cur["node"]][0] if f["n_children"] else 0
        for k in range(len(win) - 1):
            r0, r1 = win[k], win[k + 1]
            if r1 > r0:
                rc = lib().orc_list_entries(_ptr(offs), offw, r0, r1 - r0, r0, child_len, out[16 * r0:].ctypes.data)
                node["rc"] |= rc
        child_win = [int(offs[r]) if nrows else 0 for r in win] if nrows else [0] * len(win)
        node["data"] = out[: nrows * 16]
        node["children"].append(_decode_node(a, fields, cur, rb, m, ptr_base_fn, child_win))
        return node
    if kind == K_STRUCT:
        node["data"] = out[:0]
        size = param if t == T_FIXED_LIST else 1
        child_win = [r * size for r in win] if t == T_FIXED_LIST else list(win)
        for _ in range(f["n_children"]):
            node["children"].append(_decode_node(a, fields, cur, rb, m, ptr_base_fn, child_win, parent_valid=words,
                                                 parent_div=size))
        return node
    task = ColTask(kind=kind, param=param, nrows=nrows, null_count=null_count, out_data=_ptr(out))
    keep_alive = []
    if kind == K_STRVIEW:
        table = np.zeros(max(2 * (own - 2), 2), np.uint64)
        for j, sp in enumerate(spans[2:]):
            table[2 * j] = ptr_base_fn(m["body_off"], sp[0])
            table[2 * j + 1] = sp[1]
        b1 = _u8(sl(spans[1])) if spans[1][1] else np.zeros(16, np.uint8)
        task.buf1, task.buf2, task.buf2_len = _ptr(b1), _ptr(table), own - 2
        keep_alive += [b1, table]
        node["ptr_base"] = 0
    elif own >= 2:
        b1 = _u8(sl(spans[1])) if spans[1][1] else np.zeros(16, np.uint8)
        task.buf1 = _ptr(b1)
        keep_alive.append(b1)
        if own == 3:
            b2 = _u8(sl(spans[2])) if spans[2][1] else np.zeros(16, np.uint8)
            task.buf2, task.buf2_len = _ptr(b2), spans[2][1]
            keep_alive.append(b2)
        data_span = spans[2] if own == 3 else spans[1]
        task.ptr_base = ptr_base_fn(m["body_off"], data_span[0])
        node["ptr_base"] = task.ptr_base
    if kind == K_DICT:
        task.param2 = cur["dicts"][f["dict_id"]]["nrows"]
        node["dictionary"] = cur["dicts"][f["dict_id"]]
    if nrows:
        node["rc"] |= lib().orc_convert_column(C.byref(task), vptr)
    node["data"] = out[: nrows * w]
    return node


def _windows(n):
    w = list(range(0, n, VECTOR_SIZE)) + [n]
    return w if n else [0, 0]


def _flatten_fields(fields):
    """index of every top-level field inside the flattened (depth-first) field list"""
    tops, i = [], 0

    def skip(j):
        nc = fields[j]["n_children"]
        j += 1
        for _ in range(nc):
            j = skip(j)
        return j

    while i < len(fields):
        tops.append(i)
        i = skip(i)
    return tops


def decode_stream(buf, ptr_base_of=None, columns=None):
    """Decode every RecordBatch of an IPC stream with the oracle (flat and nested columns, string views, dictionaries).

    Returns (top-level fields, batches); batches[i] = dict(nrows, columns=[node...]) where a node is
    dict(name, kind, param, width, data, validity, rc, children=[...], win=[...]).  String pointers are
    `ptr_base + offset`: ptr_base_of(batch_index, body_off, buf_off) when given, else the absolute position of the data
    buffer inside `buf` (so `buf` itself is the heap).  DictionaryBatch messages (which the reference rejects,
    base_stream_reader.cpp:86-96) are decoded with the value type's plan and attached as `dictionary`."""
    a = _u8(buf)
    msgs = walk_stream(a)
    if not msgs or msgs[0]["type"] != MSG_SCHEMA:
        raise IOError("Expected Schema Arrow IPC message but got end of stream")
    fields, ntop, _ = decode_schema(a[msgs[0]["meta_off"]: msgs[0]["meta_off"] + msgs[0]["meta_len"]])
    tops = _flatten_fields(fields)
    top_fields = [fields[i] for i in tops]
    dicts = {}
    batches = []
    bi = 0
    for m in msgs[1:]:
        rb = decode_record_batch(a[m["meta_off"]: m["meta_off"] + m["meta_len"]])
        if m["type"] not in (MSG_DICTIONARY_BATCH, MSG_RECORD_BATCH):
            raise IOError("Expected RecordBatch Arrow IPC message but got type %d" % m["type"])
        base_fn = (lambda body_off, boff, _bi=bi: ptr_base_of(_bi, body_off, boff)) if ptr_base_of else (lambda body_off, boff: body_off + boff)
        if m["type"] == MSG_DICTIONARY_BATCH:
            fi = [i for i in tops if fields[i]["has_dict"] and fields[i]["dict_id"] == rb["dict_id"]][0]
            cur = dict(field=fi, node=0, buf=0, variadic=0, dicts=dicts)
            n = rb["nodes"][0][0]
            dicts[rb["dict_id"]] = _decode_node(a, fields, cur, rb, m, (lambda body_off, boff: body_off + boff), _windows(n),
                                                value_only=True)
            dicts[rb["dict_id"]]["is_delta"] = rb["is_delta"]
            continue
        cols = []
        cur = dict(field=0, node=0, buf=0, variadic=0, dicts=dicts)
        for ti, fi in enumerate(tops):
            assert cur["field"] == fi
            node = _decode_node(a, fields, cur, rb, m, base_fn, _windows(rb["length"]))
            if columns is None or fields[fi]["name"] in columns:
                cols.append(node)
        batches.append(dict(nrows=rb["length"], columns=cols, body_off=m["body_off"], body_len=m["body_len"]))
        bi += 1
    return top_fields, batches


def scan_stream(buf, max_batches=1 << 30, want_checksum=False):
    """The timed CPU baseline: body copy + FULL validation + 2048-row pull loop (oracle_scan.c)."""
    a = _u8(buf)
    st = ScanStats()
    rc = lib().orc_scan_stream(_ptr(a), C.c_int64(a.size), C.c_int32(min(max_batches, (1 << 31) - 1)),
                               C.c_int32(1 if want_checksum else 0), C.byref(st))
    return rc, dict(rows=st.rows, batches=st.batches, bytes_in=st.bytes_in, bytes_out=st.bytes_out,
                    checksum=st.checksum)


def encode_stream(buf, max_batches=1 << 30, verify=False):
    """The timed CPU baseline of the COPY TO direction (oracle_scan.c orc_encode_stream): K7 + the reference's two extra
    copies per record batch, single threaded; `seconds` covers the encode alone."""
    a = _u8(buf)
    st = EncodeStats()
    lib().orc_encode_stream.restype = C.c_int
    rc = lib().orc_encode_stream(_ptr(a), C.c_int64(a.size), C.c_int32(min(max_batches, (1 << 31) - 1)), C.c_int32(1 if verify else 0),
                                 C.byref(st))
    return rc, dict(rows=st.rows, batches=st.batches, bytes_in=st.bytes_in, bytes_out=st.bytes_out, checksum=st.checksum,
                    seconds=st.seconds, mismatches=st.mismatches)


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


class FilterLeaf(C.Structure):
    _fields_ = [("data", C.c_void_p), ("validity", C.c_void_p), ("values", C.c_void_p), ("value", C.c_int64),
                ("op", C.c_int32), ("width", C.c_int32), ("is_unsigned", C.c_int32), ("n_values", C.c_int32),
                ("ends_clause", C.c_int32), ("_pad", C.c_int32)]


_FILTER_OPS = {"=": 1, "<>": 2, "!=": 2, "<": 3, "<=": 4, ">": 5, ">=": 6, "is null": 7, "is not null": 8, "in": 9}


def filter_cnf(clauses, columns, n):
    """The rows [0, n) that pass AND-of-ORs `clauses` = [[(column, op, value), ...], ...] over `columns` =
    {name: (data ndarray of the stored integers, validity uint64 words or None)} -> ascending row indices (orc_filter_cnf).
    A column may also be a python list of bytes / str / None (VARCHAR, BLOB): then the whole predicate is evaluated here,
    row by row, with SQL's rules (a comparison with NULL is not true; bytes compare byte-wise, a proper prefix first)."""
    if any(isinstance(columns[leaf[0]], list) or isinstance(columns[leaf[0]][0], list) for clause in clauses for leaf in clause):
        def value(col, i):
            c = columns[col]
            if isinstance(c, list):
                v = c[i]
                return v.encode() if isinstance(v, str) else v
            data, valid = c
            if isinstance(data, list):
                v = data[i]
                return v.encode() if isinstance(v, str) else v
            if valid is not None and not ((int(valid[i >> 6]) >> (i & 63)) & 1):
                return None
            return int(data[i])
        norm = lambda c: c.encode() if isinstance(c, str) else c

        def leaf_true(leaf, i):
            v, op = value(leaf[0], i), leaf[1].lower()
            if op == "is null":
                return v is None
            if op == "is not null":
                return v is not None
            if v is None:
                return False
            if op == "in":
                return v in [norm(c) for c in leaf[2]]
            c = norm(leaf[2])
            if op == "starts_with":
                return v.startswith(c)
            return {"=": v == c, "==": v == c, "<>": v != c, "!=": v != c, "<": v < c, "<=": v <= c, ">": v > c, ">=": v >= c}[op]
        return np.array([i for i in range(n) if all(any(leaf_true(l, i) for l in clause) for clause in clauses)], np.uint32)
    leaves, keep = [], []
    for clause in clauses:
        for j, leaf in enumerate(clause):
            data, valid = columns[leaf[0]]
            data = np.ascontiguousarray(data)
            keep.append(data)
            L = FilterLeaf(data=data.ctypes.data, validity=valid.ctypes.data if valid is not None else None,
                           op=_FILTER_OPS[leaf[1].lower()], width=data.dtype.itemsize, is_unsigned=int(data.dtype.kind == "u"),
                           ends_clause=int(j == len(clause) - 1))
            if leaf[1].lower() == "in":
                vals = np.array(list(leaf[2]), np.int64)
                keep.append(vals)
                L.values, L.n_values = vals.ctypes.data, len(vals)
            elif len(leaf) > 2:
                L.value = int(leaf[2])
            leaves.append(L)
    arr = (FilterLeaf * max(len(leaves), 1))(*leaves)
    sel = np.zeros(max(n, 1), np.uint32)
    c = lib().orc_filter_cnf(arr, len(leaves), n, sel.ctypes.data)
    return sel[:c].copy()
