"""duckdb-arrow_amd -- MI355X-native Arrow IPC scan / encode path of the DuckDB `nanoarrow` extension.

Host-side mirror (Python flavour) of the reference's operator surface for this path, over the C ABI of
libmi_arrow_ipc.so (include/mi_arrow_ipc.h):

    read_arrow(...)        src/scanner/read_arrow.cpp:43-86            -> Connection.read_arrow
    scan_arrow_ipc(...)    src/scanner/scan_arrow_ipc.cpp:20-64        -> Connection.scan_arrow_ipc / from_arrow
    to_arrow_ipc(...)      src/writer/to_arrow_ipc.cpp:52-182          -> Connection.to_arrow_ipc
    COPY ... (FORMAT ARROWS)  src/writer/write_arrow_stream.cpp:54-272 -> Connection.copy_to
    nanoarrow_version()    src/nanoarrow_extension.cpp:20-31           -> nanoarrow_version()

The package directory name carries a hyphen (it is the name the build contract asks for); import it through the
`duckdb_arrow_amd` shim module at the repository root.  Nothing here falls back to the CPU: without the built
library or without a HIP device every scan / encode call raises.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from ._ffi import MiError, VECTOR_SIZE  # noqa: F401

__all__ = ["Connection", "Reader", "Context", "Plan", "Table", "MiError", "nanoarrow_version", "version", "build",
           "synth_lineitem_stream", "VECTOR_SIZE"]


def build(verbose=False):
    """Compiles libmi_arrow_ipc.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.run(["make", "-j8", "-C", os.path.join(here, "csrc")] + ([] if verbose else ["-s"]), check=True)


def version():
    return _ffi.lib().mi_version().decode()


def nanoarrow_version():
    """SELECT nanoarrow_version() (test/sql/nanoarrow.test:15-18)."""
    return _ffi.lib().mi_nanoarrow_version().decode()


def device_count():
    return _ffi.lib().mi_device_count()


def _field_dict(f):
    # names come from the file: a damaged stream may carry bytes that are not UTF-8
    return dict(name=f.name.decode("utf-8", "replace"), duck_type=f.duck_type.decode("utf-8", "replace"),
                format=f.format.decode("utf-8", "replace"), arrow_type=f.arrow_type,
                kind=f.kind, out_width=f.out_width, param=f.param, n_buffers=f.n_buffers, flat_index=f.flat_index,
                bit_width=f.bit_width, is_signed=f.is_signed, precision=f.precision, scale=f.scale, unit=f.unit,
                byte_width=f.byte_width, nullable=f.nullable, timezone=f.timezone.decode("utf-8", "replace"),
                has_dictionary=f.has_dictionary, dict_id=f.dict_id, dict_index_bit_width=f.dict_index_bit_width,
                dict_index_signed=f.dict_index_signed)


def _as_u8(buf):
    if isinstance(buf, np.ndarray):
        return np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    return np.frombuffer(buf, dtype=np.uint8)


# ---------------------------------------------------------------------------------------------------- host reader
class Reader:
    """IPCFileStreamReader / IPCBufferStreamReader (host only, no GPU needed)."""

    def __init__(self, path=None, buffers=None):
        self._h = C.c_void_p()
        self._keep = []
        L = _ffi.lib()
        if path is not None:
            _ffi.check(L.mi_reader_open_file(os.fsencode(path), C.byref(self._h)))
        else:
            arr = (_ffi.IpcBuffer * len(buffers))()
            for i, b in enumerate(buffers):
                if isinstance(b, tuple):
                    arr[i].ptr, arr[i].size = b
                else:
                    a = _as_u8(b)
                    self._keep.append(a)
                    arr[i].ptr, arr[i].size = a.ctypes.data, a.size
            _ffi.check(L.mi_reader_open_buffers(arr, len(buffers), C.byref(self._h)))

    def close(self):
        if self._h:
            _ffi.lib().mi_reader_close(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def schema(self):
        n = C.c_int32(0)
        _ffi.check(_ffi.lib().mi_reader_schema(self._h, None, 0, C.byref(n)))
        fields = (_ffi.Field * max(n.value, 1))()
        _ffi.check(_ffi.lib().mi_reader_schema(self._h, fields, n.value, C.byref(n)))
        return [_field_dict(f) for f in fields[: n.value]]

    def schema_metadata(self):
        cnt = C.c_int32(0)
        _ffi.check(_ffi.lib().mi_reader_schema_metadata(self._h, -1, None, None, None, None, C.byref(cnt)))
        out = []
        for i in range(cnt.value):
            k, kl, v, vl = C.c_char_p(), C.c_int32(), C.c_void_p(), C.c_int32()
            _ffi.check(_ffi.lib().mi_reader_schema_metadata(self._h, i, C.byref(k), C.byref(kl), C.byref(v), C.byref(vl),
                                                            C.byref(cnt)))
            out.append((k.value[: kl.value].decode(), C.string_at(v.value, vl.value)))
        return out

    def set_projection(self, names):
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        _ffi.check(_ffi.lib().mi_reader_set_projection(self._h, arr, len(names)))

    def next_batch(self, accept_dictionaries=False):
        b = _ffi.Batch()
        rc = _ffi.lib().mi_reader_next_batch(self._h, 1 if accept_dictionaries else 0, C.byref(b))
        if rc == _ffi.MI_ENODATA:
            return None
        _ffi.check(rc)
        nc = b.n_columns
        body = np.ctypeslib.as_array(C.cast(b.body, C.POINTER(C.c_uint8)), shape=(b.body_size,)) if b.body_size else \
            np.zeros(0, np.uint8)
        return dict(length=b.length, body=body, body_ptr=b.body or 0, body_size=b.body_size,
                    body_file_offset=b.body_file_offset, is_dictionary=bool(b.is_dictionary), dict_id=b.dict_id,
                    is_delta=bool(b.is_delta), compression=b.compression,
                    column_field=[b.column_field[i] for i in range(nc)],
                    null_count=[b.null_count[i] for i in range(nc)],
                    buffers=[(b.buffers[i].offset, b.buffers[i].length) for i in range(3 * nc)],
                    column_node=[b.column_node[i] for i in range(nc)],
                    nodes=[dict(name=b.nodes[i].name.decode("utf-8", "replace"), arrow_type=b.nodes[i].arrow_type, kind=b.nodes[i].kind,
                                out_width=b.nodes[i].out_width, parent=b.nodes[i].parent, depth=b.nodes[i].depth,
                                n_children=b.nodes[i].n_children, param=b.nodes[i].param, length=b.nodes[i].length,
                                null_count=b.nodes[i].null_count,
                                spans=[(b.node_spans[j].offset, b.node_spans[j].length)
                                       for j in range(b.nodes[i].first_span, b.nodes[i].first_span + b.nodes[i].n_spans)])
                           for i in range(b.n_nodes)])

    def export_stream(self, accept_dictionaries=False):
        """The reader (with its projection) as an Arrow C stream (mi_reader_export_stream), imported into pyarrow: what
        DuckDB's arrow scan would consume in place of the reference's IpcArrayStream.  The reader is consumed."""
        import pyarrow as pa
        stream = _ffi.ArrowArrayStream()
        _ffi.check(_ffi.lib().mi_reader_export_stream(self._h, 1 if accept_dictionaries else 0, C.byref(stream)))
        self._keep.append(stream)
        return pa.RecordBatchReader._import_from_c(C.addressof(stream))

    def index(self):
        ent = C.POINTER(_ffi.BatchIndexEntry)()
        n = C.c_int32(0)
        _ffi.check(_ffi.lib().mi_reader_index(self._h, C.byref(ent), C.byref(n)))
        return [dict(prefix_offset=ent[i].prefix_offset, meta_len=ent[i].meta_len, type=ent[i].type,
                     body_offset=ent[i].body_offset, body_len=ent[i].body_len, n_rows=ent[i].n_rows)
                for i in range(n.value)]

    def progress(self):
        return _ffi.lib().mi_reader_progress(self._h)


# ---------------------------------------------------------------------------------------------------- device
class Context:
    """mi_ctx: one per (GPU, worker).  Raises MiError(ENODEV) without a HIP device -- there is no CPU fallback."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _ffi.check(_ffi.lib().mi_ctx_create(device, C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            _ffi.lib().mi_ctx_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def numa(self):
        """(NUMA node of the GPU or -1, the node's CPUs as a set) -- mi_ctx_numa."""
        node = C.c_int32(-1)
        buf = C.create_string_buffer(4096)
        _ffi.check(_ffi.lib().mi_ctx_numa(self._h, C.byref(node), buf, len(buf)))
        cpus = set()
        for part in buf.value.decode().split(","):
            if part:
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        return node.value, cpus

    def bind_this_thread(self):
        """Pins the CALLING thread (threads it starts afterwards inherit) to the CPUs of the GPU's NUMA node, within what it may
        use: for the threads of a host program that write the files a scan reads or consume its chunks.  Returns the node, or
        -1 when nothing was bound."""
        import os
        node, cpus = self.numa()
        allowed = os.sched_getaffinity(0)
        if node < 0 or not (cpus & allowed):
            return -1
        os.sched_setaffinity(0, cpus & allowed)
        return node


def make_task(kind, nrows, buf1, out_data, *, validity=0, buf2=0, out_validity=0, out_aux=0, ptr_base=0, row_offset=0,
              buf2_len=0, param=0, param2=0, null_count=-1, depth=0, parent_div=0, sel=0, sel_count=0):
    """mi_col_task from raw device addresses (decode: out_aux = parent validity words, parent_div = rows per parent row;
    sel / sel_count: gather mode, only the selected rows are decoded, densely packed)."""
    return _ffi.ColTask(validity=validity or None, buf1=buf1 or None, buf2=buf2 or None, out_data=out_data or None,
                        out_validity=out_validity or None, out_aux=out_aux or None, ptr_base=ptr_base, nrows=nrows,
                        row_offset=row_offset, buf2_len=buf2_len, param=param, param2=param2, null_count=null_count,
                        kind=kind, flags=parent_div, depth=depth, sel=sel or None, sel_count=sel_count or None)


class Plan:
    """mi_plan: a device-resident task table; launch() = one kernel per kernel class for ALL tasks."""

    def __init__(self, ctx, tasks):
        self._h = C.c_void_p()
        self.ctx = ctx
        self.n_tasks = len(tasks)
        arr = (_ffi.ColTask * max(len(tasks), 1))(*tasks)
        _ffi.check(_ffi.lib().mi_plan_create(ctx._h, arr, len(tasks), C.byref(self._h)))

    def launch(self, stream=0):
        _ffi.check(_ffi.lib().mi_plan_launch(self._h, C.c_void_p(stream or None)))

    def status(self):
        bits = C.c_uint32(0)
        _ffi.check(_ffi.lib().mi_plan_status(self._h, C.byref(bits)))
        return bits.value

    def raise_for_status(self):
        bits = self.status()
        if bits:
            _ffi.check(_ffi.lib().mi_status_to_error(bits))
        return bits

    def stats(self):
        r, w, rows, tiles = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        _ffi.check(_ffi.lib().mi_plan_stats(self._h, C.byref(r), C.byref(w), C.byref(rows), C.byref(tiles)))
        return dict(bytes_read=r.value, bytes_written=w.value, rows=rows.value, tiles=tiles.value)

    def class_stats(self):
        """Per kernel class: algorithmic bytes read / written, rows, tiles and the kernel's name."""
        out = []
        for cls in range(_ffi.NUM_KERNEL_CLASSES):
            r, w, rows, tiles, name = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_char_p()
            _ffi.check(_ffi.lib().mi_plan_class_stats(self._h, cls, C.byref(r), C.byref(w), C.byref(rows), C.byref(tiles),
                                                      C.byref(name)))
            out.append(dict(kernel=name.value.decode(), bytes_read=r.value, bytes_written=w.value, rows=rows.value,
                            tiles=tiles.value))
        return out

    def launch_timed(self, stream=0):
        """One launch with HIP events around every kernel class; returns [ms per class] after synchronising."""
        ms = (C.c_float * _ffi.NUM_KERNEL_CLASSES)()
        _ffi.check(_ffi.lib().mi_plan_launch_timed(self._h, C.c_void_p(stream or None), ms))
        return list(ms)

    def null_counts(self):
        out = (C.c_int64 * max(self.n_tasks, 1))()
        _ffi.check(_ffi.lib().mi_plan_null_counts(self._h, out, self.n_tasks))
        return list(out[: self.n_tasks])

    def close(self):
        if self._h:
            _ffi.lib().mi_plan_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


def filter_range(ctx, values_ptr, width, validity_ptr, nrows, lo, hi, sel_ptr, count_ptr, stream=0):
    _ffi.check(_ffi.lib().mi_filter_range(ctx._h, values_ptr, width, validity_ptr or None, nrows, lo, hi, sel_ptr, count_ptr,
                                          C.c_void_p(stream or None)))


# ---------------------------------------------------------------------------------------------------- logical views
def _valid_bits(validity_ptr, n, shift=0):
    """`shift`: bit of the first word that belongs to row 0 (mi_vector.validity_shift, nested children only)."""
    if not validity_ptr:
        return np.ones(n, bool)
    words = np.ctypeslib.as_array(C.cast(validity_ptr, C.POINTER(C.c_uint64)), shape=((n + shift + 63) // 64 or 1,))
    return np.unpackbits(words.view(np.uint8), bitorder="little")[shift: shift + n].astype(bool)


def _string_values(data_ptr, n, ok, as_bytes):
    raw = np.ctypeslib.as_array(C.cast(data_ptr, C.POINTER(C.c_uint8)), shape=(n * 16,)).reshape(n, 16)
    lens = raw[:, 0:4].copy().view(np.uint32).reshape(-1)
    ptrs = raw[:, 8:16].copy().view(np.uint64).reshape(-1)
    out = []
    for i in range(n):
        if not ok[i]:
            out.append(None)
            continue
        ln = int(lens[i])
        b = raw[i, 4: 4 + ln].tobytes() if ln <= 12 else C.string_at(int(ptrs[i]), ln)
        out.append(b if as_bytes else b.decode("utf-8", "replace"))   # damaged payloads are not ours to reject
    return out


_INT_TYPES = {"TINYINT": np.int8, "UTINYINT": np.uint8, "SMALLINT": np.int16, "USMALLINT": np.uint16,
              "INTEGER": np.int32, "UINTEGER": np.uint32, "BIGINT": np.int64, "UBIGINT": np.uint64,
              "DATE": np.int32, "TIME": np.int64, "TIMESTAMP": np.int64, "TIMESTAMP_S": np.int64,
              "TIMESTAMP_MS": np.int64, "TIMESTAMP_NS": np.int64, "TIMESTAMP WITH TIME ZONE": np.int64}


def _flat_values(duck_type, width, data_ptr, validity_ptr, n, shift=0):
    """One host DuckDB flat vector -> python list of the raw stored values (ints for temporal / decimal types)."""
    ok = _valid_bits(validity_ptr, n, shift)
    if n == 0:
        return []
    if duck_type in ("VARCHAR", "BLOB"):
        return _string_values(data_ptr, n, ok, as_bytes=(duck_type == "BLOB"))
    raw = np.ctypeslib.as_array(C.cast(data_ptr, C.POINTER(C.c_uint8)), shape=(n * width,))
    if duck_type == "BOOLEAN":
        return [bool(raw[i]) if ok[i] else None for i in range(n)]
    if duck_type in ("FLOAT", "DOUBLE"):
        vals = raw.view(np.float32 if width == 4 else np.float64)
        return [float(vals[i]) if ok[i] else None for i in range(n)]
    if duck_type == '"NULL"':
        return [None] * n
    if duck_type == "INTERVAL":
        md, us = raw.view(np.int32), raw.view(np.int64)
        return [(int(md[4 * i]), int(md[4 * i + 1]), int(us[2 * i + 1])) if ok[i] else None for i in range(n)]
    if duck_type.startswith("DECIMAL") and width == 16:
        lo, hi = raw.view(np.uint64)[0::2], raw.view(np.int64)[1::2]
        return [(int(hi[i]) << 64) + int(lo[i]) if ok[i] else None for i in range(n)]
    if duck_type.startswith("DECIMAL"):
        vals = raw.view({2: np.int16, 4: np.int32, 8: np.int64}[width])
    else:
        vals = raw.view(_INT_TYPES[duck_type])
    return [vals[i].item() if ok[i] else None for i in range(n)]


def _split_top(text, sep=","):
    """Split on `sep` outside parentheses / quotes."""
    parts, depth, cur, quoted = [], 0, [], False
    for ch in text:
        if ch == '"':
            quoted = not quoted
        if not quoted:
            depth += ch == "("
            depth -= ch == ")"
            if ch == sep and depth == 0:
                parts.append("".join(cur).strip())
                cur = []
                continue
        cur.append(ch)
    if cur or parts:
        parts.append("".join(cur).strip())
    return parts


def parse_duck_type(text):
    """'STRUCT(a INTEGER, b VARCHAR)[]' -> ('list', ('struct', [('a', ('leaf', 'INTEGER')), ...])).  Mirrors the
    LogicalType the reference binds for the column (ArrowType::GetDuckType)."""
    text = text.strip()
    if text.endswith("]"):
        i = text.rindex("[")
        inner, size = parse_duck_type(text[:i]), text[i + 1: -1]
        return ("list", inner) if size == "" else ("array", inner, int(size))
    if text.startswith("STRUCT(") and text.endswith(")"):
        kids = []
        for part in _split_top(text[7:-1]):
            if part.startswith('"'):
                j = part.index('"', 1)
                name, rest = part[1:j], part[j + 1:]
            else:
                name, _, rest = part.partition(" ")
            kids.append((name, parse_duck_type(rest)))
        return ("struct", kids)
    if text.startswith("MAP(") and text.endswith(")"):
        k, v = _split_top(text[4:-1])
        return ("map", parse_duck_type(k), parse_duck_type(v))
    return ("leaf", text)


def _vector_values(v, ty, n):
    """One host mi_vector (flat, dictionary or nested) of n rows -> python values."""
    shift = v.validity_shift
    if v.kind == _ffi.K_DICT:
        dt = ty[1]
        base = _flat_values(dt, {"VARCHAR": 16, "BLOB": 16}.get(dt, _dict_width(dt)), v.dictionary, v.dictionary_validity,
                            v.dict_len + 1)
        sel = np.ctypeslib.as_array(C.cast(v.data, C.POINTER(C.c_uint32)), shape=(max(n, 1),))[:n]
        return [base[int(x)] for x in sel]
    if ty[0] == "leaf":
        return _flat_values(ty[1], v.out_width, v.data, v.validity, n, shift)
    ok = _valid_bits(v.validity, n, shift)
    if ty[0] in ("list", "map"):
        child = v.children[0]
        cty = ty[1] if ty[0] == "list" else ("struct", [("key", ty[1]), ("value", ty[2])])
        cvals = _vector_values(child, cty, child.count)
        ent = np.ctypeslib.as_array(C.cast(v.data, C.POINTER(C.c_uint64)), shape=(max(n, 1) * 2,))[: n * 2].reshape(-1, 2)
        out = []
        for r in range(n):
            if not ok[r]:
                out.append(None)
                continue
            vals = cvals[int(ent[r, 0]): int(ent[r, 0]) + int(ent[r, 1])]
            out.append([(e["key"], e["value"]) for e in vals] if ty[0] == "map" else vals)
        return out
    if ty[0] == "array":
        child = v.children[0]
        cvals = _vector_values(child, ty[1], child.count)
        return [cvals[r * ty[2]: (r + 1) * ty[2]] if ok[r] else None for r in range(n)]
    kids = [_vector_values(v.children[i], kt, n) for i, (_, kt) in enumerate(ty[1])]
    return [{nm: kid[r] for (nm, _), kid in zip(ty[1], kids)} if ok[r] else None for r in range(n)]


def chunk_to_columns(chunk, fields):
    """mi_data_chunk (host vectors) -> list of python value lists; dictionary vectors flattened, nested vectors as
    python lists / dicts / (key, value) tuples."""
    return [_vector_values(chunk.columns[ci], parse_duck_type(f["duck_type"]), chunk.size) for ci, f in enumerate(fields)]


def _dict_width(duck_type):
    if duck_type in _INT_TYPES:
        return np.dtype(_INT_TYPES[duck_type]).itemsize
    return {"BOOLEAN": 1, "FLOAT": 4, "DOUBLE": 8, "INTERVAL": 16}.get(duck_type, 8)


# ---------------------------------------------------------------------------------------------------- scan operator
class Relation:
    """Result of read_arrow / scan_arrow_ipc: bind info + a pull loop of <= 2048-row DataChunks."""

    def __init__(self, conn, handle, keep=None):
        self._conn = conn
        self._h = handle
        self._keep = keep
        n = C.c_int32(0)
        _ffi.check(_ffi.lib().mi_scan_bind(self._h, None, 0, C.byref(n)))
        fields = (_ffi.Field * max(n.value, 1))()
        _ffi.check(_ffi.lib().mi_scan_bind(self._h, fields, n.value, C.byref(n)))
        self.fields = [_field_dict(f) for f in fields[: n.value]]
        self._out_fields = self.fields
        self._projection = None
        self._initialised = False

    @property
    def columns(self):
        return [f["name"] for f in self._out_fields]

    @property
    def types(self):
        return [f["duck_type"] for f in self._out_fields]

    def project(self, names):
        """projection_pushdown = true (read_arrow.cpp:46).  Applied lazily at the first pull, like init_global."""
        by_name = {f["name"]: f for f in self.fields}
        missing = [n for n in names if n not in by_name]
        if missing:
            raise MiError(_ffi.MI_EINVAL, "Field '%s' does not exist in IPC file schema" % missing[0])
        self._projection = list(names)
        self._out_fields = [by_name[n] for n in names]
        return self

    def filter_range(self, column, lo, hi):
        _ffi.check(_ffi.lib().mi_scan_set_filter_range(self._h, column.encode(), lo, hi))
        return self

    def filter(self, expr):
        """Pushed-down predicate tree (mi_scan_set_filter).  `expr` is nested tuples:
            ("and", e1, e2, ...) | ("or", e1, e2, ...) | (column, op, value) with op in = <> != < <= > >= |
            (column, "in", [values]) | (column, "is null") | (column, "is not null") | (column, "starts_with", prefix)
        Constants are the stored integers, or str / bytes for VARCHAR / BLOB columns (byte-wise order).  NULL semantics are
        SQL's: a comparison with NULL is not true."""
        ops = {"=": _ffi.F_EQ, "==": _ffi.F_EQ, "<>": _ffi.F_NE, "!=": _ffi.F_NE, "<": _ffi.F_LT, "<=": _ffi.F_LE,
               ">": _ffi.F_GT, ">=": _ffi.F_GE, "is null": _ffi.F_IS_NULL, "is not null": _ffi.F_IS_NOT_NULL, "in": _ffi.F_IN,
               "starts_with": _ffi.F_STARTS_WITH}
        nodes, keep = [None], []

        def emit(at, e):
            if e[0] in ("and", "or") and len(e) > 1 and isinstance(e[1], tuple):
                kids = list(e[1:])
                first = len(nodes)
                nodes.extend([None] * len(kids))
                nodes[at] = _ffi.FilterNode(op=_ffi.F_AND if e[0] == "and" else _ffi.F_OR, first_child=first, n_children=len(kids))
                for k, kid in enumerate(kids):
                    emit(first + k, kid)
                return
            col, op = e[0], e[1].lower()
            n = _ffi.FilterNode(op=ops[op], column=col.encode())
            as_bytes = lambda v: v.encode() if isinstance(v, str) else bytes(v)
            if op == "in" and len(e[2]) > 0 and isinstance(e[2][0], (str, bytes)):
                vals = [as_bytes(v) for v in e[2]]
                ptrs = (C.c_char_p * len(vals))(*vals)
                lens = (C.c_int32 * len(vals))(*[len(v) for v in vals])
                keep.extend([vals, ptrs, lens])
                n.str_values, n.str_lens, n.n_values = ptrs, lens, len(vals)
            elif op == "in":
                arr = (C.c_int64 * max(len(e[2]), 1))(*[int(v) for v in e[2]])
                keep.append(arr)
                n.values, n.n_values = arr, len(e[2])
            elif op not in ("is null", "is not null") and isinstance(e[2], (str, bytes)):
                v = as_bytes(e[2])
                keep.append(v)
                n.str_value, n.str_len = v, len(v)
            elif op not in ("is null", "is not null"):
                n.value = int(e[2])
            nodes[at] = n

        emit(0, expr)
        arr = (_ffi.FilterNode * len(nodes))(*nodes)
        _ffi.check(_ffi.lib().mi_scan_set_filter(self._h, arr, len(nodes), 0))
        return self

    def chunks(self):
        self._init()
        ch = _ffi.DataChunk()
        while True:
            _ffi.check(_ffi.lib().mi_scan_next(self._h, C.byref(ch)))
            if ch.size == 0:
                return
            yield ch

    def fetch_columns(self, apply_filter=True):
        """All rows as python value lists per column (stored integer values for temporal / decimal types)."""
        out = [[] for _ in self._out_fields]
        for ch in self.chunks():
            cols = chunk_to_columns(ch, self._out_fields)
            if apply_filter and ch.sel:
                sel = [ch.sel[i] for i in range(ch.sel_count)]
                cols = [[c[i] for i in sel] for c in cols]   # (compacted chunks carry no sel: their rows are the selected ones)
            for o, c in zip(out, cols):
                o.extend(c)
        return out

    def fetchall(self):
        cols = self.fetch_columns()
        return list(zip(*cols)) if cols and cols[0] is not None else []

    def _init(self):
        if not self._initialised:
            if self._projection:
                arr = (C.c_char_p * len(self._projection))(*[n.encode() for n in self._projection])
                _ffi.check(_ffi.lib().mi_scan_init(self._h, arr, len(self._projection)))
            else:
                _ffi.check(_ffi.lib().mi_scan_init(self._h, None, 0))
            self._initialised = True

    def count(self, detail=False):
        """SELECT count(*) (rows passing the pushed-down filter, if any); the pull loop runs natively."""
        self._init()
        rows, sel, chunks = C.c_int64(), C.c_int64(), C.c_int64()
        _ffi.check(_ffi.lib().mi_scan_count(self._h, C.byref(rows), C.byref(sel), C.byref(chunks)))
        return dict(rows=rows.value, selected=sel.value, chunks=chunks.value) if detail else sel.value

    def sum_product(self, a, b, filters=()):
        """SELECT sum(a * b), count(*) WHERE lo <= f < hi ... evaluated on the GPU (mi_scan_sum_product); `filters` =
        [(column, lo, hi), ...] on stored integers.  Returns (sum as python int, rows selected, rows scanned)."""
        arr = (_ffi.RangeFilter * max(len(filters), 1))()
        for i, (c, lo, hi) in enumerate(filters):
            arr[i].column, arr[i].lo, arr[i].hi = c.encode(), lo, hi
        r = _ffi.SumProductResult()
        _ffi.check(_ffi.lib().mi_scan_sum_product(self._h, a.encode(), b.encode(), arr, len(filters), C.byref(r)))
        self._initialised = True
        return (r.sum_hi << 64) + r.sum_lo, r.rows_selected, r.rows_scanned

    def progress(self):
        return _ffi.lib().mi_scan_progress(self._h)

    def stats(self):
        """mi_scan_get_stats: record batches submitted, LZ4 bodies decompressed in HBM, bytes over PCIe, bytes decompressed."""
        st = _ffi.ScanStats()
        _ffi.check(_ffi.lib().mi_scan_get_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in _ffi.ScanStats._fields_}

    def close(self):
        if self._h:
            _ffi.lib().mi_scan_close(self._h)
            self._h = C.c_void_p()

    __del__ = close


# ---------------------------------------------------------------------------------------------------- tables (write side)
class Table:
    """A python-side stand-in for a DuckDB table: names, DuckDB types and python values per column."""

    def __init__(self, names, types, columns):
        self.names, self.types, self.columns = list(names), list(types), [list(c) for c in columns]
        assert len(self.names) == len(self.types) == len(self.columns)

    @property
    def num_rows(self):
        return len(self.columns[0]) if self.columns else 0


def _vector_from_python(values, duck_type, keep):
    """python values -> (data ndarray, validity ndarray|None) in DuckDB flat-vector layout."""
    n = len(values)
    ok = np.array([v is not None for v in values], dtype=bool)
    validity = None
    if not ok.all():
        validity = np.packbits(np.concatenate([ok, np.ones((-n) % 64, bool)]), bitorder="little").view(np.uint64).copy()
    t = duck_type.upper()
    if t in ("VARCHAR", "BLOB"):
        data = np.zeros((n, 16), np.uint8)
        for i, v in enumerate(values):
            if v is None:
                continue
            b = v.encode() if isinstance(v, str) else bytes(v)
            data[i, 0:4] = np.frombuffer(np.uint32(len(b)).tobytes(), np.uint8)
            if len(b) <= 12:
                data[i, 4: 4 + len(b)] = np.frombuffer(b, np.uint8)
            else:
                heap = np.frombuffer(b, np.uint8).copy()
                keep.append(heap)
                data[i, 4:8] = heap[:4]
                data[i, 8:16] = np.frombuffer(np.uint64(heap.ctypes.data).tobytes(), np.uint8)
        return data.reshape(-1), validity
    if t == "BOOLEAN":
        return np.array([1 if v else 0 for v in values], np.uint8), validity
    if t in ("FLOAT", "DOUBLE"):
        return np.array([0.0 if v is None else v for v in values], np.float32 if t == "FLOAT" else np.float64), validity
    if t.startswith("DECIMAL"):
        p = int(t[t.index("(") + 1: t.index(",")])
        if p > 18:
            data = np.zeros((n, 2), np.uint64)
            for i, v in enumerate(values):
                if v is not None:
                    data[i, 0] = v & 0xFFFFFFFFFFFFFFFF
                    data[i, 1] = (v >> 64) & 0xFFFFFFFFFFFFFFFF
            return data.reshape(-1), validity
        dt = np.int16 if p <= 4 else np.int32 if p <= 9 else np.int64
        return np.array([0 if v is None else v for v in values], dt), validity
    if t == "HUGEINT":
        data = np.zeros((n, 2), np.uint64)
        for i, v in enumerate(values):
            if v is not None:
                data[i, 0] = v & 0xFFFFFFFFFFFFFFFF
                data[i, 1] = (v >> 64) & 0xFFFFFFFFFFFFFFFF
        return data.reshape(-1), validity
    return np.array([0 if v is None else v for v in values], _INT_TYPES[t]), validity


def _fill_vector(vec, values, ty, keep):
    """python values -> one mi_vector (flat, or nested with child vectors) in DuckDB layout; `ty` from parse_duck_type."""
    n = len(values)
    vec.count = n
    if ty[0] == "leaf":
        data, validity = _vector_from_python(values, ty[1], keep)
        keep.append(data)
        vec.data = data.ctypes.data
    else:
        ok = np.array([v is not None for v in values], dtype=bool)
        validity = None
        if not ok.all():
            validity = np.packbits(np.concatenate([ok, np.ones((-n) % 64, bool)]), bitorder="little").view(np.uint64).copy()
        if ty[0] in ("list", "map"):
            ent = np.zeros((max(n, 1), 2), np.uint64)
            flat = []
            for i, v in enumerate(values):
                if v is None:
                    continue
                items = list(v.items()) if isinstance(v, dict) else list(v)
                ent[i] = (len(flat), len(items))
                flat.extend(items)
            keep.append(ent)
            vec.data = ent.ctypes.data
            kids = (_ffi.Vector * 1)()
            if ty[0] == "map":
                cty = ("struct", [("key", ty[1]), ("value", ty[2])])
                flat = [None if e is None else {"key": e[0], "value": e[1]} for e in flat]
            else:
                cty = ty[1]
            _fill_vector(kids[0], flat, cty, keep)
        elif ty[0] == "array":
            flat = []
            for v in values:
                flat.extend([None] * ty[2] if v is None else list(v))
            kids = (_ffi.Vector * 1)()
            _fill_vector(kids[0], flat, ty[1], keep)
        else:  # struct
            kids = (_ffi.Vector * len(ty[1]))()
            for k, (nm, kt) in enumerate(ty[1]):
                _fill_vector(kids[k], [None if v is None else v.get(nm) for v in values], kt, keep)
        keep.append(kids)
        vec.children = kids
        vec.n_children = len(kids)
    if validity is not None:
        keep.append(validity)
        vec.validity = validity.ctypes.data


def _chunks_from_table(table, keep, chunk_rows=VECTOR_SIZE):
    """Table -> list of mi_data_chunk (host vectors, <= 2048 rows each) the way DuckDB feeds a sink."""
    chunks = []
    trees = [parse_duck_type(t) for t in table.types]
    for r0 in range(0, max(table.num_rows, 0), chunk_rows):
        r1 = min(table.num_rows, r0 + chunk_rows)
        vecs = (_ffi.Vector * len(table.names))()
        for ci, (ty, col) in enumerate(zip(trees, table.columns)):
            _fill_vector(vecs[ci], col[r0:r1], ty, keep)
        keep.append(vecs)
        ch = _ffi.DataChunk(size=r1 - r0, n_columns=len(table.names), columns=vecs)
        chunks.append(ch)
    return chunks


def encode_schema(names, types):
    """The Arrow IPC Schema message the writer emits for these DuckDB columns (host only)."""
    fields = _c_fields(names, types)
    size = C.c_int64(0)
    _ffi.check(_ffi.lib().mi_encode_schema(fields, len(names), None, 0, C.byref(size)))
    buf = np.zeros(size.value, np.uint8)
    _ffi.check(_ffi.lib().mi_encode_schema(fields, len(names), buf.ctypes.data, buf.size, C.byref(size)))
    return buf.tobytes()


def _c_fields(names, types):
    arr = (_ffi.Field * len(names))()
    for i, (n, t) in enumerate(zip(names, types)):
        arr[i].name = n.encode()
        arr[i].duck_type = t.encode()
    return arr


# ---------------------------------------------------------------------------------------------------- connection
class Connection:
    """The extension's function surface for this path, bound to one GPU."""

    def __init__(self, device=0):
        import weakref
        self.ctx = Context(device)
        self._relations = weakref.WeakSet()   # scans hold streams / buffers of the context: they go first

    def close(self):
        for rel in list(self._relations):
            rel.close()
        self.ctx.close()

    def _relation(self, handle, keep=None):
        rel = Relation(self, handle, keep)
        self._relations.add(rel)
        return rel

    # -- scan ------------------------------------------------------------------------------------------
    @staticmethod
    def _options(union_by_name=False, filename=False, hive_partitioning=False, rank=0, world=1, device_resident=False,
                 accept_dictionaries=False, zero_copy_direct=None, unset_all_valid=False, filter_compact=False,
                 pipeline_depth=0, host_decompress=False, **unknown):
        for k in unknown:
            # MultiFileFunction rejects unknown named parameters (test/sql/read_arrow.test:40-43)
            raise MiError(_ffi.MI_EINVAL, 'Invalid named parameter "%s" for function read_arrow' % k)
        return _ffi.ScanOptions(union_by_name=int(union_by_name), filename=int(filename),
                                hive_partitioning=int(hive_partitioning), rank=rank, world=world,
                                device_resident=int(device_resident), accept_dictionaries=int(accept_dictionaries),
                                # None = the library's default (on for device-resident consumers), False = never
                                zero_copy_direct=0 if zero_copy_direct is None else (1 if zero_copy_direct else -1),
                                unset_all_valid=int(unset_all_valid), filter_compact=int(filter_compact),
                                pipeline_depth=int(pipeline_depth), host_decompress=(-1 if host_decompress == "gpu" else int(bool(host_decompress))))

    def read_arrow(self, paths, contexts=None, **options):
        """FROM read_arrow('file') / read_arrow(['a', 'b']) / read_arrow('dir/*.arrow') (globs expanded here).
        `contexts`: a list of Context objects (one per GPU, or several on one GPU) = the in-library multi-device scan
        (mi_scan_open_files_multi): record batches are dealt over them and chunks come back in record-batch order."""
        import glob as _glob
        if isinstance(paths, (str, bytes, os.PathLike)):
            paths = [paths]
        expanded = []
        for p in paths:
            p = os.fspath(p)
            if any(ch in p for ch in "*?["):
                hits = sorted(_glob.glob(p))
                if not hits:
                    raise MiError(_ffi.MI_EIO, 'No files found that match the pattern "%s"' % p)
                expanded.extend(hits)
            else:
                expanded.append(p)
        opts = self._options(**options)
        arr = (C.c_char_p * len(expanded))(*[os.fsencode(p) for p in expanded])
        h = C.c_void_p()
        if contexts:
            cs = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
            _ffi.check(_ffi.lib().mi_scan_open_files_multi(cs, len(contexts), arr, len(expanded), C.byref(opts), C.byref(h)))
            return self._relation(h, keep=list(contexts))
        _ffi.check(_ffi.lib().mi_scan_open_files(self.ctx._h, arr, len(expanded), C.byref(opts), C.byref(h)))
        return self._relation(h)

    def scan_arrow_ipc(self, buffers, **options):
        """FROM scan_arrow_ipc([{ptr, size}, ...]); buffers may be bytes-like objects or (ptr, size) tuples."""
        keep, arr = [], (_ffi.IpcBuffer * len(buffers))()
        for i, b in enumerate(buffers):
            if isinstance(b, tuple):
                arr[i].ptr, arr[i].size = b
            else:
                a = _as_u8(b)
                keep.append(a)
                arr[i].ptr, arr[i].size = a.ctypes.data, a.size
        opts = self._options(**options)
        h = C.c_void_p()
        _ffi.check(_ffi.lib().mi_scan_open_buffers(self.ctx._h, arr, len(buffers), C.byref(opts), C.byref(h)))
        return self._relation(h, keep=(keep, buffers))

    def from_arrow(self, message_reader, **options):
        """con.from_arrow(pyarrow.ipc.MessageReader): serialises each message back to its IPC bytes and scans the
        buffers, which is what the DuckDB python client does for this extension (test/python/test_arrow_ipc_scan.py:22-31)."""
        bufs = []
        while True:
            try:
                msg = message_reader.read_next_message()
            except StopIteration:
                break
            if msg is None:
                break
            bufs.append(msg.serialize().to_pybytes())
        return self.scan_arrow_ipc([b"".join(bufs)], **options)

    # -- write -----------------------------------------------------------------------------------------
    def to_arrow_ipc(self, table, chunk_size=120 * VECTOR_SIZE):
        """FROM to_arrow_ipc((FROM T)) -> [(ipc BLOB, header BOOL), ...]: first the schema message (header = True),
        then one record-batch message per 120 x 2048 rows (src/include/writer/to_arrow_ipc.hpp:28)."""
        L = _ffi.lib()
        w = C.c_void_p()
        _ffi.check(L.mi_ipc_serializer_create(self.ctx._h, _c_fields(table.names, table.types), len(table.names), C.byref(w)))
        try:
            blob, size = C.c_void_p(), C.c_int64()
            _ffi.check(L.mi_ipc_serialize_schema(w, C.byref(blob), C.byref(size)))
            rows = [(C.string_at(blob.value, size.value), True)]
            keep = []
            chunks = _chunks_from_table(table, keep)
            per_msg = max(1, chunk_size // VECTOR_SIZE)
            for i in range(0, len(chunks), per_msg):
                group = chunks[i: i + per_msg]
                arr = (_ffi.DataChunk * len(group))(*group)
                _ffi.check(L.mi_ipc_serialize_chunks(w, arr, len(group), C.byref(blob), C.byref(size)))
                rows.append((C.string_at(blob.value, size.value), False))
            return rows
        finally:
            L.mi_writer_close(w)

    def copy_to(self, table, path, preserve_insertion_order=True, file_size_bytes=None, arrow_large_buffer_size=False, **options):
        """COPY table TO 'path' (FORMAT ARROWS, row_group_size ..., chunk_size ..., row_group_size_bytes ...,
        row_groups_per_file ..., kv_metadata {...}).  With row_groups_per_file / file_size_bytes `path` becomes a
        directory of data_<i>.arrows files, like DuckDB's file rotation.  Returns the list of files written."""
        L = _ffi.lib()
        o = _ffi.WriteOptions()
        _ffi.check(L.mi_write_options_init(C.byref(o)))
        o.preserve_insertion_order = int(preserve_insertion_order)
        o.arrow_large_buffer_size = int(arrow_large_buffer_size)   # SET arrow_large_buffer_size=true
        for k, v in options.items():
            if k.lower() == "kv_metadata":
                if not isinstance(v, dict):
                    raise MiError(_ffi.MI_EINVAL, "Expected kv_metadata argument to be a STRUCT")
                for kk, vv in v.items():
                    vb = vv if isinstance(vv, bytes) else str(vv).encode()
                    _ffi.check(L.mi_write_options_add_kv(C.byref(o), kk.encode(), vb, len(vb)))
            elif k.lower() == "format":
                continue
            else:
                _ffi.check(L.mi_write_options_set(C.byref(o), k.encode(), None if v is None else str(v).encode()))
        _ffi.check(L.mi_write_options_finalize(C.byref(o)))
        rotate = o.row_groups_per_file > 0 or file_size_bytes is not None
        files, keep = [], []
        if isinstance(table, Relation):   # COPY (FROM read_arrow(...)) TO ...: the scan's chunks go straight into the sink
            names, types, source = table.columns, table.types, table.chunks()
        else:
            names, types, source = table.names, table.types, _chunks_from_table(table, keep)
        fields = _c_fields(names, types)

        def open_writer():
            if rotate:
                os.makedirs(path, exist_ok=True)
                p = os.path.join(path, "data_%d.arrows" % len(files))
            else:
                p = path
            w = C.c_void_p()
            _ffi.check(L.mi_writer_open(self.ctx._h, os.fsencode(p), fields, len(names), C.byref(o), C.byref(w)))
            files.append(p)
            return w

        w = open_writer()
        try:
            if isinstance(table, Relation) and not rotate:   # native pump: scan -> sink without a Python loop
                table._init()
                _ffi.check(L.mi_writer_sink_scan(w, table._h, None))
                source = ()
            for ch in source:
                _ffi.check(L.mi_writer_sink(w, C.byref(ch)))
                if rotate and L.mi_writer_rotate_next_file(w, -1 if file_size_bytes is None else file_size_bytes):
                    _ffi.check(L.mi_writer_finalize(w))
                    L.mi_writer_close(w)
                    w = open_writer()
            _ffi.check(L.mi_writer_finalize(w))
        finally:
            L.mi_writer_close(w)
        return files


# ---------------------------------------------------------------------------------------------------- synthetic input
def synth_lineitem_stream(scale_factor=1.0, seed=42, n_rows=0, rows_per_batch=0, with_validity=True, n_threads=0, out=None,
                          first_row=0):
    """Seeded TPC-H-shaped lineitem as an Arrow IPC stream (include/mi_synth.h).  Returns (uint8 ndarray, info)."""
    L = _ffi.lib()
    o = _ffi.SynthOptions(scale_factor=scale_factor, seed=seed, rows_per_batch=rows_per_batch, n_rows=n_rows,
                          first_row=first_row, with_validity=int(with_validity), n_threads=n_threads)
    rows, nb, size = C.c_int64(), C.c_int64(), C.c_int64()
    _ffi.check(L.mi_synth_lineitem_layout(C.byref(o), C.byref(rows), C.byref(nb), C.byref(size), None, 0))
    offs = (C.c_int64 * (nb.value + 1))()
    _ffi.check(L.mi_synth_lineitem_layout(C.byref(o), C.byref(rows), C.byref(nb), C.byref(size), offs, nb.value + 1))
    if out is None:
        out = np.empty(size.value, dtype=np.uint8)
    assert out.size >= size.value
    _ffi.check(L.mi_synth_lineitem_fill(C.byref(o), out.ctypes.data, out.size))
    return out[: size.value], dict(n_rows=rows.value, n_batches=nb.value, stream_size=size.value,
                                   batch_offsets=list(offs))
