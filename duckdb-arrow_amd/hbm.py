"""HBM-resident decode of a whole Arrow IPC stream: a thin ctypes wrapper over mi_hbm_* (include/mi_arrow_ipc.h).

The planner (layout of one DuckDB vector array per (record batch, field node), one task per array, ONE plan for the
whole stream) lives in the library (csrc/batch_planner.cpp + hbm_stream.cpp) and is the same one the scan operator
uses; a C client reaches this mode through the same entry points (examples/hbm_scan.c).  Here only: handing the stream
over, launching, and turning the layout into numpy views for the tests.  Device memory belongs to the library unless
`memory="torch"` asks for torch tensors (tests that want to poke at the buffers with torch).
"""
import ctypes as C

import numpy as np

from . import _ffi

VECTOR_SIZE = 2048


class _PlanView:
    """The stream's plan seen through mi_hbm_* (stats / timed launches), shaped like duckdb_arrow_amd.Plan."""

    def __init__(self, owner):
        self._o = owner

    @property
    def n_tasks(self):
        return self._o.n_tasks

    def launch(self, stream=0):
        self._o.launch(stream)

    def status(self):
        return self._o.status()

    def stats(self):
        return self._o.stats()

    def class_stats(self):
        out = []
        for cls in range(_ffi.NUM_KERNEL_CLASSES):
            r, w, rows, tiles, name = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_char_p()
            _ffi.check(_ffi.lib().mi_hbm_class_stats(self._o._h, cls, C.byref(r), C.byref(w), C.byref(rows), C.byref(tiles), C.byref(name)))
            out.append(dict(kernel=name.value.decode(), bytes_read=r.value, bytes_written=w.value, rows=rows.value, tiles=tiles.value))
        return out

    def launch_timed(self, stream=0):
        ms = (C.c_float * _ffi.NUM_KERNEL_CLASSES)()
        _ffi.check(_ffi.lib().mi_hbm_launch_timed(self._o._h, C.c_void_p(stream or None), ms))
        return list(ms)


class HbmStream:
    """An IPC stream resident in HBM plus the plan that decodes every record batch of it (mi_hbm_open)."""

    def __init__(self, ctx, host_buf, columns=None, accept_dictionaries=False, device="cuda:0", zero_copy_direct=False,
                 unset_all_valid=False, pointer_mode=_ffi.HBM_PTR_STREAM_OFFSET, memory="library", array_align=0, share_stream_of=None):
        self.ctx = ctx
        host = host_buf if isinstance(host_buf, np.ndarray) else np.frombuffer(host_buf, np.uint8)
        self.host = np.ascontiguousarray(host)
        o = _ffi.HbmOptions()
        self._keep = []
        if columns is not None:
            arr = (C.c_char_p * len(columns))(*[c.encode() for c in columns])
            self._keep.append(arr)
            o.columns, o.n_columns = arr, len(columns)
        o.accept_dictionaries = int(accept_dictionaries)
        o.zero_copy_direct = int(zero_copy_direct)
        o.unset_all_valid = int(unset_all_valid)
        o.pointer_mode = pointer_mode
        o.array_align = array_align
        self._torch_in = self._torch_out = None
        if share_stream_of is not None:   # a second layout (e.g. zero-copy) over the stream another HbmStream uploaded
            o.device_stream = share_stream_of.layout_raw.device_stream
            self._keep.append(share_stream_of)
        if memory == "torch":
            import torch
            self._torch_in = torch.empty(self.host.size + 320, dtype=torch.uint8, device=device)
            self._torch_in[: self.host.size].copy_(torch.from_numpy(self.host))
            o.device_stream = self._torch_in.data_ptr()
            o.defer_arena = 1
        self._h = C.c_void_p()
        _ffi.check(_ffi.lib().mi_hbm_open(ctx._h, self.host.ctypes.data, self.host.size, C.byref(o), C.byref(self._h)))
        self.layout_raw = _ffi.HbmLayout()
        _ffi.check(_ffi.lib().mi_hbm_layout_get(self._h, C.byref(self.layout_raw)))
        if memory == "torch":
            import torch
            self._torch_out = torch.zeros(max(self.layout_raw.arena_bytes, 256), dtype=torch.uint8, device=device)
            _ffi.check(_ffi.lib().mi_hbm_set_arena(self._h, self._torch_out.data_ptr(), self._torch_out.numel()))
            _ffi.check(_ffi.lib().mi_hbm_layout_get(self._h, C.byref(self.layout_raw)))
        L = self.layout_raw
        self.out_bytes = L.arena_bytes
        self.n_rows = L.n_rows
        self.n_tasks = L.n_tasks
        self.in_ptr, self.out_ptr = L.device_stream, L.device_arena
        self.plan = _PlanView(self)
        self._build_layout()

    # torch views of the two device buffers (memory="torch" only)
    @property
    def d_in(self):
        if self._torch_in is None:
            raise RuntimeError('HbmStream(memory="torch") exposes d_in / d_out as torch tensors; the default keeps the memory in the library')
        return self._torch_in

    @property
    def d_out(self):
        if self._torch_out is None:
            raise RuntimeError('HbmStream(memory="torch") exposes d_in / d_out as torch tensors; the default keeps the memory in the library')
        return self._torch_out

    def close(self):
        if getattr(self, "_h", None):
            _ffi.lib().mi_hbm_close(self._h)
            self._h = C.c_void_p()

    __del__ = close

    # ------------------------------------------------------------------------------------------------ layout
    def _build_layout(self):
        """mi_hbm_layout -> the dict trees the tests walk: per record batch its column nodes (children nested)."""
        L = self.layout_raw
        nodes = []
        for i in range(L.n_nodes):
            c = L.nodes[i]
            nodes.append(dict(
                name=c.name.decode("utf-8", "replace"), kind=c.kind, param=c.param, width=c.out_width, nrows=c.nrows,
                data_off=c.data_off, valid_off=c.valid_off, null_count=c.null_count, arrow_type=c.arrow_type, ptr_base=c.ptr_base,
                buffers=[(L.spans[j].offset, L.spans[j].length) for j in range(c.first_span, c.first_span + c.n_spans)],
                win=[L.windows[j] for j in range(c.first_window, c.first_window + c.n_windows)],
                alias_addr=(L.device_stream + c.alias_off) if c.alias_off >= 0 else 0, alias_off=c.alias_off,
                dict_id=c.dict_id, children=[], _parent=c.parent))
        for e in nodes:
            if e["_parent"] >= 0:
                nodes[e["_parent"]]["children"].append(e)
        self.layout, self.dict_layout = [], {}
        for bi in range(L.n_batches):
            b = L.batches[bi]
            cols = [nodes[k] for k in range(b.first_node, b.first_node + b.n_nodes) if nodes[k]["_parent"] < 0]
            for e in nodes[b.first_node: b.first_node + b.n_nodes]:
                # Arrow buffers as (position relative to the message body, length), like mi_reader_next_batch reports them
                e["buffers"] = [(o - b.body_off, ln) for (o, ln) in e["buffers"]]
                e["body_off"] = b.body_off
            if b.is_dictionary:
                self.dict_layout[b.dict_id] = cols[0]
            else:
                self.layout.append(dict(nrows=b.nrows, columns=cols, body_off=b.body_off, body_len=b.body_len,
                                        arena=(b.arena_begin, b.arena_end)))

    # ------------------------------------------------------------------------------------------------ run
    def launch(self, stream=None):
        _ffi.check(_ffi.lib().mi_hbm_launch(self._h, C.c_void_p(stream or None)))

    def status(self):
        bits = C.c_uint32(0)
        _ffi.check(_ffi.lib().mi_hbm_status(self._h, C.byref(bits)))
        return bits.value

    def stats(self):
        r, w, rows, tiles = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        _ffi.check(_ffi.lib().mi_hbm_stats(self._h, C.byref(r), C.byref(w), C.byref(rows), C.byref(tiles)))
        return dict(bytes_read=r.value, bytes_written=w.value, rows=rows.value, tiles=tiles.value)

    def _d2h(self, off, length, from_stream=False):
        out = np.empty(max(length, 0), np.uint8)
        if length > 0:
            _ffi.check(_ffi.lib().mi_hbm_fetch(self._h, 1 if from_stream else 0, off, length, out.ctypes.data))
        return out

    def fetch(self, batches=None):
        """D2H of the output arena -> per record batch, per column a node: data bytes + validity words (numpy) + children.
        `batches`: only these record batches (their arena ranges are copied one by one; the result keeps list positions,
        other entries are None) -- what a sampled check of a table much larger than host memory needs."""
        self.status()   # waits for the stream the plan last ran on
        want = None if batches is None else set(batches)
        state = {"base": 0, "out": None}

        def node(e):
            n, base, out = e["nrows"], state["base"], state["out"]
            if e["alias_off"] >= 0:     # zero-copy: the values are the stream bytes themselves
                d = self._d2h(e["alias_off"], n * e["width"], from_stream=True)
            else:
                d = out[e["data_off"] - base: e["data_off"] - base + n * e["width"]].copy()
            if e["valid_off"] >= 0:
                v = out[e["valid_off"] - base: e["valid_off"] - base + ((n + 63) // 64) * 8].copy().view(np.uint64)
            else:                       # not materialised = every row valid (canonical: all ones)
                v = np.full((n + 63) // 64, np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64)
            r = dict(name=e["name"], kind=e["kind"], param=e["param"], width=e["width"], data=d, validity=v, rc=0, nrows=n,
                     buffers=e["buffers"], ptr_base=e["ptr_base"], null_count=e["null_count"], win=e["win"],
                     children=[node(c) for c in e["children"]], validity_unset=e["valid_off"] < 0, aliased=e["alias_off"] >= 0)
            if e["kind"] == _ffi.K_DICT:
                r["dictionary"] = dicts[e["dict_id"]]
            return r

        first_batch_at = self.layout[0]["arena"][0] if self.layout else self.out_bytes
        if batches is None:
            state["base"], state["out"] = 0, self._d2h(0, self.out_bytes)
        else:                           # dictionaries sit at the start of the arena, before the first record batch
            state["base"], state["out"] = 0, self._d2h(0, first_batch_at)
        dicts = {did: node(e) for did, e in self.dict_layout.items()}
        res = []
        for bi, b in enumerate(self.layout):
            if want is not None:
                if bi not in want:
                    res.append(None)
                    continue
                state["base"] = b["arena"][0]
                state["out"] = self._d2h(b["arena"][0], b["arena"][1] - b["arena"][0])
            res.append(dict(nrows=b["nrows"], columns=[node(e) for e in b["columns"]], body_off=b["body_off"], body_len=b["body_len"]))
        return res
