"""HBM-resident decode of a whole Arrow IPC stream through the kernel-level C ABI (mi_plan_*).

torch is plumbing only (device memory + the stream handle): the stream's bytes are uploaded once, the host
reader (mi_reader_*) slices every record batch into buffers, ONE plan holds a task per (record batch, column)
and a launch is a handful of kernels regardless of the number of batches.  This is what bench.py times and what
the GPU parity tests compare against the CPU checker.
"""
import ctypes as C

import numpy as np

from . import _ffi
from . import Reader, Plan, make_task


def _round_up(v, a=256):
    return (v + a - 1) // a * a


class HbmStream:
    """An IPC stream resident in HBM plus the plan that decodes every record batch of it."""

    def __init__(self, ctx, host_buf, columns=None, accept_dictionaries=False, device="cuda:0", with_validity_out=True):
        import torch
        self.torch = torch
        self.ctx = ctx
        host = host_buf if isinstance(host_buf, np.ndarray) else np.frombuffer(host_buf, np.uint8)
        self.host = host
        rd = Reader(buffers=[host])
        self.fields = rd.schema()
        if columns is not None:
            rd.set_projection(list(columns))
            by_name = {f["name"]: f for f in self.fields}
            self.out_fields = [by_name[c] for c in columns]
        else:
            self.out_fields = self.fields
        # host parse: every message -> per-column buffer spans (absolute positions inside the stream)
        self.batches, self.dict_batches = [], {}
        while True:
            b = rd.next_batch(accept_dictionaries=accept_dictionaries)
            if b is None:
                break
            (self.dict_batches.__setitem__(b["dict_id"], b) if b["is_dictionary"] else self.batches.append(b))
        rd.close()
        # HBM: the stream itself (8-byte slack so the last buffer's padding is addressable) + the output arena
        self.d_in = torch.empty(_round_up(host.size + 64), dtype=torch.uint8, device=device)
        self.d_in[: host.size].copy_(torch.from_numpy(host), non_blocking=False)
        base = self.d_in.data_ptr()
        layout, off = [], 0
        tasks = []
        by_top = {i: f for i, f in enumerate(self.fields)}

        def add_task(b, ci, field, kind, param, width, nrows, out_rows, param2=0):
            nonlocal off
            data_off = off
            off += _round_up(out_rows * width + 16)
            valid_off = off
            off += _round_up(((out_rows + 63) // 64) * 8 + 8)
            sp = b["buffers"][3 * ci: 3 * ci + 3]
            body = b["body_file_offset"]
            nbuf = 3 if kind in (_ffi.K_STR32, _ffi.K_STR64) else 2
            data_span = sp[2] if nbuf == 3 else sp[1]
            entry = dict(name=field["name"], kind=kind, param=param, width=width, nrows=nrows, data_off=data_off,
                         valid_off=valid_off, null_count=b["null_count"][ci], ptr_base=body + data_span[0],
                         buffers=sp, body_off=body)
            if nrows > 0:
                tasks.append((entry, dict(kind=kind, nrows=nrows, buf1=base + body + sp[1][0],
                                          validity=(base + body + sp[0][0]) if sp[0][1] else 0,
                                          buf2=(base + body + sp[2][0]) if nbuf == 3 else 0,
                                          buf2_len=sp[2][1] if nbuf == 3 else 0, ptr_base=body + data_span[0],
                                          param=param, param2=param2, null_count=b["null_count"][ci])))
            return entry

        self.dict_layout = {}
        for did, b in self.dict_batches.items():
            f = by_top[b["column_field"][0]]
            vkind, vparam, vwidth = _value_plan(f)
            n = _dict_rows(b, vkind, vparam)
            self.dict_layout[did] = add_task(b, 0, f, vkind, vparam, vwidth, n, n + 1)
        for b in self.batches:
            cols = []
            for ci, top in enumerate(b["column_field"]):
                f = by_top[top]
                param2 = self.dict_layout[f["dict_id"]]["nrows"] if f["kind"] == _ffi.K_DICT else 0
                cols.append(add_task(b, ci, f, f["kind"], f["param"], f["out_width"], b["length"], b["length"], param2))
            layout.append(dict(nrows=b["length"], columns=cols, body_off=b["body_file_offset"], body_len=b["body_size"]))
        self.layout = layout
        self.out_bytes = off
        self.d_out = torch.zeros(max(off, 256), dtype=torch.uint8, device=device)
        obase = self.d_out.data_ptr()
        ctasks = []
        for entry, t in tasks:
            ctasks.append(make_task(t["kind"], t["nrows"], t["buf1"], obase + entry["data_off"], validity=t["validity"],
                                    buf2=t["buf2"], out_validity=(obase + entry["valid_off"]) if with_validity_out else 0,
                                    ptr_base=t["ptr_base"], buf2_len=t["buf2_len"], param=t["param"], param2=t["param2"],
                                    null_count=t["null_count"]))
        self.plan = Plan(ctx, ctasks)
        self.n_rows = sum(b["length"] for b in self.batches)

    def launch(self, stream=None):
        s = self.torch.cuda.current_stream().cuda_stream if stream is None else stream
        self.plan.launch(s)

    def status(self):
        return self.plan.status()

    def stats(self):
        return self.plan.stats()

    def fetch(self):
        """D2H of the output arena -> per record batch, per column: data bytes + validity words (numpy)."""
        self.torch.cuda.synchronize()
        out = self.d_out[: max(self.out_bytes, 1)].cpu().numpy()
        dicts = {}
        for did, e in self.dict_layout.items():
            n = e["nrows"]
            d = out[e["data_off"]: e["data_off"] + (n + 1) * e["width"]].copy()
            v = out[e["valid_off"]: e["valid_off"] + ((n + 1 + 63) // 64) * 8].copy().view(np.uint64)
            dicts[did] = dict(kind=e["kind"], param=e["param"], width=e["width"], data=d[: n * e["width"]],
                              validity=v[: max((n + 63) // 64, 0)], nrows=n, ptr_base=e["ptr_base"])
        batches = []
        by_name = {f["name"]: f for f in self.fields}
        for b in self.layout:
            cols = []
            for e in b["columns"]:
                n = e["nrows"]
                d = out[e["data_off"]: e["data_off"] + n * e["width"]].copy()
                v = out[e["valid_off"]: e["valid_off"] + ((n + 63) // 64) * 8].copy().view(np.uint64)
                f = by_name[e["name"]]
                cols.append(dict(name=e["name"], kind=e["kind"], param=e["param"], width=e["width"], data=d, validity=v,
                                 rc=0, buffers=e["buffers"], ptr_base=e["ptr_base"], null_count=e["null_count"],
                                 dictionary=dicts.get(f["dict_id"]) if e["kind"] == _ffi.K_DICT else None))
            batches.append(dict(nrows=b["nrows"], columns=cols, body_off=b["body_off"], body_len=b["body_len"]))
        return batches


def _value_plan(f):
    """(kind, param, width) of a dictionary's VALUE type (the field's own plan is MI_K_DICT)."""
    t = f["arrow_type"]
    if t in (5, 4):
        return _ffi.K_STR32, 0, 16
    if t in (20, 19):
        return _ffi.K_STR64, 0, 16
    if t == 2:
        return _ffi.K_COPY, f["bit_width"] // 8, f["bit_width"] // 8
    if t == 3:
        w = 4 if f["precision"] == 1 else 8
        return _ffi.K_COPY, w, w
    if t == 6:
        return _ffi.K_BOOL, 0, 1
    if t == 8 and f["unit"] == 0:
        return _ffi.K_COPY, 4, 4
    if t == 7 and f["bit_width"] == 128:
        p = f["precision"]
        return (_ffi.K_DEC128, 2, 2) if p <= 4 else (_ffi.K_DEC128, 4, 4) if p <= 9 else (_ffi.K_DEC128, 8, 8) if p <= 18 \
            else (_ffi.K_COPY, 16, 16)
    raise NotImplementedError("dictionary value type %s" % f["format"])


def _dict_rows(b, kind, param):
    """A DictionaryBatch's row count from its buffers (the reader reports lengths through the spans)."""
    return b["length"]
