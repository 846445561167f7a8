"""HBM-resident decode of a whole Arrow IPC stream through the kernel-level C ABI (mi_plan_*).

torch is plumbing only (device memory + the stream handle): the stream's bytes are uploaded once, the host
reader (mi_reader_*) slices every record batch into buffers, ONE plan holds a task per (record batch, field node)
and a launch is a handful of kernels regardless of the number of batches.  This is what bench.py times and what
the GPU parity tests compare against the CPU checker.
"""
import numpy as np

from . import _ffi
from . import Reader, Plan, make_task

VECTOR_SIZE = 2048
ARRAY_ALIGN = int(__import__("os").environ.get("MI_HBM_ARRAY_ALIGN", "65536"))


def _round_up(v, a=256):
    return (v + a - 1) // a * a


def _windows(n):
    return (list(range(0, n, VECTOR_SIZE)) + [n]) if n else [0, 0]


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
        # host parse: every message -> field nodes with their buffer spans (positions relative to the message body)
        self.batches, self.dict_batches = [], {}
        while True:
            b = rd.next_batch(accept_dictionaries=accept_dictionaries)
            if b is None:
                break
            (self.dict_batches.__setitem__(b["dict_id"], b) if b["is_dictionary"] else self.batches.append(b))
        rd.close()
        # HBM: the stream itself (slack so the last buffer's padding is addressable) + the output arena + small tables
        self.d_in = torch.empty(_round_up(host.size + 64), dtype=torch.uint8, device=device)
        self.d_in[: host.size].copy_(torch.from_numpy(host), non_blocking=False)
        self._off = 0
        self._aux = []        # (offset, int64/uint64 ndarray): list window tables, string-view buffer tables
        self._aux_bytes = 0
        self._tasks = []      # (entry, kwargs with symbolic aux / out references)
        self.dict_layout = {}
        for did, b in self.dict_batches.items():
            self.dict_layout[did] = self._add_node(b, b["column_node"][0], _windows(b["length"]), True, extra_rows=1)
        self.layout = []
        for b in self.batches:
            arena0 = self._off
            cols = [self._add_node(b, ni, _windows(b["length"]), True) for ni in b["column_node"]]
            self.layout.append(dict(nrows=b["length"], columns=cols, body_off=b["body_file_offset"], body_len=b["body_size"],
                                    arena=(arena0, self._off)))
        self.out_bytes = self._off
        self.d_out = torch.zeros(max(self._off, 256), dtype=torch.uint8, device=device)
        aux = np.zeros(max(self._aux_bytes // 8, 1), np.uint64)
        for off, arr in self._aux:
            aux[off // 8: off // 8 + arr.size] = arr.view(np.uint64)
        self.d_aux = torch.from_numpy(aux.view(np.uint8).copy()).to(device)
        ibase, obase, abase = self.d_in.data_ptr(), self.d_out.data_ptr(), self.d_aux.data_ptr()
        ctasks = []
        self._aliasable = []   # plain fixed-width top-level columns without NULLs: zero-copy candidates
        for e, t in self._tasks:
            self._aliasable.append(t["kind"] == _ffi.K_COPY and t["null_count"] == 0 and t["depth"] == 0)
            e["alias_addr"] = (ibase + t["buf1"]) if self._aliasable[-1] else 0
            ctasks.append(make_task(
                t["kind"], t["nrows"], ibase + t["buf1"] if t["buf1"] is not None else 0, obase + e["data_off"],
                validity=(ibase + t["validity"]) if t["validity"] is not None else 0,
                buf2=(ibase + t["buf2"]) if t.get("buf2") is not None else ((abase + t["aux"]) if t.get("aux") is not None else 0),
                out_validity=(obase + e["valid_off"]) if (with_validity_out or t["kind"] == _ffi.K_STRUCT) else 0,
                out_aux=(obase + t["parent_valid_off"]) if t.get("parent_valid_off") is not None else 0,
                ptr_base=t.get("ptr_base", 0), buf2_len=t.get("buf2_len", 0), param=t.get("param", 0), param2=t.get("param2", 0),
                null_count=t["null_count"], depth=t["depth"], parent_div=t.get("parent_div", 0)))
        self.plan = Plan(ctx, ctasks)
        self._ctasks = ctasks
        self._zero_copy_plan = None
        self.n_rows = sum(b["length"] for b in self.batches)

    def zero_copy_plan(self):
        """The plan a device-resident consumer needs when plain fixed-width columns alias the IPC body in HBM (the
        reference's zero-copy DirectConversion; mi_scan_options.zero_copy_direct): only the columns that really need a
        transcode keep their task; the aliased vectors are entry["alias_addr"] with validity = all valid."""
        if self._zero_copy_plan is None:
            self._zero_copy_plan = Plan(self.ctx, [t for t, a in zip(self._ctasks, self._aliasable) if not a])
        return self._zero_copy_plan

    # ------------------------------------------------------------------------------------------------ layout
    def _alloc(self, rows, width):
        # Every output array starts on a 64 KiB boundary of the arena (itself 2 MiB aligned).  Measured on MI355X with
        # fresh processes on one box (SF10 lineitem, ms per step): 256 B alignment 3.39-3.43, 4 KiB 3.39, 64 KiB 3.34-3.36,
        # 2 MiB 3.40 -- the copy and dec128 kernels gain 2-3 % when a 16 KB tile never straddles a 64 KiB page fragment.
        # Costs ~32 KiB of padding per (batch, column): 0.5 GB of 288 GB at SF10.  MI_HBM_ARRAY_ALIGN overrides (A/B).
        data_off = self._off
        self._off += _round_up(rows * width + 16, ARRAY_ALIGN)
        valid_off = self._off
        self._off += _round_up(((rows + 63) // 64) * 8 + 8, ARRAY_ALIGN if rows >= 65536 else 256)
        return data_off, valid_off

    def _aux_table(self, arr):
        off = self._aux_bytes
        self._aux.append((off, np.ascontiguousarray(arr)))
        self._aux_bytes += _round_up(arr.size * 8, 64)
        return off

    def _add_node(self, b, ni, win, win_is_tiles, parent_valid_off=None, parent_div=0, extra_rows=0):
        nodes = b["nodes"]
        nd = nodes[ni]
        kind, width, n, body = nd["kind"], nd["out_width"], nd["length"], b["body_file_offset"]
        if kind == 0:
            raise NotImplementedError("field %r (arrow type %d) is not decoded by the path" % (nd["name"], nd["arrow_type"]))
        sp = nd["spans"]
        data_off, valid_off = self._alloc(n + extra_rows, max(width, 1))
        children = [i for i in range(ni + 1, len(nodes)) if nodes[i]["parent"] == ni]
        pos = lambda s: body + s[0]
        entry = dict(name=nd["name"], kind=kind, param=nd["param"], width=width, nrows=n, data_off=data_off, valid_off=valid_off,
                     null_count=nd["null_count"], buffers=sp, body_off=body, win=list(win), children=[], arrow_type=nd["arrow_type"],
                     ptr_base=0)
        t = dict(kind=kind, nrows=n, param=nd["param"], null_count=nd["null_count"], depth=nd["depth"],
                 validity=pos(sp[0]) if (len(sp) > 0 and sp[0][1]) else None, buf1=pos(sp[1]) if len(sp) > 1 else None)
        if parent_valid_off is not None:
            t["parent_valid_off"], t["parent_div"] = parent_valid_off, parent_div
        if kind in (_ffi.K_STR32, _ffi.K_STR64):
            t.update(buf2=pos(sp[2]), buf2_len=sp[2][1], ptr_base=pos(sp[2]))
            entry["ptr_base"] = pos(sp[2])
        elif kind == _ffi.K_FIXED_BINARY:
            t.update(ptr_base=pos(sp[1]))
            entry["ptr_base"] = pos(sp[1])
        elif kind == _ffi.K_STRVIEW:
            table = np.zeros(max(2 * (len(sp) - 2), 2), np.uint64)
            for j, s in enumerate(sp[2:]):
                table[2 * j], table[2 * j + 1] = pos(s), s[1]   # addresses = positions inside the stream (heap = stream)
            t.update(aux=self._aux_table(table), buf2_len=len(sp) - 2)
        elif kind == _ffi.K_DICT:
            t["param2"] = self.dict_layout[self._dict_id_of(nd["name"])]["nrows"]
            entry["dict_id"] = self._dict_id_of(nd["name"])
        elif kind in (_ffi.K_LIST32, _ffi.K_LIST64):
            offw = 4 if kind == _ffi.K_LIST32 else 8
            offs = self.host[pos(sp[1]): pos(sp[1]) + (n + 1) * offw].view(np.int32 if offw == 4 else np.int64) if n else np.zeros(1, np.int64)
            t["param"] = nodes[children[0]]["length"]
            if not win_is_tiles:
                t.update(aux=self._aux_table(np.array(win, np.int64)), buf2_len=len(win))
            child_win = [int(offs[r]) for r in win] if n else [0] * len(win)
            if any(b < a for a, b in zip(child_win, child_win[1:])) or (child_win and (child_win[0] < 0 or child_win[-1] > t["param"])):
                # windows place the child vectors (and are dereferenced by the kernel): never from unchecked offsets
                raise ValueError("list offsets of %r are not monotonically non-decreasing inside the child column" % nd["name"])
        if n > 0 or kind == _ffi.K_STRUCT:
            if n > 0:
                self._tasks.append((entry, t))
        if kind in (_ffi.K_LIST32, _ffi.K_LIST64):
            entry["children"].append(self._add_node(b, children[0], child_win, False))
        elif kind == _ffi.K_STRUCT:
            is_fixed_list = nd["arrow_type"] == 16
            size = int(nd["param"]) if is_fixed_list else 1
            cwin = [r * size for r in win] if is_fixed_list else list(win)
            for c in children:
                entry["children"].append(self._add_node(b, c, cwin, win_is_tiles and not is_fixed_list, parent_valid_off=valid_off,
                                                        parent_div=size if is_fixed_list else 1))
        return entry

    def _dict_id_of(self, name):
        for f in self.fields:
            if f["name"] == name and f["has_dictionary"]:
                return f["dict_id"]
        raise KeyError(name)

    # ------------------------------------------------------------------------------------------------ run
    def launch(self, stream=None):
        s = self.torch.cuda.current_stream().cuda_stream if stream is None else stream
        self.plan.launch(s)

    def status(self):
        return self.plan.status()

    def stats(self):
        return self.plan.stats()

    def fetch(self, batches=None):
        """D2H of the output arena -> per record batch, per column a node: data bytes + validity words (numpy) + children.
        `batches`: only these record batches (their arena ranges are copied one by one; the result keeps list positions,
        other entries are None) -- what a sampled check of a table much larger than host memory needs."""
        self.torch.cuda.synchronize()
        if batches is None:
            base = 0
            out = self.d_out[: max(self.out_bytes, 1)].cpu().numpy()
        want = None if batches is None else set(batches)

        def node(e, dict_extra=0):
            n = e["nrows"]
            d = out[e["data_off"] - base: e["data_off"] - base + n * e["width"]].copy()
            v = out[e["valid_off"] - base: e["valid_off"] - base + ((n + 63) // 64) * 8].copy().view(np.uint64)
            r = dict(name=e["name"], kind=e["kind"], param=e["param"], width=e["width"], data=d, validity=v, rc=0, nrows=n,
                     buffers=e["buffers"], ptr_base=e["ptr_base"], null_count=e["null_count"], win=e["win"],
                     children=[node(c) for c in e["children"]])
            if e["kind"] == _ffi.K_DICT:
                r["dictionary"] = dicts[e["dict_id"]]
            return r

        if batches is not None:   # dictionaries sit at the start of the arena, before the first record batch
            base, end = 0, (self.layout[0]["arena"][0] if self.layout else self.out_bytes)
            out = self.d_out[base: max(end, 1)].cpu().numpy()
        dicts = {did: node(e) for did, e in self.dict_layout.items()}
        res = []
        for bi, b in enumerate(self.layout):
            if want is not None:
                if bi not in want:
                    res.append(None)
                    continue
                base, end = b["arena"]
                out = self.d_out[base: max(end, base + 1)].cpu().numpy()
            res.append(dict(nrows=b["nrows"], columns=[node(e) for e in b["columns"]], body_off=b["body_off"], body_len=b["body_len"]))
        return res
