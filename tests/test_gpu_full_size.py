"""BASELINE.json's full single-GPU size (configs[1]: TPC-H SF10 lineitem, 59 986 052 rows, 489 record batches) through
the C ABI, checked with size-independent properties that never leave the GPU: encode(decode(stream)) reproduces every
Arrow buffer of the stream bit for bit (K7 inverts K1-K4 on all 7 824 (batch, column) pairs), order keys stay sorted across
the whole table, value ranges and string_t invariants hold on every row, and the fused Q6 aggregate equals the same
aggregate computed with torch on the decoded vectors."""
import numpy as np
import pytest

import duckdb_arrow_amd as da
from duckdb_arrow_amd import _ffi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sf10():
    torch = pytest.importorskip("torch")
    from duckdb_arrow_amd.hbm import HbmStream
    buf, info = da.synth_lineitem_stream(scale_factor=10.0, seed=42)
    ctx = da.Context(0)
    hs = HbmStream(ctx, buf, memory="torch")   # torch owns the buffers: the checks below run on the device through torch
    hs.launch()
    assert hs.status() == 0
    torch.cuda.synchronize()
    return torch, ctx, hs, buf, info


def test_sf10_properties_on_device(sf10):
    torch, ctx, hs, buf, info = sf10
    assert info["n_rows"] == 59986052 and info["n_batches"] == 489
    out = hs.d_out
    last = torch.tensor(-1, dtype=torch.int64, device="cuda")
    rows = 0
    ok = torch.ones((), dtype=torch.bool, device="cuda")
    for lay in hs.layout:
        n = lay["nrows"]
        rows += n
        cols = {e["name"]: e for e in lay["columns"]}
        view = lambda e, dt: out[e["data_off"]: e["data_off"] + n * e["width"]].view(dt)
        key = view(cols["l_orderkey"], torch.int64)
        ok &= (key[0] >= last) & (key[1:] >= key[:-1]).all()
        last = key[-1]
        ship = view(cols["l_shipdate"], torch.int32)
        ok &= (ship.min() >= 8036) & (ship.max() <= 10561)
        disc = view(cols["l_discount"], torch.int64)
        ok &= (disc.min() >= 0) & (disc.max() <= 10)
        s = view(cols["l_comment"], torch.int32).reshape(-1, 4)
        ok &= (s[:, 0].min() >= 10) & (s[:, 0].max() <= 43)
        for e in lay["columns"]:   # lineitem has no NULLs: every validity word is all ones (pad bits included)
            words = out[e["valid_off"]: e["valid_off"] + ((n + 63) // 64) * 8].view(torch.int64)
            ok &= (words == -1).all()
    assert rows == info["n_rows"] and bool(ok.item())


def test_sf10_encode_inverts_decode_on_device(sf10):
    torch, ctx, hs, buf, info = sf10
    in_base, out_base = hs.d_in.data_ptr(), hs.d_out.data_ptr()
    enc_kind = {_ffi.K_COPY: _ffi.K_ENC_COPY, _ffi.K_DEC128: _ffi.K_ENC_DEC128, _ffi.K_STR32: _ffi.K_ENC_STR32}
    total, spans = 0, []
    for lay in hs.layout:
        n = lay["nrows"]
        for e in lay["columns"]:
            nb = 3 if e["kind"] == _ffi.K_STR32 else 2
            sz = [(n + 7) // 8, e["buffers"][1][1], e["buffers"][2][1] if nb == 3 else 0]
            offs = []
            for b in sz:
                offs.append(total)
                total += (b + 63) // 64 * 64 + 64
            spans.append((lay, e, offs, sz))
    arena = torch.zeros(total + 256, dtype=torch.uint8, device="cuda")
    ab = arena.data_ptr()
    tasks = []
    for lay, e, offs, sz in spans:
        is_str = e["kind"] == _ffi.K_STR32
        tasks.append(da.make_task(enc_kind[e["kind"]], lay["nrows"], out_base + e["data_off"], ab + offs[1],
                                  validity=out_base + e["valid_off"], out_validity=ab + offs[0], out_aux=(ab + offs[2]) if is_str else 0,
                                  buf2=in_base, ptr_base=0, buf2_len=sz[2] if is_str else 0, param=0 if is_str else e["param"]))
    plan = da.Plan(ctx, tasks)
    plan.launch(torch.cuda.current_stream().cuda_stream)
    assert plan.status() == 0
    ok = torch.ones((), dtype=torch.bool, device="cuda")
    d_in = hs.d_in
    for lay, e, offs, sz in spans:
        body, n = lay["body_off"], lay["nrows"]
        for k in (1, 2):   # offsets / data buffers are byte identical to the source stream
            if sz[k]:
                src = body + e["buffers"][k][0]
                ok &= torch.equal(arena[offs[k]: offs[k] + sz[k]], d_in[src: src + sz[k]])
        # validity: always emitted, all valid, pad bits 1 (ArrowAppender)
        ok &= (arena[offs[0]: offs[0] + sz[0]] == 0xFF).all()
    assert bool(ok.item())
    assert sum(plan.null_counts()) == 0


def test_sf10_fused_q6_equals_torch_on_decoded_vectors(sf10, tmp_path_factory):
    torch, ctx, hs, buf, info = sf10
    want, selected = 0, 0
    out = hs.d_out
    for lay in hs.layout:
        n = lay["nrows"]
        cols = {e["name"]: e for e in lay["columns"]}
        view = lambda nm, dt: out[cols[nm]["data_off"]: cols[nm]["data_off"] + n * cols[nm]["width"]].view(dt)
        ship, disc, qty, price = view("l_shipdate", torch.int32), view("l_discount", torch.int64), view("l_quantity", torch.int64), view("l_extendedprice", torch.int64)
        keep = (ship >= 8766) & (ship < 9131) & (disc >= 5) & (disc < 8) & (qty < 2400)
        want += int((price[keep] * disc[keep]).sum().item())   # < 2^63 for one batch of lineitem
        selected += int(keep.sum().item())
    con = da.Connection(0)
    rel = con.scan_arrow_ipc([buf])
    total, sel, scanned = rel.sum_product("l_extendedprice", "l_discount",
                                          [("l_shipdate", 8766, 9131), ("l_discount", 5, 8), ("l_quantity", -2**63, 2400)])
    assert (total, sel, scanned) == (want, selected, info["n_rows"])


def test_config5_commits_shape_bit_exact_and_equal_to_pyarrow():
    """BASELINE configs[4] ("config 5" of SURVEY 8d) at its shape: arrow-commits columns (40-byte commit, ts[us, UTC], int32,
    bool, 9-513 byte message) with 10 % NULLs in every nullable column and two dictionary-encoded columns (int32 indices),
    1 230 704 rows in 11 record batches (the last one 1 904 rows: a ragged final tile), resident in HBM.  Every batch bit-exact against the oracle (dictionaries included);
    validity words and fixed-width values of every batch, and all values of the first and the last batch, equal pyarrow's
    reading of the same stream.  Beyond the reference, which cannot read dictionary-encoded IPC at all
    (src/ipc/stream_reader/base_stream_reader.cpp:86-96): parity for K5 is pinned by pyarrow, not by the reference."""
    import os
    import sys
    import pyarrow as pa
    import pyarrow.ipc as ipc
    from duckdb_arrow_amd.hbm import HbmStream
    from oracle import pyoracle as po
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from commits_bench import commits_stream
    from helpers import canon_stream, pyarrow_columns
    from test_gpu_decode_parity import assert_streams_equal

    n_rows = 1228800 + 1904
    buf = commits_stream(n_rows)
    ctx = da.Context(0)
    hs = HbmStream(ctx, buf, accept_dictionaries=True)
    hs.launch()
    assert hs.status() == 0
    got = hs.fetch()
    fields, want = po.decode_stream(buf)
    assert len(got) == 11 and got[-1]["nrows"] == 1904 and sum(b["nrows"] for b in got) == n_rows
    assert_streams_equal(got, want)
    for gb, wb in zip(got, want):
        for gc, wc in zip(gb["columns"], wb["columns"]):
            if "dictionary" in wc:
                assert np.array_equal(gc["dictionary"]["data"], wc["dictionary"]["data"]), gc["name"]
                assert np.array_equal(gc["dictionary"]["validity"], wc["dictionary"]["validity"]), gc["name"]
    # pyarrow, every batch: validity bits and the fixed-width values of the valid rows
    batches = list(ipc.open_stream(pa.py_buffer(buf)))
    names = batches[0].schema.names
    for gb, pb in zip(got, batches):
        n = pb.num_rows
        for ci, name in enumerate(names):
            col = pb.column(ci)
            ok = np.unpackbits(gb["columns"][ci]["validity"].view(np.uint8), bitorder="little")[:n].astype(bool)
            assert np.array_equal(ok, np.asarray(col.is_valid())), name
            if name == "time":
                assert np.array_equal(gb["columns"][ci]["data"].view(np.int64)[ok], np.asarray(col.cast(pa.int64()).drop_null()))
            elif name == "files":
                assert np.array_equal(gb["columns"][ci]["data"].view(np.int32)[ok], np.asarray(col.drop_null()))
            elif name == "merge":
                assert np.array_equal(gb["columns"][ci]["data"].view(np.uint8)[ok].astype(bool), np.asarray(col.drop_null()))
            elif name in ("author", "component"):
                assert np.array_equal(gb["columns"][ci]["data"].view(np.uint32)[ok], np.asarray(col.indices.drop_null()).astype(np.uint32))
            else:   # strings: the lengths of every row (the bytes: first and last batch below)
                lens = gb["columns"][ci]["data"].reshape(-1, 16)[:, :4].copy().view(np.uint32).reshape(-1)
                off = np.frombuffer(col.buffers()[1], np.int32, n + 1)
                assert np.array_equal(lens[ok], np.diff(off).astype(np.uint32)[ok]), name
    # pyarrow, first and last batch: every value
    for bi in (0, len(got) - 1):
        vals = canon_stream(fields, [got[bi]], buf)
        table = pa.Table.from_batches([batches[bi]])
        for name, exp in zip(names, pyarrow_columns(table)):
            assert vals[name] == exp, (bi, name)
    hs.close()
