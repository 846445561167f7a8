"""GPU parity: the HIP decode path (through the C ABI, mi_plan_*) vs the CPU oracle, bit for bit, on the golden
fixtures, on seeded synthetic inputs, and on the error / edge cases.  Run with -m gpu on an MI355X."""
import ctypes as C
import os

import numpy as np
import pytest

import duckdb_arrow_amd as da
from duckdb_arrow_amd import _ffi
from oracle import pyoracle as po

from helpers import canon_stream, column_digest
from test_oracle_golden import STREAM_FILES, load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return da.Context(0)


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def assert_nodes_equal(gc, wc, where):
    assert wc["rc"] == 0, where
    assert gc["kind"] == wc["kind"] and gc["width"] == wc["width"] and gc["nrows"] == wc["nrows"], where
    assert np.array_equal(gc["validity"], wc["validity"]), (where, "validity")
    if not np.array_equal(gc["data"], wc["data"]):
        w = max(gc["width"], 1)
        bad = np.nonzero(np.any(gc["data"].reshape(-1, w) != wc["data"].reshape(-1, w), axis=1))[0]
        raise AssertionError("%s: %d rows differ, first row %d: got %s want %s" % (
            where, len(bad), bad[0], gc["data"].reshape(-1, w)[bad[0]].tolist(), wc["data"].reshape(-1, w)[bad[0]].tolist()))
    if gc["kind"] == po.K_DICT:
        assert_nodes_equal(gc["dictionary"], wc["dictionary"], where + ".dictionary")
    assert len(gc["children"]) == len(wc["children"]), where
    if gc["children"]:
        assert gc["win"] == wc["win"], where
    for g, w_ in zip(gc["children"], wc["children"]):
        assert_nodes_equal(g, w_, where + "." + g["name"])
    gc["field"] = wc["field"]  # lets helpers.canon_node interpret the GPU result with the oracle's schema info


def assert_streams_equal(got, want):
    assert len(got) == len(want)
    for bi, (gb, wb) in enumerate(zip(got, want)):
        assert gb["nrows"] == wb["nrows"]
        assert [c["name"] for c in gb["columns"]] == [c["name"] for c in wb["columns"]]
        for gc, wc in zip(gb["columns"], wb["columns"]):
            assert_nodes_equal(gc, wc, "batch %d column %s" % (bi, gc["name"]))


@pytest.mark.parametrize("rel", STREAM_FILES)
def test_golden_files_bit_exact(ctx, golden_dir, expected, rel):
    from duckdb_arrow_amd.hbm import HbmStream
    buf = load(golden_dir, rel)
    hs = HbmStream(ctx, buf, accept_dictionaries=True)
    hs.launch()
    assert hs.status() == 0
    got = hs.fetch()
    fields, want = po.decode_stream(buf)
    assert_streams_equal(got, want)
    # and, independently of the oracle, the logical values pyarrow sees (expected.json)
    vals = canon_stream(fields, got, buf)
    for name, v in vals.items():
        assert column_digest(v) == expected[rel]["columns"][name], name


@pytest.mark.parametrize("with_validity", [True, False])
def test_synthetic_lineitem_bit_exact(ctx, with_validity):
    from duckdb_arrow_amd.hbm import HbmStream
    buf, info = da.synth_lineitem_stream(scale_factor=0.1, seed=3, with_validity=with_validity)
    hs = HbmStream(ctx, buf)
    hs.launch()
    assert hs.status() == 0
    _, want = po.decode_stream(buf)
    assert_streams_equal(hs.fetch(), want)
    part = hs.fetch(batches=[1, len(want) - 1])   # sampled fetch (what bench.py uses for tables larger than host memory)
    assert part[0] is None and [i for i, b in enumerate(part) if b is not None] == [1, len(want) - 1]
    assert_streams_equal([part[1], part[-1]], [want[1], want[-1]])
    st = hs.stats()
    assert st["rows"] == 16 * info["n_rows"]
    # algorithmic bytes: 158 B/row written; ~174.85 (172.85 without bitmaps, which are skipped when null_count = 0)
    assert abs(st["bytes_written"] / info["n_rows"] - 158.0) < 0.1
    assert 172.0 < st["bytes_read"] / info["n_rows"] < 173.5


def test_projection_decodes_only_requested_columns(ctx, golden_dir):
    from duckdb_arrow_amd.hbm import HbmStream
    buf = load(golden_dir, "ref_data/test.arrows")
    hs = HbmStream(ctx, buf, columns=["message", "time"])
    hs.launch()
    assert hs.status() == 0
    _, want = po.decode_stream(buf, columns={"message", "time"})
    got = hs.fetch()
    for gb, wb in zip(got, want):
        w = {c["name"]: c for c in wb["columns"]}
        assert [c["name"] for c in gb["columns"]] == ["message", "time"]
        for gc in gb["columns"]:
            assert np.array_equal(gc["data"], w[gc["name"]]["data"])


def _run_task(ctx, torch, kind, nrows, buf1, *, validity=None, buf2=None, width, param=0, param2=0, null_count=-1,
              row_offset=0, ptr_base=0):
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).copy()).cuda() if a is not None else None
    pad = lambda a: None if a is None else np.concatenate([np.ascontiguousarray(a).view(np.uint8), np.zeros(16, np.uint8)])
    d1, dv, d2 = dev(pad(buf1)), dev(pad(validity)), dev(pad(buf2))
    out = torch.zeros(max(nrows * width, 16) + 16, dtype=torch.uint8, device="cuda")
    outv = torch.zeros(((nrows + 63) // 64) * 8 + 16, dtype=torch.uint8, device="cuda")
    t = da.make_task(kind, nrows, d1.data_ptr(), out.data_ptr(), validity=dv.data_ptr() if dv is not None else 0,
                     buf2=d2.data_ptr() if d2 is not None else 0, out_validity=outv.data_ptr(), ptr_base=ptr_base,
                     row_offset=row_offset, buf2_len=(len(np.ascontiguousarray(buf2).view(np.uint8)) if buf2 is not None else 0),
                     param=param, param2=param2, null_count=null_count)
    plan = da.Plan(ctx, [t])
    plan.launch(torch.cuda.current_stream().cuda_stream)
    status = plan.status()
    return out.cpu().numpy()[: nrows * width], outv.cpu().numpy()[: ((nrows + 63) // 64) * 8].view(np.uint64), status


@pytest.mark.parametrize("row_offset", [0, 1, 7, 8, 13, 64, 100, 2048, 2051])
@pytest.mark.parametrize("nrows", [1, 63, 64, 65, 2047, 2048, 2049, 5000])
def test_validity_and_bool_at_arbitrary_offsets(ctx, torch, row_offset, nrows):
    """K1 / K2 with a non-zero Arrow array offset (the CPU path's shift-right case)."""
    rng = np.random.default_rng(row_offset * 7919 + nrows)
    total = row_offset + nrows
    bitmap = rng.integers(0, 256, (total + 7) // 8 + 8, dtype=np.uint8)
    bitmap = bitmap[: ((total + 7) // 8 + 7) // 8 * 8]
    bits = rng.integers(0, 256, len(bitmap), dtype=np.uint8)
    data, valid, status = _run_task(ctx, torch, _ffi.K_BOOL, nrows, bits, validity=bitmap, width=1, row_offset=row_offset)
    assert status == 0
    want_valid = np.zeros((nrows + 63) // 64, np.uint64)
    po.lib().orc_validity(bitmap.ctypes.data, -1, row_offset, nrows, want_valid.ctypes.data)
    want = np.zeros(nrows, np.uint8)
    po.lib().orc_bool.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    po.lib().orc_bool(bits.ctypes.data, row_offset, nrows, want.ctypes.data)
    assert np.array_equal(valid, want_valid)
    assert np.array_equal(data, want)


@pytest.mark.parametrize("row_offset", [0, 3, 2048])
def test_strings_and_decimals_with_row_offset(ctx, torch, row_offset):
    rng = np.random.default_rng(5 + row_offset)
    nrows, total = 3000, 3000 + row_offset
    lens = rng.integers(0, 30, total)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    payload = rng.integers(32, 127, int(off[-1]) + 8, dtype=np.uint8)
    bitmap = rng.integers(0, 256, (total + 7) // 8 + 16, dtype=np.uint8)
    bitmap = bitmap[: len(bitmap) // 8 * 8]
    data, valid, status = _run_task(ctx, torch, _ffi.K_STR32, nrows, off, validity=bitmap, buf2=payload[: int(off[-1])],
                                    width=16, row_offset=row_offset, ptr_base=0x7000_0000_0000)
    assert status == 0
    wv = np.zeros((nrows + 63) // 64, np.uint64)
    po.lib().orc_validity(bitmap.ctypes.data, -1, row_offset, nrows, wv.ctypes.data)
    want = np.zeros(nrows * 16, np.uint8)
    rc = po.lib().orc_string32(off.ctypes.data, payload.ctypes.data, wv.ctypes.data, row_offset, nrows, 0x7000_0000_0000,
                               want.ctypes.data)
    assert rc == 0 and np.array_equal(valid, wv) and np.array_equal(data, want)
    dec = rng.integers(-10**15, 10**15, total).astype(np.int64)
    dec128 = np.stack([dec, np.where(dec < 0, -1, 0).astype(np.int64)], axis=1).reshape(-1)
    data, valid, status = _run_task(ctx, torch, _ffi.K_DEC128, nrows, dec128, validity=bitmap, width=8, param=8,
                                    row_offset=row_offset)
    ok = po.valid_bits(wv, nrows)
    assert status == 0 and np.array_equal(data.view(np.int64), np.where(ok, dec[row_offset: row_offset + nrows], 0))


def test_null_count_zero_ignores_the_bitmap(ctx, torch):
    """GetValidityMask copies the bitmap only when null_count != 0."""
    vals = np.arange(1000, dtype=np.int64)
    bitmap = np.zeros(128, np.uint8)  # all "null" -- but null_count says 0
    data, valid, status = _run_task(ctx, torch, _ffi.K_COPY, 1000, vals, validity=bitmap, width=8, param=8, null_count=0)
    assert status == 0 and np.array_equal(data.view(np.int64), vals) and (valid == np.uint64(0xFFFFFFFFFFFFFFFF)).all()


def test_device_status_flags(ctx, torch):
    # decreasing offsets => FULL validation failure
    off = np.array([0, 5, 3, 9, 12], np.int32)
    _, _, st = _run_task(ctx, torch, _ffi.K_STR32, 4, off, buf2=np.zeros(16, np.uint8), width=16)
    assert st & _ffi.ST_BAD_OFFSETS
    # last offset beyond the data buffer
    off = np.array([0, 5, 7, 9, 40], np.int32)
    _, _, st = _run_task(ctx, torch, _ffi.K_STR32, 4, off, buf2=np.zeros(16, np.uint8), width=16)
    assert st & _ffi.ST_BAD_OFFSETS
    with pytest.raises(da.MiError, match="offsets buffer is not monotonically"):
        _ffi.check(_ffi.lib().mi_status_to_error(st))
    # timestamp[s] * 1e6 overflow => ConversionException
    src = np.array([1, 2**62, 5], np.int64)
    data, _, st = _run_task(ctx, torch, _ffi.K_MUL_I64, 3, src, width=8, param=1000000)
    assert st == _ffi.ST_MUL_OVERFLOW and data.view(np.int64).tolist() == [1000000, 0, 5000000]
    with pytest.raises(da.MiError, match="Could not convert") as e:
        _ffi.check(_ffi.lib().mi_status_to_error(st))
    assert e.value.code == _ffi.MI_ERANGE
    # int64 string offsets past 4 GB
    off = np.array([0, 5, 2**32 + 10], np.int64)
    _, _, st = _run_task(ctx, torch, _ffi.K_STR64, 1, off, buf2=np.zeros(16, np.uint8), width=16)
    assert st == 0
    big = np.array([0, 2**32 + 10], np.int64)
    t = da.make_task(_ffi.K_STR64, 1, 8, 16, buf2=0, buf2_len=2**33)  # never launched: validation only
    del t
    # negative dictionary index => "DuckDB only supports indices that fit on an uint32"
    idx = np.array([0, 1, -1, 2], np.int32)
    data, _, st = _run_task(ctx, torch, _ffi.K_DICT, 4, idx, width=4, param=4 | (1 << 8), param2=3)
    assert st == _ffi.ST_INDEX_RANGE
    # an index past the dictionary: flagged, and the slot selects the dictionary's NULL entry instead of pointing outside
    idx = np.array([0, 7, 2, 3], np.int32)
    data, _, st = _run_task(ctx, torch, _ffi.K_DICT, 4, idx, width=4, param=4 | (1 << 8), param2=3)
    assert st == _ffi.ST_DICT_INDEX and data.view(np.uint32).tolist() == [0, 3, 2, 3]
    with pytest.raises(da.MiError, match="dictionary index out of range"):
        _ffi.check(_ffi.lib().mi_status_to_error(st))
    # decimal that does not fit its declared physical type
    dec = np.array([5, 0, 2**40, 0], np.int64)
    _, _, st = _run_task(ctx, torch, _ffi.K_DEC128, 2, dec, width=4, param=4)
    assert st == _ffi.ST_DECIMAL_RANGE


def test_plan_rejects_bad_tasks(ctx):
    with pytest.raises(da.MiError, match="unknown kind"):
        da.Plan(ctx, [da.make_task(99, 10, 16, 16)])
    with pytest.raises(da.MiError, match="out_data must be 16-byte aligned"):
        da.Plan(ctx, [da.make_task(_ffi.K_COPY, 10, 16, 24, param=8)])
    with pytest.raises(da.MiError, match="COPY width"):
        da.Plan(ctx, [da.make_task(_ffi.K_COPY, 10, 16, 32, param=3)])
    da.Plan(ctx, [])  # an empty plan is fine


def test_filter_range_matches_oracle(ctx, torch, golden_dir, expected):
    """K6 on the decoded l_shipdate column: 1994-01-01 <= d < 1995-01-01 (TPC-H Q6), plus the Q6 revenue KAT computed
    from GPU-decoded vectors and GPU selection vectors (arrow_test.js:423-424: 1193053.2253)."""
    from duckdb_arrow_amd.hbm import HbmStream
    buf = load(golden_dir, "lineitem_sf0_01_q6.arrows")
    hs = HbmStream(ctx, buf)
    hs.launch()
    assert hs.status() == 0
    revenue = passing = 0
    obase = hs.out_ptr
    for b, lay in zip(hs.fetch(), hs.layout):
        n = b["nrows"]
        cols = {c["name"]: c for c in b["columns"]}
        e = {c["name"]: c for c in lay["columns"]}["l_shipdate"]
        sel = torch.zeros(n + 16, dtype=torch.int32, device="cuda")
        cnt = torch.zeros((n + 2047) // 2048 + 4, dtype=torch.int32, device="cuda")
        da.filter_range(ctx, obase + e["data_off"], 4, obase + e["valid_off"], n, 8766, 9131, sel.data_ptr(), cnt.data_ptr(),
                        torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        sel, cnt = sel.cpu().numpy().view(np.uint32), cnt.cpu().numpy().view(np.uint32)
        ship = cols["l_shipdate"]["data"].view(np.int32)
        rows = []
        for w in range((n + 2047) // 2048):
            m = min(2048, n - w * 2048)
            want = np.zeros(m, np.uint32)
            wc = po.lib().orc_filter_range_i32(ship[w * 2048:].ctypes.data, cols["l_shipdate"]["validity"][w * 32:].ctypes.data, m,
                                               8766, 9131, want.ctypes.data)
            assert cnt[w] == wc and np.array_equal(sel[w * 2048: w * 2048 + wc], want[:wc])
            rows.extend((w * 2048 + sel[w * 2048: w * 2048 + wc]).tolist())
        rows = np.array(rows, dtype=np.int64)
        qty = cols["l_quantity"]["data"].view(np.int64)[rows]
        price = cols["l_extendedprice"]["data"].view(np.int64)[rows]
        disc = cols["l_discount"]["data"].view(np.int64)[rows]
        keep = (disc >= 5) & (disc <= 7) & (qty < 2400)
        revenue += int(np.sum(price[keep] * disc[keep]))
        passing += int(keep.sum())
    assert passing == expected["kat"]["q6_sf0_01_rows_passing"] == 1191
    assert revenue == expected["kat"]["q6_sf0_01_revenue_scale4"] == 11930532253


def test_sf1_lineitem_properties_and_sampled_parity(ctx, torch):
    """Full-size check (SF1 = 6 001 215 rows, 49 batches): size-independent properties on every row + bit-exact parity
    with the oracle on sampled batches."""
    from duckdb_arrow_amd.hbm import HbmStream
    buf, info = da.synth_lineitem_stream(scale_factor=1.0, seed=42)
    assert info["n_rows"] == 6001215 and info["n_batches"] == 49
    hs = HbmStream(ctx, buf)
    hs.launch()
    assert hs.status() == 0
    got = hs.fetch()
    # properties: row count, sortedness of the sparse order keys, value ranges, string_t invariants
    assert sum(b["nrows"] for b in got) == 6001215
    last = -1
    for b in got:
        cols = {c["name"]: c for c in b["columns"]}
        ok = cols["l_orderkey"]["data"].view(np.int64)
        assert ok[0] >= last and (np.diff(ok) >= 0).all()
        last = ok[-1]
        ship = cols["l_shipdate"]["data"].view(np.int32)
        assert ship.min() >= 8036 and ship.max() <= 10561
        disc = cols["l_discount"]["data"].view(np.int64)
        assert disc.min() >= 0 and disc.max() <= 10
        s = cols["l_comment"]["data"].reshape(-1, 16)
        lens = s[:, :4].copy().view(np.uint32).reshape(-1)
        assert lens.min() >= 10 and lens.max() <= 43
        ptr = s[:, 8:].copy().view(np.uint64).reshape(-1)
        long_ = lens > 12
        off = (ptr[long_] - np.uint64(cols["l_comment"]["ptr_base"])).astype(np.int64)
        assert (np.diff(off) > 0).all() and off.min() >= 0
        assert all((c["validity"] == np.uint64(0xFFFFFFFFFFFFFFFF)).all() or c["validity"][-1] != 0 for c in b["columns"])
    # sampled bit-exact parity
    msgs = [m for m in po.walk_stream(buf) if m["type"] == po.MSG_RECORD_BATCH]
    fields, _, _ = po.decode_schema(buf[po.walk_stream(buf, 1)[0]["meta_off"]:][: po.walk_stream(buf, 1)[0]["meta_len"]])
    for bi in (0, 17, 48):
        m = msgs[bi]
        sub = np.concatenate([buf[: msgs[0]["prefix_off"]], buf[m["prefix_off"]: m["body_off"] + m["body_len"]]])
        shift = m["prefix_off"] - msgs[0]["prefix_off"]
        _, want = po.decode_stream(sub, ptr_base_of=lambda i, body_off, boff: body_off + boff + shift)
        assert_streams_equal([got[bi]], want)


def test_zero_copy_layout_leaves_direct_columns_in_the_body(ctx):
    """mi_hbm_options.zero_copy_direct: int64 / date32 columns without NULLs get no task -- their vector is the Arrow
    buffer in HBM (alias_off) -- and every other column is transcoded exactly as in the full plan.  With unset_all_valid
    on top, columns without NULLs carry no validity words at all (the reference leaves the mask unset)."""
    from duckdb_arrow_amd.hbm import HbmStream
    buf, info = da.synth_lineitem_stream(scale_factor=0.02, seed=5)
    full = HbmStream(ctx, buf)
    _, want = po.decode_stream(buf)
    for unset in (False, True):
        hs = HbmStream(ctx, buf, zero_copy_direct=True, unset_all_valid=unset, share_stream_of=full)
        assert hs.n_tasks == full.n_tasks * 9 // 16        # 7 of lineitem's 16 columns are plain fixed width
        assert hs.stats()["bytes_written"] < full.stats()["bytes_written"]
        hs.launch()
        assert hs.status() == 0
        got = hs.fetch()
        for gb, wb in zip(got, want):
            for gc, wc in zip(gb["columns"], wb["columns"]):
                assert gc["aliased"] == (gc["kind"] == _ffi.K_COPY), gc["name"]
                assert gc["validity_unset"] == (unset or gc["aliased"]), gc["name"]
                assert_nodes_equal(gc, wc, gc["name"])    # aliased data = the stream bytes, unset validity = all ones
        hs.close()


@pytest.mark.parametrize("width,dtype", [(2, np.int16), (4, np.int32), (8, np.int64)])
@pytest.mark.parametrize("n", [1, 7, 8, 2047, 2048, 2049, 4 * 2048 - 3, 4 * 2048, 4 * 2048 + 9, 11 * 2048 + 1234])
def test_filter_range_sizes_widths_and_nulls(ctx, torch, width, dtype, n):
    """K6 at sizes around the 2048-row window and the 4-window workgroup boundaries, every value width, with NULLs: per
    window the count and the ascending window-relative indices equal a numpy filter (NULL rows never pass)."""
    rng = np.random.default_rng(n * 10 + width)
    vals = rng.integers(-1000, 1000, n).astype(dtype)
    ok = rng.random(n) < 0.8
    words = np.packbits(np.concatenate([ok, np.zeros((-n) % 64, bool)]), bitorder="little").view(np.uint64).copy()
    d_vals = torch.from_numpy(vals.view(np.uint8).copy()).cuda()
    d_valid = torch.from_numpy(words.view(np.uint8).copy()).cuda()
    nw = (n + 2047) // 2048
    sel = torch.full((nw * 2048 + 16,), -1, dtype=torch.int32, device="cuda")
    cnt = torch.full((nw + 8,), -1, dtype=torch.int32, device="cuda")
    da.filter_range(ctx, d_vals.data_ptr(), width, d_valid.data_ptr(), n, -100, 250, sel.data_ptr(), cnt.data_ptr(),
                    torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    sel, cnt = sel.cpu().numpy(), cnt.cpu().numpy()
    for w in range(nw):
        lo, hi = w * 2048, min(n, (w + 1) * 2048)
        want = np.nonzero((vals[lo:hi] >= -100) & (vals[lo:hi] < 250) & ok[lo:hi])[0]
        assert cnt[w] == len(want)
        assert np.array_equal(sel[w * 2048: w * 2048 + len(want)], want)
    assert (cnt[nw:] == -1).all()      # nothing written past the last window
