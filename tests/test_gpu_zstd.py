"""K8 for ZSTD bodies (kernels_zstd.inl + the copy stages of kernels_lz4.hip).  ZSTD is the codec the reference registers a
decompressor for (DuckDBDecompressZstd, base_stream_reader.cpp:11-32) and the one its benchmark writes
(benchmark/lineitem.py:135).  Same boundary as test_gpu_lz4.py: a device-resident scan -- compressed bytes over PCIe, entropy
stage per block, repeat offsets per frame, link / resolve / emit -- equals the host-consumer scan of the same file, whose
bodies libzstd decompresses on the reader's threads; mi_scan_get_stats shows which path ran.  The entropy stage itself is
also checked on the CPU against the same frames (tests/test_sanitizers.py::test_zstd_stages_on_the_cpu)."""
import os
import struct

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pytest

import duckdb_arrow_amd as da
from helpers import canon_python, pyarrow_columns
from test_gpu_lz4 import _device_scan, _frames, _tables

pytestmark = pytest.mark.gpu


def _c_getenv(name):
    """The C environment (what the library and the HIP runtime see): os.environ is Python's copy from start-up and does not
    show what the library's constructor set."""
    import ctypes
    libc = ctypes.CDLL(None)
    libc.getenv.restype = ctypes.c_char_p
    v = libc.getenv(name.encode())
    return v.decode() if v else None


@pytest.fixture(scope="module")
def con():
    return da.Connection(0)


def _write(path, table, chunk, level=1):
    with ipc.new_stream(path, table.schema, options=ipc.IpcWriteOptions(compression=pa.Codec("zstd", compression_level=level))) as w:
        w.write_table(table, max_chunksize=chunk)


@pytest.mark.parametrize("level", [1, 9, 19])
@pytest.mark.parametrize("case", ["mixed", "small_batches", "empty", "periodic"])
def test_zstd_bodies_decompressed_in_hbm_equal_the_host_decompressor(con, tmp_path, case, level):
    if level > 1 and case in ("empty", "small_batches"):
        pytest.skip("one level is enough for the small cases")
    name, table, chunk = next(t for t in _tables() if t[0] == case)
    path = str(tmp_path / (name + ".arrows"))
    _write(path, table, chunk, level)
    want = [canon_python(c) for c in con.read_arrow(path, accept_dictionaries=True, host_decompress=True).fetch_columns()]   # libzstd on host threads
    got, st = _device_scan(con, path, host_decompress="gpu")      # ZSTD in HBM on request
    assert got == want
    assert got == pyarrow_columns(table)      # and both equal pyarrow's reading of the table that was written
    nonempty = sum(1 for b in ipc.open_stream(path) if b.num_rows > 0)
    assert st["zstd_batches_on_device"] >= nonempty and st["lz4_batches_on_device"] == 0, st
    if nonempty:
        assert 0 < st["h2d_bytes"] < os.path.getsize(path) + 4096 and st["decompressed_bytes"] > 0
    rel = con.read_arrow(path, accept_dictionaries=True, host_decompress="gpu")      # a host consumer on the K8 path
    assert [canon_python(c) for c in rel.fetch_columns()] == want
    assert (rel.stats()["zstd_batches_on_device"] > 0) == (nonempty > 0)
    # auto: a device-resident consumer gets K8 when the process has hardware queues for many record batches side by side (the
    # library asks for 24 when it is loaded before the first HIP call; a deployment that pins GPU_MAX_HW_QUEUES low keeps the
    # host threads), and the host threads otherwise; a host consumer always keeps them
    got_auto, st_auto = _device_scan(con, path)
    many_queues = int(_c_getenv("GPU_MAX_HW_QUEUES") or 0) >= 12
    assert got_auto == want and (st_auto["zstd_batches_on_device"] > 0) == (many_queues and nonempty > 0), st_auto
    rel = con.read_arrow(path, accept_dictionaries=True)
    assert [canon_python(c) for c in rel.fetch_columns()] == want and rel.stats()["zstd_batches_on_device"] == 0
    got_host, st_host = _device_scan(con, path, host_decompress=True)
    assert got_host == want and st_host["zstd_batches_on_device"] == 0


def test_zstd_golden_files_projection_and_fused_consumers(con, golden_dir, tmp_path):
    for rel_path in ("lineitem_sf0_01_head.arrows", "edge_types.arrows", "edge_nested.arrows", "edge_dict.arrows"):
        t = ipc.open_stream(os.path.join(golden_dir, rel_path)).read_all()
        for level in (1, 12):
            path = str(tmp_path / ("zstd%d_%s" % (level, rel_path)))
            _write(path, t, 4096 if level == 1 else 60000, level)
            want = [canon_python(c) for c in con.read_arrow(path, accept_dictionaries=True, host_decompress=True).fetch_columns()]
            got, st = _device_scan(con, path, host_decompress="gpu")
            assert got == want, (rel_path, level)
            if rel_path == "lineitem_sf0_01_head.arrows":
                assert got == pyarrow_columns(t)
            assert (st["zstd_batches_on_device"] > 0) == (rel_path != "edge_nested.arrows"), rel_path   # list offsets are sampled on the host
    t = ipc.open_stream(os.path.join(golden_dir, "lineitem_sf0_01_q6.arrows")).read_all()
    path = str(tmp_path / "q6_zstd.arrows")
    _write(path, t, 8192)
    plain = os.path.join(golden_dir, "lineitem_sf0_01_q6.arrows")
    rel = con.read_arrow(path, device_resident=True, host_decompress="gpu")
    rel.filter_range("l_shipdate", 8766, 9130)
    a = rel.count(detail=True)
    assert rel.stats()["zstd_batches_on_device"] > 0
    rel2 = con.read_arrow(plain)
    rel2.filter_range("l_shipdate", 8766, 9130)
    b = rel2.count(detail=True)
    assert (a["rows"], a["selected"]) == (b["rows"], b["selected"])
    s1 = con.read_arrow(path, device_resident=True, host_decompress="gpu").sum_product("l_extendedprice", "l_discount", [("l_shipdate", 8766, 9130)])
    assert s1 == con.read_arrow(plain).sum_product("l_extendedprice", "l_discount", [("l_shipdate", 8766, 9130)])
    # projection: only the projected columns' frames cross PCIe
    t = ipc.open_stream(os.path.join(golden_dir, "lineitem_sf0_01_head.arrows")).read_all()
    path = str(tmp_path / "zstd_proj.arrows")
    _write(path, t, 4096)
    rel = con.read_arrow(path, device_resident=True, host_decompress="gpu").project(["l_shipdate", "l_quantity"])
    assert rel.count(detail=True)["rows"] == t.num_rows
    assert 0 < rel.stats()["h2d_bytes"] < os.path.getsize(path) * 0.5


def test_damaged_zstd_input_is_an_error_not_a_crash(con, tmp_path):
    """Damaged entropy tables, bitstreams that end early or late, offsets in front of the buffer, a wrong declared length: every
    table index and every position is bounded, the scan ends with the EIO of base_stream_reader.cpp:24-29 (or, when the host
    walk already refuses the frame, with libzstd's own error through the host path)."""
    n = 200000
    rng = np.random.default_rng(5)
    t = pa.table({"k": pa.array(np.arange(n, dtype=np.int64) % 1000), "s": pa.array(["row %d" % (i % 313) for i in range(n)]),
                  "v": pa.array(rng.integers(0, 1 << 20, n, dtype=np.int64))})
    path = str(tmp_path / "ok.arrows")
    _write(path, t, n)
    good = bytearray(open(path, "rb").read())
    (body_off, body_len), = _frames(bytes(good))
    outcomes = set()
    for trial in range(int(os.environ.get("MI_ZSTD_FUZZ_TRIALS", "32"))):
        bad = bytearray(good)
        if trial == 0:   # the declared uncompressed length of the first buffer that has one
            for at in range(body_off, body_off + body_len - 8, 8):
                v = struct.unpack_from("<q", bad, at)[0]
                if 0 < v < (1 << 31) and bad[at + 8: at + 12] == b"\x28\xb5\x2f\xfd":
                    struct.pack_into("<q", bad, at, v - 8)
                    break
        else:
            for _ in range(1 + trial % 5):
                at = body_off + 64 + int(rng.integers(0, body_len - 128))
                bad[at] = int(rng.integers(0, 256))
        p = str(tmp_path / ("bad_%d.arrows" % trial))
        open(p, "wb").write(bytes(bad))
        try:
            _device_scan(con, p, host_decompress="gpu")
            outcomes.add("ok")          # the damage hit a raw literal or padding
        except da.MiError as e:
            outcomes.add("error")
            assert e.code in (da._ffi.MI_EIO, da._ffi.MI_EINVAL), (trial, str(e))
    assert "error" in outcomes
    got, _ = _device_scan(con, path, host_decompress="gpu")    # the context still works
    assert got == [canon_python(c) for c in con.read_arrow(path, host_decompress=True).fetch_columns()]

