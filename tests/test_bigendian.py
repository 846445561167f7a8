"""Big-endian Arrow IPC streams (Schema.endianness = Big).  The reference reads them through nanoarrow's decoder
(src/ipc/stream_reader/base_stream_reader.cpp:68-69) and its integration tests read arrow-testing's 1.0.0-bigendian files
(test/python/test_integration.py:28,88), which are not in this image.  The fixtures here are made from the little-endian
golden files by tests/sanitize/make_bigendian.cpp; the tool is checked first: pyarrow, which swaps on load, must read its
output back as the original table."""
import os
import shutil
import subprocess

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pytest

import duckdb_arrow_amd as da

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["edge_types.arrows", "edge_types2.arrows", "edge_nested.arrows", "edge_dict.arrows", "lineitem_sf0_01_head.arrows", "ref_data/test.arrows"]


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("be") / "make_bigendian")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "sanitize", "make_bigendian.cpp"),
                    os.path.join(ROOT, "duckdb-arrow_amd", "csrc", "ipc_format.cpp"), "-o", exe], check=True, capture_output=True)
    return exe


def _big(tool, golden_dir, rel, tmp_path):
    out = str(tmp_path / (rel.replace("/", "_") + ".be"))
    subprocess.run([tool, os.path.join(golden_dir, rel), out], check=True)
    return out


def _same(a, b):
    # NaN payloads and views compare through python values
    return a.schema.names == b.schema.names and a.num_rows == b.num_rows and a.to_pydict().__repr__() == b.to_pydict().__repr__()


@pytest.mark.parametrize("rel", FILES)
def test_host_reader_swaps_big_endian_bodies(tool, golden_dir, tmp_path, rel):
    """CPU: the exported Arrow C stream of a big-endian stream holds the values of the little-endian original."""
    be = _big(tool, golden_dir, rel, tmp_path)
    want = ipc.open_stream(os.path.join(golden_dir, rel)).read_all()
    try:
        assert _same(ipc.open_stream(be).read_all(), want)      # the fixture is what a big-endian producer writes
    except pa.ArrowNotImplementedError:
        pass                                                    # pyarrow does not swap string views; the reader below does
    got = da.Reader(path=be).export_stream(accept_dictionaries=True).read_all()
    assert _same(got, want)
    data = np.fromfile(be, np.uint8)                            # caller-owned buffers: swapped in a private copy
    before = data.copy()
    got = da.Reader(buffers=[data]).export_stream(accept_dictionaries=True).read_all()
    assert _same(got, want) and np.array_equal(data, before)


@pytest.mark.gpu
@pytest.mark.parametrize("rel", FILES)
def test_scan_of_a_big_endian_stream_equals_the_little_endian_scan(tool, golden_dir, tmp_path, rel):
    be = _big(tool, golden_dir, rel, tmp_path)
    con = da.Connection(0)
    want = repr(con.read_arrow(os.path.join(golden_dir, rel), accept_dictionaries=True).fetch_columns())   # repr: NaN == NaN
    assert repr(con.read_arrow(be, accept_dictionaries=True).fetch_columns()) == want
    data = np.fromfile(be, np.uint8)
    assert repr(con.scan_arrow_ipc([data], accept_dictionaries=True).fetch_columns()) == want
    first = con.read_arrow(be, accept_dictionaries=True).columns[0]
    want_first = repr(con.read_arrow(os.path.join(golden_dir, rel), accept_dictionaries=True).project([first]).fetch_columns())
    assert repr(con.read_arrow(be, accept_dictionaries=True).project([first]).fetch_columns()) == want_first
