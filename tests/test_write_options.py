"""COPY ... (FORMAT ARROWS) option binding (ArrowWriteBind, src/writer/write_arrow_stream.cpp:54-125) -- CPU only.
Error texts are the reference's (test/sql/test_copy_to.test:48-56)."""
import ctypes as C

import pytest

import duckdb_arrow_amd as da
from duckdb_arrow_amd import _ffi


def opts():
    o = _ffi.WriteOptions()
    _ffi.check(_ffi.lib().mi_write_options_init(C.byref(o)))
    return o


def set_(o, k, v):
    _ffi.check(_ffi.lib().mi_write_options_set(C.byref(o), k.encode(), None if v is None else str(v).encode()))


def test_defaults():
    o = opts()
    _ffi.check(_ffi.lib().mi_write_options_finalize(C.byref(o)))
    assert o.row_group_size == 122880                 # write_arrow_stream.cpp:33
    assert o.row_group_size_bytes == 122880 * 1024    # :36,114-118
    assert o.row_groups_per_file == 0


def test_row_group_size_and_chunk_size_are_the_same_option():
    o = opts()
    set_(o, "ROW_GROUP_SIZE", 10)
    assert o.row_group_size == 10
    with pytest.raises(da.MiError, match="ROW_GROUP_SIZE and ROW_GROUP_SIZE_BYTES are mutually exclusive"):
        set_(o, "chunk_size", 100)


def test_row_group_size_bytes_needs_unordered_inserts():
    o = opts()
    set_(o, "row_group_size_bytes", 100)
    with pytest.raises(da.MiError, match=r"ROW_GROUP_SIZE_BYTES does not work while preserving insertion order. Use "
                                         r"\"SET preserve_insertion_order=false;\" to disable preserving insertion order."):
        _ffi.check(_ffi.lib().mi_write_options_finalize(C.byref(o)))
    o.preserve_insertion_order = 0
    _ffi.check(_ffi.lib().mi_write_options_finalize(C.byref(o)))
    assert o.row_group_size_bytes == 100


def test_memory_strings_and_argument_count():
    o = opts()
    set_(o, "row_group_size_bytes", "2KB")
    assert o.row_group_size_bytes == 2000
    set_(o, "row_group_size_bytes", "1 MiB")
    assert o.row_group_size_bytes == 1 << 20
    with pytest.raises(da.MiError, match="ROW_GROUPS_PER_FILE requires exactly one argument"):
        set_(o, "row_groups_per_file", None)
    set_(o, "row_groups_per_file", 3)
    assert o.row_groups_per_file == 3
    set_(o, "some_other_option", 1)  # not ours: ignored by the bind loop
