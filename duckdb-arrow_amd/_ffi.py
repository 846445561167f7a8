"""ctypes declarations of include/mi_arrow_ipc.h + include/mi_synth.h (the C ABI of libmi_arrow_ipc.so).

This is the reference-side binding shape a Python host would use; INTEGRATION.md shows the C++ (DuckDB extension)
equivalent.  No torch types, no oracle imports: the product path fails loudly when the library or the GPU is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi_arrow_ipc.so")

MI_OK, MI_EIO, MI_ENOMEM, MI_ENODEV, MI_EINVAL, MI_ERANGE, MI_ENODATA, MI_ENOTSUP = 0, 5, 12, 19, 22, 34, 61, 95
VECTOR_SIZE = 2048
NUM_KERNEL_CLASSES = 7

# enum mi_kind
K_COPY, K_BOOL, K_DEC128, K_DATE64, K_MUL_I32, K_MUL_I64, K_DIV_I64, K_STR32, K_STR64, K_DICT, K_FIXED_BINARY, \
    K_DURATION, K_INTERVAL_MONTHS, K_INTERVAL_MDN, K_NARROW, K_HALF_FLOAT, K_NULL, K_STRVIEW, K_LIST32, K_LIST64, \
    K_STRUCT = range(1, 22)
K_ENC_COPY, K_ENC_DEC128, K_ENC_BOOL, K_ENC_STR32, K_ENC_VALIDITY, K_ENC_LIST32 = 32, 33, 34, 35, 36, 37

ST_BAD_OFFSETS, ST_STRING_TOO_LARGE, ST_MUL_OVERFLOW, ST_INDEX_RANGE, ST_DECIMAL_RANGE, ST_OFFSET_OVERFLOW, ST_DICT_INDEX, ST_INTERNAL = \
    1, 2, 4, 8, 16, 32, 64, 128


class Field(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("timezone", C.c_char * 64), ("duck_type", C.c_char * 1024),
                ("format", C.c_char * 32), ("arrow_type", C.c_int32), ("bit_width", C.c_int32),
                ("is_signed", C.c_int32), ("precision", C.c_int32), ("scale", C.c_int32), ("unit", C.c_int32),
                ("byte_width", C.c_int32), ("nullable", C.c_int32), ("has_dictionary", C.c_int32),
                ("dict_index_bit_width", C.c_int32), ("dict_index_signed", C.c_int32), ("dict_id", C.c_int64),
                ("kind", C.c_int32), ("out_width", C.c_int32), ("param", C.c_int64), ("n_buffers", C.c_int32),
                ("flat_index", C.c_int32)]


class IpcBuffer(C.Structure):
    _fields_ = [("ptr", C.c_uint64), ("size", C.c_uint64)]


class BufferSpan(C.Structure):
    _fields_ = [("offset", C.c_int64), ("length", C.c_int64)]


class BatchNode(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("arrow_type", C.c_int32), ("kind", C.c_int32), ("out_width", C.c_int32),
                ("parent", C.c_int32), ("depth", C.c_int32), ("n_children", C.c_int32), ("first_span", C.c_int32),
                ("n_spans", C.c_int32), ("param", C.c_int64), ("length", C.c_int64), ("null_count", C.c_int64)]


class Batch(C.Structure):
    _fields_ = [("length", C.c_int64), ("body", C.c_void_p), ("body_size", C.c_int64),
                ("body_file_offset", C.c_int64), ("n_columns", C.c_int32), ("is_dictionary", C.c_int32),
                ("dict_id", C.c_int64), ("is_delta", C.c_int32), ("compression", C.c_int32),
                ("column_field", C.POINTER(C.c_int32)), ("null_count", C.POINTER(C.c_int64)),
                ("buffers", C.POINTER(BufferSpan)), ("n_nodes", C.c_int32), ("_pad", C.c_int32),
                ("nodes", C.POINTER(BatchNode)), ("node_spans", C.POINTER(BufferSpan)), ("column_node", C.POINTER(C.c_int32))]


class BatchIndexEntry(C.Structure):
    _fields_ = [("prefix_offset", C.c_int64), ("meta_len", C.c_int32), ("type", C.c_int32),
                ("body_offset", C.c_int64), ("body_len", C.c_int64), ("n_rows", C.c_int64)]


class ColTask(C.Structure):
    _fields_ = [("validity", C.c_void_p), ("buf1", C.c_void_p), ("buf2", C.c_void_p), ("out_data", C.c_void_p),
                ("out_validity", C.c_void_p), ("out_aux", C.c_void_p), ("ptr_base", C.c_uint64),
                ("nrows", C.c_int64), ("row_offset", C.c_int64), ("buf2_len", C.c_int64), ("param", C.c_int64),
                ("param2", C.c_int64), ("null_count", C.c_int64), ("kind", C.c_int32), ("flags", C.c_int32),
                ("depth", C.c_int32), ("_reserved", C.c_int32), ("sel", C.c_void_p), ("sel_count", C.c_void_p)]


class ScanOptions(C.Structure):
    _fields_ = [("union_by_name", C.c_int32), ("filename", C.c_int32), ("hive_partitioning", C.c_int32),
                ("rank", C.c_int32), ("world", C.c_int32), ("device_resident", C.c_int32),
                ("accept_dictionaries", C.c_int32), ("zero_copy_direct", C.c_int32), ("unset_all_valid", C.c_int32),
                ("filter_compact", C.c_int32), ("pipeline_depth", C.c_int32), ("host_decompress", C.c_int32),
                ("_reserved", C.c_int32 * 3)]


class RangeFilter(C.Structure):
    _fields_ = [("column", C.c_char_p), ("lo", C.c_int64), ("hi", C.c_int64)]


class FilterNode(C.Structure):
    _fields_ = [("op", C.c_int32), ("first_child", C.c_int32), ("n_children", C.c_int32), ("n_values", C.c_int32),
                ("column", C.c_char_p), ("value", C.c_int64), ("values", C.POINTER(C.c_int64)),
                ("str_value", C.c_char_p), ("str_len", C.c_int32), ("_pad", C.c_int32),
                ("str_values", C.POINTER(C.c_char_p)), ("str_lens", C.POINTER(C.c_int32))]


F_EQ, F_NE, F_LT, F_LE, F_GT, F_GE, F_IS_NULL, F_IS_NOT_NULL, F_IN, F_STARTS_WITH, F_AND, F_OR = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 16, 17


class HbmOptions(C.Structure):
    _fields_ = [("columns", C.POINTER(C.c_char_p)), ("n_columns", C.c_int32), ("accept_dictionaries", C.c_int32),
                ("zero_copy_direct", C.c_int32), ("unset_all_valid", C.c_int32), ("pointer_mode", C.c_int32),
                ("defer_arena", C.c_int32), ("array_align", C.c_int64), ("device_stream", C.c_void_p),
                ("device_arena", C.c_void_p), ("device_arena_bytes", C.c_int64)]


HBM_PTR_DEVICE, HBM_PTR_STREAM_OFFSET, HBM_PTR_HOST = 0, 1, 2


class HbmNode(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("kind", C.c_int32), ("out_width", C.c_int32), ("arrow_type", C.c_int32),
                ("depth", C.c_int32), ("parent", C.c_int32), ("batch", C.c_int32), ("param", C.c_int64),
                ("nrows", C.c_int64), ("null_count", C.c_int64), ("dict_id", C.c_int64), ("data_off", C.c_int64),
                ("valid_off", C.c_int64), ("alias_off", C.c_int64), ("ptr_base", C.c_uint64), ("first_span", C.c_int32),
                ("n_spans", C.c_int32), ("first_window", C.c_int32), ("n_windows", C.c_int32)]


class HbmBatch(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("body_off", C.c_int64), ("body_len", C.c_int64), ("arena_begin", C.c_int64),
                ("arena_end", C.c_int64), ("first_node", C.c_int32), ("n_nodes", C.c_int32), ("n_columns", C.c_int32),
                ("is_dictionary", C.c_int32), ("dict_id", C.c_int64)]


class HbmLayout(C.Structure):
    _fields_ = [("batches", C.POINTER(HbmBatch)), ("n_batches", C.c_int32), ("n_nodes", C.c_int32),
                ("nodes", C.POINTER(HbmNode)), ("spans", C.POINTER(BufferSpan)), ("windows", C.POINTER(C.c_int64)),
                ("arena_bytes", C.c_int64), ("stream_bytes", C.c_int64), ("n_rows", C.c_int64),
                ("device_stream", C.c_void_p), ("device_arena", C.c_void_p), ("n_tasks", C.c_int32), ("_pad", C.c_int32)]


class SumProductResult(C.Structure):
    _fields_ = [("sum_lo", C.c_uint64), ("sum_hi", C.c_int64), ("rows_scanned", C.c_int64), ("rows_selected", C.c_int64)]


class ArrowArrayStream(C.Structure):   # Arrow C stream interface: 4 callbacks + private_data
    _fields_ = [("get_schema", C.c_void_p), ("get_next", C.c_void_p), ("get_last_error", C.c_void_p),
                ("release", C.c_void_p), ("private_data", C.c_void_p)]


class Vector(C.Structure):
    pass


Vector._fields_ = [("data", C.c_void_p), ("validity", C.c_void_p), ("kind", C.c_int32), ("out_width", C.c_int32),
                   ("dictionary", C.c_void_p), ("dictionary_validity", C.c_void_p), ("dict_len", C.c_int64),
                   ("children", C.POINTER(Vector)), ("n_children", C.c_int32), ("validity_shift", C.c_int32),
                   ("count", C.c_int64), ("heap", C.c_void_p), ("heap_size", C.c_int64)]


class DataChunk(C.Structure):
    _fields_ = [("size", C.c_int64), ("n_columns", C.c_int32), ("file_index", C.c_int32),
                ("batch_index", C.c_int64), ("chunk_offset", C.c_int64), ("columns", C.POINTER(Vector)),
                ("sel", C.POINTER(C.c_uint32)), ("sel_count", C.c_int64), ("source_rows", C.c_int64)]


MAX_KV = 16


class WriteOptions(C.Structure):
    _fields_ = [("row_group_size", C.c_int64), ("row_group_size_bytes", C.c_int64),
                ("row_groups_per_file", C.c_int64), ("row_group_size_set", C.c_int32),
                ("row_group_size_bytes_set", C.c_int32), ("preserve_insertion_order", C.c_int32),
                ("n_kv_metadata", C.c_int32), ("kv_keys", (C.c_char * 64) * MAX_KV),
                ("kv_values", (C.c_char * 256) * MAX_KV), ("kv_value_lens", C.c_int32 * MAX_KV),
                ("arrow_large_buffer_size", C.c_int32), ("_reserved", C.c_int32)]


class ScanStats(C.Structure):
    _fields_ = [("record_batches", C.c_int64), ("lz4_batches_on_device", C.c_int64), ("h2d_bytes", C.c_int64),
                ("decompressed_bytes", C.c_int64), ("lz4_blocks", C.c_int64), ("lz4_parse_rounds", C.c_int64),
                ("lz4_parse_rounds_max", C.c_int64), ("zstd_batches_on_device", C.c_int64), ("d2h_bytes", C.c_int64),
                ("aliased_bytes", C.c_int64)]


class SynthOptions(C.Structure):
    _fields_ = [("scale_factor", C.c_double), ("seed", C.c_uint64), ("rows_per_batch", C.c_int64),
                ("n_rows", C.c_int64), ("first_row", C.c_int64), ("with_validity", C.c_int32), ("n_threads", C.c_int32)]


# every symbol include/mi_arrow_ipc.h and include/mi_synth.h declare: name -> (restype, argtypes)
P = C.c_void_p
PP = C.POINTER(C.c_void_p)
SIGNATURES = {
    "mi_last_error": (C.c_char_p, []),
    "mi_version": (C.c_char_p, []),
    "mi_nanoarrow_version": (C.c_char_p, []),
    "mi_reader_open_file": (C.c_int, [C.c_char_p, PP]),
    "mi_reader_open_buffers": (C.c_int, [C.POINTER(IpcBuffer), C.c_int32, PP]),
    "mi_reader_close": (None, [P]),
    "mi_reader_schema": (C.c_int, [P, C.POINTER(Field), C.c_int32, C.POINTER(C.c_int32)]),
    "mi_reader_schema_metadata": (C.c_int, [P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32),
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mi_reader_set_projection": (C.c_int, [P, C.POINTER(C.c_char_p), C.c_int32]),
    "mi_reader_next_batch": (C.c_int, [P, C.c_int32, C.POINTER(Batch)]),
    "mi_reader_progress": (C.c_double, [P]),
    "mi_reader_index": (C.c_int, [P, C.POINTER(C.POINTER(BatchIndexEntry)), C.POINTER(C.c_int32)]),
    "mi_ctx_create": (C.c_int, [C.c_int32, PP]),
    "mi_ctx_destroy": (None, [P]),
    "mi_ctx_numa": (C.c_int, [P, C.POINTER(C.c_int32), C.c_char_p, C.c_int32]),
    "mi_device_count": (C.c_int, []),
    "mi_plan_create": (C.c_int, [P, C.POINTER(ColTask), C.c_int32, PP]),
    "mi_plan_destroy": (None, [P]),
    "mi_plan_launch": (C.c_int, [P, P]),
    "mi_plan_status": (C.c_int, [P, C.POINTER(C.c_uint32)]),
    "mi_plan_stats": (C.c_int, [P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                C.POINTER(C.c_int64)]),
    "mi_plan_class_stats": (C.c_int, [P, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_char_p)]),
    "mi_plan_launch_timed": (C.c_int, [P, P, C.POINTER(C.c_float)]),
    "mi_plan_null_counts": (C.c_int, [P, C.POINTER(C.c_int64), C.c_int32]),
    "mi_status_to_error": (C.c_int, [C.c_uint32]),
    "mi_filter_range": (C.c_int, [P, P, C.c_int32, P, C.c_int64, C.c_int64, C.c_int64, P, P, P]),
    "mi_reader_export_stream": (C.c_int, [P, C.c_int32, C.POINTER(ArrowArrayStream)]),
    "mi_scan_open_files": (C.c_int, [P, C.POINTER(C.c_char_p), C.c_int32, C.POINTER(ScanOptions), PP]),
    "mi_scan_open_buffers": (C.c_int, [P, C.POINTER(IpcBuffer), C.c_int32, C.POINTER(ScanOptions), PP]),
    "mi_scan_close": (None, [P]),
    "mi_scan_bind": (C.c_int, [P, C.POINTER(Field), C.c_int32, C.POINTER(C.c_int32)]),
    "mi_scan_init": (C.c_int, [P, C.POINTER(C.c_char_p), C.c_int32]),
    "mi_scan_set_filter_range": (C.c_int, [P, C.c_char_p, C.c_int64, C.c_int64]),
    "mi_scan_set_filter": (C.c_int, [P, C.POINTER(FilterNode), C.c_int32, C.c_int32]),
    "mi_scan_open_files_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_char_p), C.c_int32,
                                           C.POINTER(ScanOptions), PP]),
    "mi_hbm_open": (C.c_int, [P, P, C.c_int64, C.POINTER(HbmOptions), PP]),
    "mi_hbm_close": (None, [P]),
    "mi_hbm_set_arena": (C.c_int, [P, P, C.c_int64]),
    "mi_hbm_layout_get": (C.c_int, [P, C.POINTER(HbmLayout)]),
    "mi_hbm_launch": (C.c_int, [P, P]),
    "mi_hbm_launch_timed": (C.c_int, [P, P, C.POINTER(C.c_float)]),
    "mi_hbm_status": (C.c_int, [P, C.POINTER(C.c_uint32)]),
    "mi_hbm_stats": (C.c_int, [P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mi_hbm_class_stats": (C.c_int, [P, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_char_p)]),
    "mi_hbm_fetch": (C.c_int, [P, C.c_int32, C.c_int64, C.c_int64, P]),
    "mi_scan_next": (C.c_int, [P, C.POINTER(DataChunk)]),
    "mi_scan_count": (C.c_int, [P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mi_scan_sum_product": (C.c_int, [P, C.c_char_p, C.c_char_p, C.POINTER(RangeFilter), C.c_int32, C.POINTER(SumProductResult)]),
    "mi_scan_progress": (C.c_double, [P]),
    "mi_scan_get_stats": (C.c_int, [P, P]),
    "mi_write_options_init": (C.c_int, [C.POINTER(WriteOptions)]),
    "mi_write_options_set": (C.c_int, [C.POINTER(WriteOptions), C.c_char_p, C.c_char_p]),
    "mi_write_options_add_kv": (C.c_int, [C.POINTER(WriteOptions), C.c_char_p, C.c_char_p, C.c_int32]),
    "mi_write_options_finalize": (C.c_int, [C.POINTER(WriteOptions)]),
    "mi_encode_schema": (C.c_int, [C.POINTER(Field), C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "mi_writer_open": (C.c_int, [P, C.c_char_p, C.POINTER(Field), C.c_int32, C.POINTER(WriteOptions), PP]),
    "mi_writer_sink": (C.c_int, [P, C.POINTER(DataChunk)]),
    "mi_writer_sink_scan": (C.c_int, [P, P, C.POINTER(C.c_int64)]),
    "mi_writer_local_create": (C.c_int, [P, PP]),
    "mi_writer_local_sink": (C.c_int, [P, C.POINTER(DataChunk)]),
    "mi_writer_local_combine": (C.c_int, [P]),
    "mi_writer_local_destroy": (None, [P]),
    "mi_writer_finalize": (C.c_int, [P]),
    "mi_writer_close": (None, [P]),
    "mi_writer_row_groups": (C.c_int64, [P]),
    "mi_writer_file_size": (C.c_int64, [P]),
    "mi_writer_rotate_next_file": (C.c_int, [P, C.c_int64]),
    "mi_writer_append_message": (C.c_int, [P, P, C.c_int64]),
    "mi_ipc_serializer_create": (C.c_int, [P, C.POINTER(Field), C.c_int32, PP]),
    "mi_ipc_serialize_schema": (C.c_int, [P, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "mi_ipc_serialize_chunks": (C.c_int, [P, C.POINTER(DataChunk), C.c_int32, C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_int64)]),
    "mi_synth_lineitem_layout": (C.c_int, [C.POINTER(SynthOptions), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int64]),
    "mi_synth_lineitem_fill": (C.c_int, [C.POINTER(SynthOptions), P, C.c_int64]),
}

_lib = None


class MiError(Exception):
    """Failure reported by the C ABI: .code is the errno-style status, str() the reference-compatible message."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


def lib():
    """Loads libmi_arrow_ipc.so (built in-tree by `make -C duckdb-arrow_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: build it with `make -C %s/csrc` (the product path has no fallback)"
                              % (LIB_PATH, _HERE))
        # One HIP/HSA runtime per process: PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64 and a second
        # runtime cannot open the GPU ("No HIP GPUs are available").  When torch is used for device memory (tests,
        # bench), load it first so that this library binds to the runtime torch already brought in.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != MI_OK:
        raise MiError(rc, lib().mi_last_error().decode("utf-8", "replace"))
    return rc
