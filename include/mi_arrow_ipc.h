/*
 * mi_arrow_ipc.h -- C ABI of the MI355X-native Arrow IPC scan / encode path.
 *
 * This is the drop-in boundary for the ONE hot path of pdet/duckdb-arrow (the DuckDB `nanoarrow` extension):
 * Arrow IPC record-batch bodies -> DuckDB vectors (and the inverse for COPY TO / to_arrow_ipc).  The library behind
 * it (duckdb-arrow_amd/csrc -> libmi_arrow_ipc.so) is host C++ (IPC framing, flatbuffer metadata, pinned staging,
 * stream scheduling) plus hand-written HIP kernels for gfx950.  No C++ or torch types cross this line: plain
 * pointers, sizes and POD structs only.  Every entry point returns an errno-style int (0 = ok) and never throws;
 * the message of the last failure is available from mi_last_error() -- the same shape as the reference's C stream
 * boundary (src/include/ipc/array_stream.hpp:29-48: exceptions -> EIO/EINVAL/ENOMEM + last_msg).
 *
 * Each section names the reference interface it replaces (file:line relative to the reference repository).
 * INTEGRATION.md shows the DuckDB-side glue a maintainer would write on top of these calls.
 */
#ifndef MI_ARROW_IPC_H
#define MI_ARROW_IPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_ABI_VERSION 2

/* ---------------------------------------------------------------------------------------------------------
 * Status codes.  Same errno values the reference maps its exceptions to at the Arrow C stream boundary
 * (src/include/ipc/array_stream.hpp:33-46) plus ENODATA for end-of-stream (ipc_file_stream_reader.cpp:63-66).
 * ------------------------------------------------------------------------------------------------------- */
#define MI_OK 0
#define MI_EIO 5        /* IOException: framing / file errors */
#define MI_ENOMEM 12    /* allocation failure (host, pinned or HBM) */
#define MI_ENODEV 19    /* no usable HIP device: the product path never falls back to the CPU */
#define MI_EINVAL 22    /* InternalException / InvalidInputException / BinderException */
#define MI_ENODATA 61   /* end of stream */
#define MI_ERANGE 34    /* ConversionException: value out of range during transcode */
#define MI_ENOTSUP 95   /* NotImplementedException: type or feature outside the path */

/* Message of the last failing call on this thread ("" when none). Never NULL. */
const char* mi_last_error(void);
/* ABI + build info: "mi_arrow_ipc <abi> gfx950 <nanoarrow-compat-version>".  The last token is what the
 * reference's nanoarrow_version() scalar function returns (src/nanoarrow_extension.cpp:20-31, test/sql/nanoarrow.test:15-18). */
const char* mi_version(void);
const char* mi_nanoarrow_version(void);

/* ---------------------------------------------------------------------------------------------------------
 * DuckDB physical layouts produced / consumed by the path.
 * ------------------------------------------------------------------------------------------------------- */
#define MI_VECTOR_SIZE 2048 /* STANDARD_VECTOR_SIZE (src/writer/to_arrow_ipc.cpp:21) */

typedef struct mi_string_t { /* duckdb::string_t, 16 bytes */
  union {
    struct { uint32_t length; char prefix[4]; uint64_t ptr; } pointer; /* length > 12: first 4 bytes + payload address */
    struct { uint32_t length; char inlined[12]; } inlined;             /* length <= 12: payload, zero padded */
  } value;
} mi_string_t;

typedef struct mi_hugeint_t { uint64_t lower; int64_t upper; } mi_hugeint_t;              /* duckdb::hugeint_t */
typedef struct mi_interval_t { int32_t months; int32_t days; int64_t micros; } mi_interval_t; /* duckdb::interval_t */
typedef uint64_t mi_validity_t; /* bit = 1: valid, LSB first (same polarity and order as Arrow) */
typedef uint32_t mi_sel_t;

/* ---------------------------------------------------------------------------------------------------------
 * Schema model.  Replaces ArrowSchema + ArrowTableFunction::PopulateArrowTableType as used by
 * src/file_scanner/arrow_file_scan.cpp:13-22 and src/scanner/scan_arrow_ipc.cpp:34-44.
 * ------------------------------------------------------------------------------------------------------- */
/* Arrow type ids = Schema.fbs `Type` union tags */
enum mi_arrow_type {
  MI_AT_NONE = 0, MI_AT_NULL = 1, MI_AT_INT = 2, MI_AT_FLOAT = 3, MI_AT_BINARY = 4, MI_AT_UTF8 = 5, MI_AT_BOOL = 6,
  MI_AT_DECIMAL = 7, MI_AT_DATE = 8, MI_AT_TIME = 9, MI_AT_TIMESTAMP = 10, MI_AT_INTERVAL = 11, MI_AT_LIST = 12,
  MI_AT_STRUCT = 13, MI_AT_UNION = 14, MI_AT_FIXED_BINARY = 15, MI_AT_FIXED_LIST = 16, MI_AT_MAP = 17,
  MI_AT_DURATION = 18, MI_AT_LARGE_BINARY = 19, MI_AT_LARGE_UTF8 = 20, MI_AT_LARGE_LIST = 21, MI_AT_RUN_END = 22,
  MI_AT_BINARY_VIEW = 23, MI_AT_UTF8_VIEW = 24
};

/* Transcode kinds: which kernel converts a column. */
enum mi_kind {
  MI_K_COPY = 1,          /* K3a fixed-width direct; param = width in bytes (1,2,4,8,16) */
  MI_K_BOOL = 2,          /* K2  bit -> byte */
  MI_K_DEC128 = 3,        /* K3b decimal128 -> int16/32/64; param = out width (2,4,8) */
  MI_K_DATE64 = 4,        /* K3c date64[ms] -> date32 */
  MI_K_MUL_I32 = 5,       /* K3c int32 * param -> int64 (time32[s], time32[ms]) */
  MI_K_MUL_I64 = 6,       /* K3c int64 * param -> int64, overflow checked (timestamp[s|ms] with tz) */
  MI_K_DIV_I64 = 7,       /* K3c int64 / param (time64[ns], timestamp[ns] with tz) */
  MI_K_STR32 = 8,         /* K4a utf8/binary, int32 offsets -> string_t */
  MI_K_STR64 = 9,         /* K4b large_utf8/large_binary, int64 offsets -> string_t */
  MI_K_DICT = 10,         /* K5  dictionary indices -> sel_t; param = idx width | signed<<8; param2 = dict_len */
  MI_K_FIXED_BINARY = 11, /* K4c fixed_size_binary -> string_t; param = byte width */
  MI_K_DURATION = 12,     /* K3c duration -> interval_t; param = factor (>0 multiply, <0 divide by -param) */
  MI_K_INTERVAL_MONTHS = 13, /* K3c interval[months] int32 -> interval_t{months,0,0} */
  MI_K_INTERVAL_MDN = 14, /* K3c interval[month_day_nano] -> interval_t{months, days, nanos/1000} */
  MI_K_NARROW = 15,       /* K3b decimal32/64 -> int16/32 (valid rows); param = src width | dst width << 8 */
  MI_K_HALF_FLOAT = 16,   /* float16 -> float32 (DuckDB FLOAT) */
  MI_K_NULL = 17,         /* arrow null type: no buffers, all rows NULL (1-byte placeholder data) */
  MI_K_STRVIEW = 18,      /* K4c utf8_view / binary_view -> string_t; buf2 = table of {u64 address, i64 length} per variadic
                             data buffer (addresses as the consumer sees them), buf2_len = number of buffers */
  MI_K_LIST32 = 19,       /* list / map, int32 offsets -> list_entry_t{u64 offset, u64 length}; offsets are relative to the
                             start of the row's top-level 2048-row window (ConvertArrowListOffsets); param = child length;
                             buf2 = i64 window starts in this list's row space (NULL: windows are the 2048-row tiles) */
  MI_K_LIST64 = 20,       /* large_list */
  MI_K_STRUCT = 21,       /* struct / fixed_size_list: validity only (children are their own tasks) */
  /* encode direction (K7), used by mi_encode_* plans */
  MI_K_ENC_COPY = 32,     /* K7b fixed-width copy; param = width */
  MI_K_ENC_DEC128 = 33,   /* K7b int16/32/64 -> decimal128 sign extension; param = in width */
  MI_K_ENC_BOOL = 34,     /* K7c byte -> bit, bits start as 1 */
  MI_K_ENC_STR32 = 35,    /* K7d string_t -> int32 offsets + data (flags bit 0: int64 offsets, LargeUtf8 / LargeList) */
  MI_K_ENC_VALIDITY = 36, /* K7a alone: validity words -> always-present bitmap + NULL count (struct / fixed_size_list node) */
  MI_K_ENC_LIST32 = 37    /* list_entry_t{u64 offset, u64 length} rows -> bitmap + int32 Arrow offsets (running sum of the
                           * lengths of the valid rows, NULL rows repeat the offset: ArrowListData::AppendOffsets); the
                           * child rows are a node of their own, gathered in list order */
};

typedef struct mi_field {
  char name[128];
  char timezone[64];
  char duck_type[1024];   /* DuckDB logical type as `typeof` prints it: BIGINT, DECIMAL(15,2), TIMESTAMP WITH TIME ZONE,
                           * STRUCT(a INTEGER, b VARCHAR[])[] ... (nested types need the room; longer ones are refused) */
  char format[32];        /* Arrow C data interface format string: "l", "u", "d:15,2", "tdD", "tsu:UTC" */
  int32_t arrow_type;     /* enum mi_arrow_type */
  int32_t bit_width, is_signed, precision, scale, unit, byte_width, nullable;
  int32_t has_dictionary, dict_index_bit_width, dict_index_signed;
  int64_t dict_id;
  int32_t kind;           /* enum mi_kind that decodes this column, 0 when unsupported on the path */
  int32_t out_width;      /* bytes per row of the DuckDB vector */
  int64_t param;          /* kind parameter */
  int32_t n_buffers;      /* buffers this field owns in a RecordBatch */
  int32_t flat_index;     /* depth-first flattened field index (IPCStreamReader::CountFields, base_stream_reader.cpp:271-277) */
} mi_field;

/* ---------------------------------------------------------------------------------------------------------
 * Host IPC readers.  Replace IPCStreamReader / IPCFileStreamReader / IPCBufferStreamReader
 * (src/ipc/stream_reader/{base,ipc_file,ipc_buffer}_stream_reader.cpp) and the stream factories
 * (src/ipc/stream_factory.cpp:41-63).  Host only: usable without a GPU.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct mi_reader mi_reader;

/* == ArrowIPCBuffer{ptr,size} (src/include/table_function/scan_arrow_ipc.hpp:19-23); caller keeps the memory alive. */
typedef struct mi_ipc_buffer { uint64_t ptr; uint64_t size; } mi_ipc_buffer;

/* FileIPCStreamFactory::InitReader (stream_factory.cpp:57-63) */
int mi_reader_open_file(const char* path, mi_reader** out);
/* BufferIPCStreamFactory::InitReader (stream_factory.cpp:46-50) */
int mi_reader_open_buffers(const mi_ipc_buffer* buffers, int32_t n_buffers, mi_reader** out);
void mi_reader_close(mi_reader* r);

/* IPCStreamReader::GetBaseSchema (base_stream_reader.cpp:52-74): reads the Schema message once.
 * Fills up to `cap` top-level fields; *n_fields = number of top-level fields in the file. */
int mi_reader_schema(mi_reader* r, mi_field* fields, int32_t cap, int32_t* n_fields);
/* Schema-level custom metadata (kv_metadata COPY option, arrow_stream_writer.cpp:26-44). idx in [0, count). */
int mi_reader_schema_metadata(mi_reader* r, int32_t idx, const char** key, int32_t* key_len, const char** value,
                              int32_t* value_len, int32_t* count);
/* IPCStreamReader::SetColumnProjection (base_stream_reader.cpp:146-212), same error strings:
 * "Can't request zero fields projected from IpcStreamReader", "Field 'x' does not exist in IPC file schema",
 * "Field 'x' refers to a duplicate column name in IPC file schema". */
int mi_reader_set_projection(mi_reader* r, const char* const* names, int32_t n);

typedef struct mi_buffer_span { int64_t offset; int64_t length; } mi_buffer_span; /* Buffer{offset,length} in the body */

/* One field node of a record batch (depth-first order): nested columns (list / struct / map / fixed_size_list) and
 * string views own more than the three buffers of a flat column. */
typedef struct mi_batch_node {
  char name[64];
  int32_t arrow_type;      /* enum mi_arrow_type */
  int32_t kind;            /* enum mi_kind (0 = not decodable) */
  int32_t out_width;
  int32_t parent;          /* node index, -1 for a top-level column */
  int32_t depth;
  int32_t n_children;
  int32_t first_span;      /* index into mi_batch.node_spans */
  int32_t n_spans;         /* validity, buffer 1, buffer 2, ... (+ variadic data buffers of views) */
  int64_t param;           /* kind parameter (fixed_size_list: list size) */
  int64_t length;
  int64_t null_count;
} mi_batch_node;

/* One decoded RecordBatch message: what IPCStreamReader::GetNextBatch (base_stream_reader.cpp:86-144) hands to
 * DuckDB as an ArrowArray, flattened.  Pointers stay valid until the next mi_reader_next_batch / close. */
typedef struct mi_batch {
  int64_t length;               /* rows */
  const uint8_t* body;          /* host address of the message body (file readers: an internal buffer) */
  int64_t body_size;
  int64_t body_file_offset;     /* position of the body in the file / buffer (for progress + sharding) */
  int32_t n_columns;            /* projected (or all top-level) columns */
  int32_t is_dictionary;        /* 1: this is a DictionaryBatch for dict_id */
  int64_t dict_id;
  int32_t is_delta;
  int32_t compression;          /* -1 none, 0 LZ4_FRAME, 1 ZSTD */
  const int32_t* column_field;  /* [n_columns] index into the base schema's top-level fields */
  const int64_t* null_count;    /* [n_columns] */
  const mi_buffer_span* buffers;/* [n_columns * 3]: validity, buf1, buf2 (length 0 when absent) */
  int32_t n_nodes;              /* every node of the projected columns, children right after their parent's subtree order */
  int32_t _pad;
  const mi_batch_node* nodes;
  const mi_buffer_span* node_spans;
  const int32_t* column_node;   /* [n_columns] node index of every column */
} mi_batch;

/* Returns MI_OK and fills *out, or MI_ENODATA at end of stream (EOS marker, truncated stream, or buffers
 * exhausted).  Framing errors are MI_EIO with the reference's messages ("Expected continuation token
 * (0xFFFFFFFF) but got N", "Expected metadata size >= 0 but got N", "Expected RecordBatch Arrow IPC message
 * but got Schema").  `accept_dictionaries` = 0 reproduces the reference (RecordBatch only); 1 also returns
 * DictionaryBatch messages (BASELINE config 5, beyond the reference). */
int mi_reader_next_batch(mi_reader* r, int32_t accept_dictionaries, mi_batch* out);
/* IPCFileStreamReader::GetProgress (ipc_file_stream_reader.cpp:22-29): percent of the file consumed. */
double mi_reader_progress(mi_reader* r);
/* Batch index without reading bodies: walks message headers (stream format) or the footer (file format), so
 * that record batches can be sharded over GPUs.  Arrays are owned by the reader. */
typedef struct mi_batch_index_entry { int64_t prefix_offset; int32_t meta_len; int32_t type; int64_t body_offset;
                                      int64_t body_len; int64_t n_rows; } mi_batch_index_entry;
int mi_reader_index(mi_reader* r, const mi_batch_index_entry** entries, int32_t* n);

/* The reader as an Arrow C stream -- the narrowest seam of the reference: IpcArrayStream::{GetSchema, GetNext, Wrap}
 * (src/ipc/array_stream.cpp:11-26, src/include/ipc/array_stream.hpp:29-48) behind ArrowIPCStreamFactory::Produce
 * (src/ipc/stream_factory.cpp:14-30), i.e. what DuckDB's own arrow scan consumes.  The reader (with its projection) moves
 * into the stream: afterwards `r` only accepts mi_reader_close.  Arrays are zero-copy views of the message bodies, each
 * keeps its body alive until released; get_next errors are EIO / ENOTSUP / ENOMEM + get_last_error, like the reference.
 * Host only.  The structs are the Arrow C data / stream interface (apache/arrow: format/CDataInterface.rst). */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
#endif
#ifndef ARROW_C_STREAM_INTERFACE
#define ARROW_C_STREAM_INTERFACE
struct ArrowArrayStream {
  int (*get_schema)(struct ArrowArrayStream*, struct ArrowSchema* out);
  int (*get_next)(struct ArrowArrayStream*, struct ArrowArray* out);
  const char* (*get_last_error)(struct ArrowArrayStream*);
  void (*release)(struct ArrowArrayStream*);
  void* private_data;
};
#endif
int mi_reader_export_stream(mi_reader* r, int32_t accept_dictionaries, struct ArrowArrayStream* out);

/* ---------------------------------------------------------------------------------------------------------
 * Device context + transcode plans.  Replace the per-value loops of DuckDB core that the reference calls:
 * ArrowTableFunction::ArrowScanFunction -> ArrowToDuckDB (call sites src/scanner/scan_arrow_ipc.cpp:56,
 * src/file_scanner/arrow_file_scan.cpp:68-72) and ArrowConverter::ToArrowArray / ArrowAppender (call sites
 * src/writer/column_data_collection_serializer.cpp:85, src/writer/to_arrow_ipc.cpp:134-141).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct mi_ctx mi_ctx;
typedef struct mi_plan mi_plan;

/* One context per (GPU, worker).  Fails with MI_ENODEV when no HIP device is present. */
int mi_ctx_create(int32_t device_id, mi_ctx** out);
void mi_ctx_destroy(mi_ctx* ctx);
int mi_device_count(void);
/* Where the context's GPU hangs in the host: its NUMA node (-1 = the platform does not say) and that node's CPUs as the
 * kernel prints them ("0-63,128-191", NUL-terminated, truncated to cap).  The library's own host threads (read-ahead,
 * I/O pool) run there and allocate their pinned buffers there by themselves (MI_NUMA_BIND=0 turns that off); a host
 * process that wants its own threads and page cache on the same node (the threads that write the files a scan will read,
 * the consumer of the chunks) binds them with this.  DuckDB has no such seam: the reference reads through DuckDB's
 * FileSystem on whatever thread the scheduler picks (src/file_scanner/arrow_multi_file_info.cpp:77-86). */
int mi_ctx_numa(mi_ctx* ctx, int32_t* node, char* cpulist, int32_t cap);

/* One column of one record batch.  All pointers are DEVICE addresses (HBM).  Arrow buffers must be 8-byte
 * aligned and padded to a multiple of 8 bytes, as the IPC format guarantees for message bodies. */
typedef struct mi_col_task {
  const void* validity;   /* Arrow validity bitmap, or NULL when absent (buffer length 0) */
  const void* buf1;       /* fixed-width data / offsets / dictionary indices / (encode) DuckDB vector data */
  const void* buf2;       /* string data / (encode) string heap: the bytes long string_t rows point into (ptr - ptr_base).
                             Any layout is encoded exactly; when the long strings of 64 consecutive rows lie there as they
                             will lie in the Arrow data buffer (an Arrow data buffer itself, or a heap staged in row order
                             with the slots of inline strings left open) they are moved as one coalesced copy. */
  void* out_data;         /* DuckDB vector data, nrows * out_width bytes (encode: Arrow buffer 1) */
  void* out_validity;     /* mi_validity_t[ceil(nrows/64)] or NULL to skip (encode: Arrow bitmap) */
  void* out_aux;          /* encode: Arrow buffer 2 (string data); decode: validity words of the PARENT vector when NULLs
                             propagate from it (struct / fixed_size_list parents), else NULL */
  uint64_t ptr_base;      /* address the consumer will see for byte 0 of buf2 (string_t long-string pointers) */
  int64_t nrows;
  int64_t row_offset;     /* Arrow array offset: first row inside the buffers (0 for IPC-decoded arrays) */
  int64_t buf2_len;       /* bytes in buf2 (offset validation) */
  int64_t param;
  int64_t param2;
  int64_t null_count;     /* 0: bitmap ignored, all rows valid (GetValidityMask) */
  int32_t kind;           /* enum mi_kind */
  int32_t flags;          /* decode with out_aux: parent row = row / flags (0 or 1: same row; n: fixed_size_list of n) */
  int32_t depth;          /* nesting depth: tasks run depth by depth so a child sees its parent's finished validity */
  int32_t _reserved;
  /* Gather mode (late materialisation through a pushed-down filter's selection vector): when `sel` is set, only the rows
   * sel[2048 w + i], i < sel_count[w], of every 2048-row window w are decoded; they land compacted at the start of the
   * window's vector slot (out_data + 2048 w * width, validity word 32 w).  Top-level flat kinds only. */
  const void* sel;        /* mi_sel_t[nrows]: ascending window-relative row indices per window (device) */
  const void* sel_count;  /* uint32_t[ceil(nrows / 2048)] (device) */
} mi_col_task;

/* Error bits a plan accumulates on the device (polled by mi_plan_status). */
#define MI_ST_BAD_OFFSETS 1u      /* offsets decreasing / negative / past the data buffer (FULL validation) */
#define MI_ST_STRING_TOO_LARGE 2u /* "DuckDB does not support Strings over 4GB" */
#define MI_ST_MUL_OVERFLOW 4u     /* "Could not convert ... to Microsecond" */
#define MI_ST_INDEX_RANGE 8u      /* "DuckDB only supports indices that fit on an uint32" */
#define MI_ST_DECIMAL_RANGE 16u   /* decimal128 value does not fit the declared precision's physical type */
#define MI_ST_OFFSET_OVERFLOW 32u /* encode: int32 offsets exceed INT32_MAX ("SET arrow_large_buffer_size=true") */
#define MI_ST_DICT_INDEX 64u      /* a valid row's dictionary index is >= the dictionary length (FULL validation) */
#define MI_ST_INTERNAL 128u       /* a kernel gave up waiting for another workgroup (bounded spin): results are not valid */
#define MI_ST_DECOMPRESS 256u     /* a compressed buffer is malformed or does not expand to its declared length (EIO,
                                   * base_stream_reader.cpp:24-29) */

/* Uploads the task table to HBM (descriptor table + tile index) and returns a reusable plan.  One plan =
 * any number of (batch, column) tasks = ONE fused kernel launch per mi_plan_launch. */
int mi_plan_create(mi_ctx* ctx, const mi_col_task* tasks, int32_t n_tasks, mi_plan** out);
void mi_plan_destroy(mi_plan* plan);
/* Enqueues the fused transcode on `stream` (a hipStream_t; NULL = the context's own stream).  Asynchronous. */
int mi_plan_launch(mi_plan* plan, void* stream);
/* Waits for the stream the plan last ran on, returns the accumulated MI_ST_* bits and resets them. */
int mi_plan_status(mi_plan* plan, uint32_t* status_bits);
/* Algorithmic bytes of one launch: Arrow buffer bytes consumed + DuckDB vector bytes produced, and tiles. */
int mi_plan_stats(const mi_plan* plan, int64_t* bytes_read, int64_t* bytes_written, int64_t* rows, int64_t* tiles);
/* Kernel classes of a plan: 0 copy, 1 dec128, 2 string, 3 misc, 4 encode-fixed, 5 encode-string, 6 gather (tasks with `sel`). */
#define MI_NUM_KERNEL_CLASSES 7
/* Per-class share of mi_plan_stats (0 for classes the plan does not use) + the kernel's name as rocprof prints it. */
int mi_plan_class_stats(const mi_plan* plan, int32_t kernel_class, int64_t* bytes_read, int64_t* bytes_written,
                        int64_t* rows, int64_t* tiles, const char** kernel_name);
/* Like mi_plan_launch, but brackets every class launch with HIP events on `stream` and, after synchronising,
 * returns the device time of each class in milliseconds (0 for unused classes).  Measurement aid for bench.py. */
int mi_plan_launch_timed(mi_plan* plan, void* stream, float* ms_per_class /* [MI_NUM_KERNEL_CLASSES] */);
/* Encode plans: NULL count per task (FieldNode.null_count), in the order the tasks were given. Waits for the plan. */
int mi_plan_null_counts(mi_plan* plan, int64_t* out, int32_t n_tasks);
/* Maps a status word to the errno + message the reference would raise. Returns MI_OK for 0. */
int mi_status_to_error(uint32_t status_bits);

/* ---------------------------------------------------------------------------------------------------------
 * HBM-resident streams (the mode the roofline is measured in).  A whole Arrow IPC stream is uploaded
 * once; mi_hbm_open parses every message on the host, lays out one DuckDB vector array per (record batch, field node) in
 * one output arena and builds ONE plan over all of them: a launch is a handful of kernels however many record batches
 * the stream holds.  The layout is the scan operator's own planner (one planner for both modes).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct mi_hbm mi_hbm;

enum mi_hbm_pointer_mode {
  MI_HBM_PTR_DEVICE = 0,         /* string_t long pointers are device addresses inside the resident stream (GPU consumers) */
  MI_HBM_PTR_STREAM_OFFSET = 1,  /* ... are byte positions inside the stream (position-independent; what the parity tests compare) */
  MI_HBM_PTR_HOST = 2            /* ... are host addresses inside `host_stream` (host consumers after a D2H of the vectors) */
};

typedef struct mi_hbm_options {
  const char* const* columns;    /* projection by name (IPCStreamReader::SetColumnProjection), NULL = all columns */
  int32_t n_columns;
  int32_t accept_dictionaries;   /* decode DictionaryBatch messages + dictionary-encoded columns (last batch per id; no deltas) */
  int32_t zero_copy_direct;      /* plain fixed-width columns without NULLs get no task: their vector IS the Arrow buffer in
                                  * HBM (mi_hbm_node.alias_off), like the reference's DirectConversion */
  int32_t unset_all_valid;       /* columns with null_count == 0 get no validity words (mi_hbm_node.valid_off = -1), like
                                  * the reference, which leaves the ValidityMask of such a vector unset */
  int32_t pointer_mode;          /* enum mi_hbm_pointer_mode */
  int32_t defer_arena;           /* 1: do not allocate the output arena; the caller binds one with mi_hbm_set_arena */
  int64_t array_align;           /* alignment of every output array inside the arena, power of two; 0 = 65536 */
  void* device_stream;           /* optional: the stream already resident in HBM (caller-owned, >= size + 64 bytes readable);
                                  * NULL = the library uploads `host_stream` into its own allocation */
  void* device_arena;            /* optional: caller-owned output arena of device_arena_bytes (see mi_hbm_layout.arena_bytes) */
  int64_t device_arena_bytes;
} mi_hbm_options;

/* One field node of one message (depth first; nodes of a batch are consecutive; children name their parent). */
typedef struct mi_hbm_node {
  char name[64];
  int32_t kind;                  /* enum mi_kind */
  int32_t out_width;
  int32_t arrow_type;            /* enum mi_arrow_type */
  int32_t depth;
  int32_t parent;                /* global node index, -1 = a top-level column */
  int32_t batch;                 /* index into mi_hbm_layout.batches */
  int64_t param, nrows, null_count;
  int64_t dict_id;               /* MI_K_DICT: the dictionary its sel_t values index, else -1 */
  int64_t data_off;              /* arena offset of the vector data; -1 when aliased */
  int64_t valid_off;             /* arena offset of the validity words; -1 = not materialised: every row valid */
  int64_t alias_off;             /* >= 0: zero-copy, the values are the stream bytes at this position */
  uint64_t ptr_base;             /* string kinds: string_t long pointers = ptr_base + offset inside the Arrow data buffer */
  int32_t first_span, n_spans;   /* mi_hbm_layout.spans: the node's Arrow buffers as {position in the stream, length} */
  int32_t first_window, n_windows; /* mi_hbm_layout.windows: first row (this node's row space) of every 2048-row chunk + end */
} mi_hbm_node;

typedef struct mi_hbm_batch {
  int64_t nrows;
  int64_t body_off, body_len;    /* the message body inside the stream */
  int64_t arena_begin, arena_end;/* this message's part of the arena */
  int32_t first_node, n_nodes;
  int32_t n_columns;             /* its top-level columns = the nodes with parent -1, in order */
  int32_t is_dictionary;         /* 1: the decoded values of dictionary dict_id (nrows entries + one NULL slot) */
  int64_t dict_id;
} mi_hbm_batch;

typedef struct mi_hbm_layout {
  const mi_hbm_batch* batches;   /* dictionaries first, then the record batches in stream order */
  int32_t n_batches;
  int32_t n_nodes;
  const mi_hbm_node* nodes;
  const mi_buffer_span* spans;
  const int64_t* windows;
  int64_t arena_bytes;
  int64_t stream_bytes;
  int64_t n_rows;                /* rows of all record batches */
  void* device_stream;           /* the stream in HBM */
  void* device_arena;            /* NULL until an arena is bound */
  int32_t n_tasks;               /* tasks of the plan */
  int32_t _pad;
} mi_hbm_layout;

int mi_hbm_open(mi_ctx* ctx, const void* host_stream, int64_t size, const mi_hbm_options* opts, mi_hbm** out);
void mi_hbm_close(mi_hbm* h);
int mi_hbm_set_arena(mi_hbm* h, void* device_arena, int64_t bytes);
/* Pointers stay valid until mi_hbm_close. */
int mi_hbm_layout_get(mi_hbm* h, mi_hbm_layout* out);
/* One pass of the hot path over the whole stream; asynchronous on `stream` (a hipStream_t, NULL = the context's own). */
int mi_hbm_launch(mi_hbm* h, void* stream);
/* Like mi_plan_launch_timed / _status / _stats / _class_stats, for the stream's plan. */
int mi_hbm_launch_timed(mi_hbm* h, void* stream, float* ms_per_class /* [MI_NUM_KERNEL_CLASSES] */);
int mi_hbm_status(mi_hbm* h, uint32_t* status_bits);
int mi_hbm_stats(mi_hbm* h, int64_t* bytes_read, int64_t* bytes_written, int64_t* rows, int64_t* tiles);
int mi_hbm_class_stats(mi_hbm* h, int32_t kernel_class, int64_t* bytes_read, int64_t* bytes_written, int64_t* rows,
                       int64_t* tiles, const char** kernel_name);
/* D2H of [offset, offset + length) of the arena (from_stream = 0) or of the resident stream (1); synchronous. */
int mi_hbm_fetch(mi_hbm* h, int32_t from_stream, int64_t offset, int64_t length, void* host_dst);

/* K6 at kernel level (extension: the reference sets filter_pushdown=false, read_arrow.cpp:47-48): range predicate
 * lo <= v < hi on a decoded fixed-width vector (width 1, 2, 4 or 8, signed) + validity -> per-2048-row-window selection
 * vectors.  sel_out[window*2048 ...] holds ascending window-relative row indices, count_out[window] their number. */
int mi_filter_range(mi_ctx* ctx, const void* values, int32_t width, const void* validity, int64_t nrows, int64_t lo,
                    int64_t hi, mi_sel_t* sel_out, uint32_t* count_out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Scan operator.  Replaces the TableFunction bodies: read_arrow (src/scanner/read_arrow.cpp:43-86 via
 * ArrowMultiFileInfo / ArrowFileScan, src/file_scanner/) and scan_arrow_ipc (src/scanner/scan_arrow_ipc.cpp:20-64).
 * bind -> init -> repeated next() that yields one DataChunk (<= 2048 rows) per call, like
 * ArrowTableFunction::ArrowScanFunction.  Record-batch bodies are staged in pinned memory, DMA'd to HBM on a copy
 * stream, transcoded by the fused kernel, and the vectors DMA'd back into pinned chunk buffers.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct mi_scan mi_scan;

typedef struct mi_scan_options {
  int32_t union_by_name;        /* multi-file: match columns by name (README.md:104-107) */
  int32_t filename;             /* add a `filename` VARCHAR column */
  int32_t hive_partitioning;    /* add key=value path components as VARCHAR columns */
  int32_t rank;                 /* record-batch sharding: this scan takes batches with index % world == rank */
  int32_t world;                /* 0 or 1 = no sharding */
  int32_t device_resident;      /* 1: chunks stay in HBM (pointers are device addresses), no D2H */
  int32_t accept_dictionaries;  /* 1: decode DictionaryBatch + dictionary-encoded columns (beyond the reference) */
  int32_t zero_copy_direct;     /* 1: fixed-width columns that need no conversion and hold no NULLs are not copied: the
                                 * vector's data points INTO the record-batch body (host body for host consumers, which
                                 * then is not DMA'd to the GPU at all; HBM copy of the body when device_resident) and
                                 * its validity is NULL = all valid -- the reference's zero-copy DirectConversion +
                                 * unset ValidityMask.  The body stays alive until the chunk after the batch's last.
                                 * 0 = default = ON, what the reference does: a GPU consumer reads the Arrow buffer in HBM,
                                 * a host consumer the pinned host body (lineitem: 7 of 16 columns cross PCIe in neither
                                 * direction); -1 = never (every vector is materialised by a kernel). */
  int32_t unset_all_valid;      /* 1: a column without NULLs in a record batch gets no validity words: mi_vector.validity is
                                 * NULL (= all valid), like the reference's unset ValidityMask; 0: 32 all-ones words per chunk */
  int32_t filter_compact;       /* with a pushed-down filter: 1 = late materialisation, chunks hold ONLY the selected rows
                                 * (flat vectors of mi_data_chunk.size = sel_count rows, sel = NULL; rows that fail the
                                 * predicate are never decoded or copied back); 0 = full vectors + a selection vector.
                                 * Needs flat projected columns (no nested types, no string views). */
  int32_t pipeline_depth;       /* record batches in flight on the GPU (pinned + HBM slots), at most 16; 0 = the scan decides: 3, and
                                 * 8 / 16 once it meets LZ4_FRAME / ZSTD bodies that it expands in HBM (latency-bound kernels
                                 * that leave the chip idle unless many record batches run side by side) */
  int32_t host_decompress;      /* compressed bodies: 0 = auto: the bodies of a device-resident consumer are shipped over PCIe as
                                 * they are and decompressed in HBM (K8) -- LZ4_FRAME always, ZSTD when the process has
                                 * hardware queues for 16 record batches side by side (GPU_MAX_HW_QUEUES >= 12 in the
                                 * environment; the HIP runtime reads it once, at its first call, default 4: the library sets
                                 * it to 24 when it is loaded and nobody has set it) -- everything else is decompressed by the
                                 * reader's host threads; 1 = host threads always; -1 = K8 for LZ4_FRAME and ZSTD bodies, also
                                 * for host consumers (string payloads are copied back beside the vectors).  Dictionary
                                 * batches, big-endian streams, record batches with list columns and ZSTD frames with a
                                 * dictionary id or a content checksum always take the host threads. */
  int32_t _reserved[3];
} mi_scan_options;

/* read_arrow('path') / read_arrow(['a','b']) (read_arrow.cpp:78-83).  Globs are expanded by the caller. */
int mi_scan_open_files(mi_ctx* ctx, const char* const* paths, int32_t n_paths, const mi_scan_options* opts,
                       mi_scan** out);
/* scan_arrow_ipc([{ptr,size},...]) (scan_arrow_ipc.cpp:24-33). */
int mi_scan_open_buffers(mi_ctx* ctx, const mi_ipc_buffer* buffers, int32_t n_buffers, const mi_scan_options* opts,
                         mi_scan** out);
/* read_arrow over several GPUs of one process (the reference's unit of parallelism is one thread per file,
 * src/file_scanner/arrow_file_scan.cpp:35-42, arrow_multi_file_info.cpp:77-86).  One context per device (or several per
 * device: each brings its own streams and pinned ring).  Record batch k of the file list (files first, then the batches
 * inside a file) is decoded on context k mod n_ctxs; mi_scan_next returns the chunks in record-batch order whatever
 * device produced them (batch_index), mi_scan_count / mi_scan_sum_product drain every device on its own thread.
 * opts->rank / world compose: this scan takes every world-th batch first, then deals its share over the contexts.
 * Every other mi_scan_* call works on the returned handle unchanged. */
int mi_scan_open_files_multi(mi_ctx* const* ctxs, int32_t n_ctxs, const char* const* paths, int32_t n_paths,
                             const mi_scan_options* opts, mi_scan** out);
void mi_scan_close(mi_scan* s);
/* Bind result: column names + DuckDB types (names deduplicated like QueryResult::DeduplicateColumns,
 * arrow_file_scan.cpp:19). "Provided table/dataframe must have at least one column" on empty schemas. */
int mi_scan_bind(mi_scan* s, mi_field* fields, int32_t cap, int32_t* n_fields);
/* projection_pushdown = true (read_arrow.cpp:46): column names to produce, in output order. NULL/0 = all. */
int mi_scan_init(mi_scan* s, const char* const* projected_names, int32_t n_projected);

typedef struct mi_vector {
  void* data;               /* out_width bytes per row */
  mi_validity_t* validity;  /* 32 words per chunk; all ones when the column has no NULLs */
  int32_t kind;             /* enum mi_kind that produced it */
  int32_t out_width;
  const void* dictionary;   /* MI_K_DICT: decoded dictionary values (dict_len + 1 entries, last = NULL) */
  const mi_validity_t* dictionary_validity;
  int64_t dict_len;
  /* nested vectors (list / map / struct / fixed_size_list): children of this chunk's vector.  A list's child covers the
   * child rows [child_offset, child_offset + child_count) of the record batch; its data pointer is already advanced, its
   * validity pointer addresses the word that holds its first row and `validity_shift` = first row mod 64 (child windows
   * are not word aligned; 0 for top-level vectors and struct children of them). */
  const struct mi_vector* children;
  int32_t n_children;
  int32_t validity_shift;
  int64_t count;            /* rows in this vector (== chunk size for top-level vectors) */
  /* string vectors: the one allocation every long-string pointer of this vector points into (the Arrow data buffer of
   * the record batch), or NULL when unknown.  A sink may copy [heap, heap + heap_size) wholesale instead of string by
   * string; without it pointers are followed one at a time. */
  const void* heap;
  int64_t heap_size;
} mi_vector;

typedef struct mi_data_chunk {
  int64_t size;             /* rows in this chunk, 0 = scan exhausted */
  int32_t n_columns;
  int32_t file_index;       /* which input file / buffer list this chunk came from */
  int64_t batch_index;      /* global record-batch ordinal (restores order under sharding) */
  int64_t chunk_offset;     /* first row of the chunk inside its record batch */
  const mi_vector* columns; /* valid until the next mi_scan_next on this scan */
  const mi_sel_t* sel;      /* pushed-down filter: ascending chunk-relative row indices, NULL when no filter is set */
  int64_t sel_count;        /* rows selected (== size when no filter is set) */
  int64_t source_rows;      /* rows of the record batch this chunk covers (== size unless filter_compact dropped rows) */
} mi_data_chunk;

/* Pushed-down predicates (K6).  The reference sets filter_pushdown = false (read_arrow.cpp:47-48), so this is an
 * extension: DuckDB's own filter above the scan yields the same rows.  The forms are the ones DuckDB's TableFilterSet
 * hands a scan: col <op> constant with = <> < <= > >=, IS NULL, IS NOT NULL, IN (list), combined
 * by AND / OR trees over any number of columns.  Comparison columns are fixed-width integer-like after the scan
 * (integers, BOOLEAN, DATE, TIME / TIMESTAMP, DECIMAL(<=18)) and constants are the stored integers (DECIMAL(15,2) 0.05
 * is 5), or VARCHAR / BLOB columns with = <> < <= > >= IN and MI_F_STARTS_WITH against byte strings (byte-wise order,
 * a proper prefix sorts first: DuckDB's default collation); IS [NOT] NULL takes any column.  SQL semantics: a comparison with NULL is not true, so the row is dropped
 * unless another branch of an OR keeps it.  A filter column need not be projected.  The tree is normalised to at most
 * 24 leaves in conjunctive normal form; larger ones are refused with MI_ENOTSUP (DuckDB then keeps the filter above the
 * scan).  Chunks carry a selection vector (or only the selected rows: mi_scan_options.filter_compact).  Call between
 * bind and init. */
enum mi_filter_op {
  MI_F_EQ = 1, MI_F_NE = 2, MI_F_LT = 3, MI_F_LE = 4, MI_F_GT = 5, MI_F_GE = 6, MI_F_IS_NULL = 7, MI_F_IS_NOT_NULL = 8,
  MI_F_IN = 9, MI_F_STARTS_WITH = 10 /* VARCHAR / BLOB: the row begins with str_value (LIKE 'abc%', prefix()) */,
  MI_F_AND = 16, MI_F_OR = 17
};
typedef struct mi_filter_node {
  int32_t op;             /* enum mi_filter_op */
  int32_t first_child;    /* MI_F_AND / MI_F_OR: the children are nodes[first_child .. first_child + n_children) */
  int32_t n_children;
  int32_t n_values;       /* MI_F_IN */
  const char* column;     /* leaves */
  int64_t value;          /* comparison constant */
  const int64_t* values;  /* MI_F_IN */
  /* VARCHAR / BLOB columns (utf8, large_utf8, binary, fixed_size_binary, dictionary-encoded or not): MI_F_EQ / MI_F_NE /
   * MI_F_LT / MI_F_LE / MI_F_GT / MI_F_GE / MI_F_STARTS_WITH take str_value (str_len bytes, no terminator needed), MI_F_IN
   * takes str_values / str_lens; byte-wise comparison like DuckDB's.  Leave NULL for the integer forms above. */
  const char* str_value;
  int32_t str_len;
  int32_t _pad;
  const char* const* str_values;
  const int32_t* str_lens;
} mi_filter_node;
int mi_scan_set_filter(mi_scan* s, const mi_filter_node* nodes, int32_t n_nodes, int32_t root);
/* Shorthand for lo <= column < hi. */
int mi_scan_set_filter_range(mi_scan* s, const char* column, int64_t lo, int64_t hi);
int mi_scan_next(mi_scan* s, mi_data_chunk* out);
/* SELECT count(*) FROM read_arrow(...) (test/sql/read_arrow.test:35-38): pulls every remaining chunk natively.
 * rows = scanned, selected = rows passing the pushed-down filter (== rows without one), chunks = DataChunks seen. */
int mi_scan_count(mi_scan* s, int64_t* rows, int64_t* selected, int64_t* chunks);
/* Fused consumer (the reference's benchmark query is TPC-H Q6, benchmark/lineitem.py:22-34):
 *   SELECT sum(a * b), count(*) FROM scan WHERE lo_k <= f_k < hi_k  (k < n_filters <= 4)
 * evaluated on the GPU over the decoded vectors of every record batch, which never leave HBM: the only D2H traffic is
 * this 32-byte result.  Columns must be fixed-width integer-like after the scan (integers, DATE, TIME/TIMESTAMP,
 * DECIMAL(<=18)); values are the stored integers (DECIMAL(15,2) 0.05 is 5).  NULL in a filter column drops the row,
 * NULL in a or b contributes nothing (SQL SUM).  Call after bind instead of init/next: it projects the columns it
 * needs itself and drains the scan. */
typedef struct mi_range_filter {
  const char* column;
  int64_t lo, hi;            /* lo <= value < hi */
} mi_range_filter;
typedef struct mi_sum_product_result {
  uint64_t sum_lo;           /* 128-bit two's complement sum of a*b: low / high half (DuckDB sums DECIMAL products in a hugeint) */
  int64_t sum_hi;
  int64_t rows_scanned;
  int64_t rows_selected;
} mi_sum_product_result;
int mi_scan_sum_product(mi_scan* s, const char* column_a, const char* column_b, const mi_range_filter* filters,
                        int32_t n_filters, mi_sum_product_result* out);
double mi_scan_progress(mi_scan* s);
/* What the scan has moved so far (summed over the devices of a multi-device scan). */
typedef struct mi_scan_stats {
  int64_t record_batches;           /* submitted to the GPU */
  int64_t lz4_batches_on_device;    /* of those: LZ4_FRAME bodies decompressed in HBM (K8) instead of on host threads */
  int64_t h2d_bytes;                /* body bytes copied host -> HBM (compressed bytes for the K8 batches) */
  int64_t decompressed_bytes;       /* bytes the K8 kernels produced */
  int64_t lz4_blocks;               /* compressed LZ4 blocks walked by the K8 parse kernel ... */
  int64_t lz4_parse_rounds;         /* ... rounds of its 64 speculative lanes, summed over the blocks (2-3 when the guesses
                                     * fall in step, 65 = the serial walk) ... */
  int64_t lz4_parse_rounds_max;     /* ... and the worst block */
  int64_t zstd_batches_on_device;   /* ZSTD bodies decompressed in HBM (K8: entropy stage per block, then the LZ4 copy stages) */
  int64_t d2h_bytes;                /* vector (and mirrored string payload) bytes copied HBM -> pinned host memory */
  int64_t aliased_bytes;            /* vector bytes that were never copied: DirectConversion columns aliasing the body */
} mi_scan_stats;
int mi_scan_get_stats(mi_scan* s, mi_scan_stats* out);

/* ---------------------------------------------------------------------------------------------------------
 * Writer.  Replaces ColumnDataCollectionSerializer (src/writer/column_data_collection_serializer.cpp),
 * ArrowStreamWriter (src/writer/arrow_stream_writer.cpp), the COPY ... (FORMAT ARROWS) sink
 * (src/writer/write_arrow_stream.cpp:54-272) and to_arrow_ipc (src/writer/to_arrow_ipc.cpp:72-182).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct mi_writer mi_writer;

#define MI_MAX_KV_METADATA 16
typedef struct mi_write_options {
  int64_t row_group_size;        /* ROW_GROUP_SIZE / CHUNK_SIZE, default 122880 (write_arrow_stream.cpp:28-33) */
  int64_t row_group_size_bytes;  /* default row_group_size * 1024 (write_arrow_stream.cpp:36,114-118) */
  int64_t row_groups_per_file;   /* 0 = unlimited (write_arrow_stream.cpp:198-219) */
  int32_t row_group_size_set, row_group_size_bytes_set;
  int32_t preserve_insertion_order; /* DuckDB's setting of the same name; default 1 */
  int32_t n_kv_metadata;         /* kv_metadata STRUCT (write_arrow_stream.cpp:87-104) */
  char kv_keys[MI_MAX_KV_METADATA][64];
  char kv_values[MI_MAX_KV_METADATA][256]; /* BLOB values are written raw, others as their string form */
  int32_t kv_value_lens[MI_MAX_KV_METADATA];
  int32_t arrow_large_buffer_size; /* DuckDB's setting of the same name (ClientProperties.arrow_offset_size, passed to the
                                    * serializer at arrow_stream_writer.cpp:11-13): VARCHAR / BLOB / LIST export as
                                    * LargeUtf8 / LargeBinary / LargeList with int64 offsets; default 0 */
  int32_t _reserved;
} mi_write_options;

/* ArrowWriteBind (write_arrow_stream.cpp:54-125): _init fills the defaults; _set parses one COPY option (name is
 * matched case-insensitively: row_group_size | chunk_size | row_group_size_bytes | row_groups_per_file; a NULL value
 * is "<NAME> requires exactly one argument"); _add_kv appends one kv_metadata entry; _finalize applies the cross-option
 * rules.  Errors are MI_EINVAL with the reference's BinderException texts: "ROW_GROUP_SIZE and ROW_GROUP_SIZE_BYTES are
 * mutually exclusive", "ROW_GROUP_SIZE_BYTES does not work while preserving insertion order. Use \"SET
 * preserve_insertion_order=false;\" to disable preserving insertion order.". */
int mi_write_options_init(mi_write_options* o);
int mi_write_options_set(mi_write_options* o, const char* name, const char* value);
int mi_write_options_add_kv(mi_write_options* o, const char* key, const char* value, int32_t value_len);
int mi_write_options_finalize(mi_write_options* o);

/* The Schema message (encapsulated: continuation token + length + flatbuffer, padded to 8) the writer emits for
 * these columns -- ColumnDataCollectionSerializer::SerializeSchema (column_data_collection_serializer.cpp:59-65) over
 * ArrowConverter::ToArrowSchema.  Host only (no GPU needed).  *size = bytes needed; copied when cap suffices. */
int mi_encode_schema(const mi_field* fields, int32_t n_fields, uint8_t* out, int64_t cap, int64_t* size);

/* ArrowWriteInitializeGlobal (write_arrow_stream.cpp:127-139): creates the file, truncating an existing one
 * (FILE_FLAGS_FILE_CREATE_NEW = create-or-truncate in DuckDB, arrow_stream_writer.cpp:49-53) and writes the Schema message.
 * `fields`: name + duck_type per column: "BIGINT", "DECIMAL(15,2)", "VARCHAR", "DATE", "BOOLEAN", ... and the nested
 * types "T[]" (LIST), "T[N]" (ARRAY), "STRUCT(a T, b U)", "MAP(K, V)", exported the way ArrowConverter::ToArrowSchema
 * does (list child "l", map child "entries" {key not null, value}).  Nested vectors arrive as mi_vector trees. */
int mi_writer_open(mi_ctx* ctx, const char* path, const mi_field* fields, int32_t n_fields,
                   const mi_write_options* opts, mi_writer** out);
/* ArrowWriteSink (write_arrow_stream.cpp:141-159): appends one DataChunk (host vectors, DuckDB layout); flushes
 * a record batch through the encode kernels when row_group_size / row_group_size_bytes is reached. */
int mi_writer_sink(mi_writer* w, const mi_data_chunk* chunk);
/* Per-thread sink state: ArrowWriteInitializeLocal / ArrowWriteSink / ArrowWriteCombine (write_arrow_stream.cpp:141-174).
 * The reference's sink is called by every DuckDB thread with its own LocalFunctionData; here every local state also
 * serializes its own row groups (own pinned staging, own HIP stream), so appending, encoding and writing of different
 * row groups overlap.  Row groups reach the file in the order the threads finish them (DuckDB's sink without
 * preserve_insertion_order); mi_writer_sink (one thread) and mi_writer_sink_scan keep the input order.
 * _sink appends and flushes a record batch when row_group_size / row_group_size_bytes is reached, _combine flushes the
 * tail; call mi_writer_finalize once every local state is combined. */
typedef struct mi_writer_local mi_writer_local;
int mi_writer_local_create(mi_writer* w, mi_writer_local** out);
int mi_writer_local_sink(mi_writer_local* l, const mi_data_chunk* chunk);
int mi_writer_local_combine(mi_writer_local* l);
void mi_writer_local_destroy(mi_writer_local* l);
/* COPY (FROM read_arrow(...)) TO 'file' (FORMAT ARROWS): pulls every remaining record batch of `scan` (host consumer
 * mode) into the sink, natively -- the pump DuckDB's executor is between a scan and a copy sink.  Row groups are cut
 * where the one-thread sink would cut them and are staged / encoded / written by several sink threads
 * (MI_WRITER_THREADS, default cores / 3, at most 6); they reach the file in input order.  *rows = copied. */
int mi_writer_sink_scan(mi_writer* w, mi_scan* scan, int64_t* rows);
/* ArrowWriteFlushBatch (write_arrow_stream.cpp:240-245): appends one record-batch message that was serialised elsewhere --
 * the header||body blob of mi_ipc_serialize_chunks, which ArrowWritePrepareBatch (:227-238) runs concurrently, one
 * serializer per call -- at the next free position of the file and counts it as a row group.  size 0 = an empty batch. */
int mi_writer_append_message(mi_writer* w, const uint8_t* blob, int64_t size);
/* ArrowWriteCombine + ArrowWriteFinalize (write_arrow_stream.cpp:161-174): flush the tail, write EOS
 * {FF FF FF FF 00 00 00 00} (arrow_stream_writer.cpp:78-82), close. */
int mi_writer_finalize(mi_writer* w);
void mi_writer_close(mi_writer* w);
int64_t mi_writer_row_groups(const mi_writer* w);  /* ArrowStreamWriter::NumberOfRowGroups */
int64_t mi_writer_file_size(const mi_writer* w);   /* ArrowStreamWriter::FileSize */
/* ArrowWriteRotateNextFile (write_arrow_stream.cpp:204-219) */
int mi_writer_rotate_next_file(const mi_writer* w, int64_t file_size_bytes /* <0: unset */);

/* to_arrow_ipc: serialise to memory instead of a file.  mi_ipc_serialize_schema == SerializeSchema
 * (column_data_collection_serializer.cpp:57-64); mi_ipc_serialize_chunks == Serialize + header||body concat
 * (to_arrow_ipc.cpp:72-87).  Output blobs are owned by the writer until the next call. */
int mi_ipc_serializer_create(mi_ctx* ctx, const mi_field* fields, int32_t n_fields, mi_writer** out);
int mi_ipc_serialize_schema(mi_writer* w, const uint8_t** blob, int64_t* size);
int mi_ipc_serialize_chunks(mi_writer* w, const mi_data_chunk* chunks, int32_t n_chunks, const uint8_t** blob,
                            int64_t* size);

#ifdef __cplusplus
}
#endif
#endif /* MI_ARROW_IPC_H */
