// ipc_format.hpp -- Arrow IPC metadata model: Schema / RecordBatch / DictionaryBatch / Footer decode and
// Schema / RecordBatch encode.  This is the part of the path the reference delegates to nanoarrow_ipc
// (ArrowIpcDecoderDecodeHeader / DecodeSchema / DecodeArrayFromShared at
// src/ipc/stream_reader/base_stream_reader.cpp:52-144 and ArrowIpcEncoderEncodeSchema / EncodeSimpleRecordBatch at
// src/writer/column_data_collection_serializer.cpp:57-92), plus the Arrow -> DuckDB type mapping that DuckDB's
// ArrowTableFunction::PopulateArrowTableType performs for src/file_scanner/arrow_file_scan.cpp:17-18.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mi_arrow_ipc.h"

namespace miarrow {

using idx_t = uint64_t;

//! [offset, offset + length) lies inside [0, size).  Written without the addition: offsets and lengths come from files,
//! and `offset + length > size` wraps for values near INT64_MAX and then passes.
inline bool SpanInside(int64_t offset, int64_t length, int64_t size) {
  return offset >= 0 && length >= 0 && offset <= size && length <= size - offset;
}

// Exception types mirror the DuckDB exception classes the reference throws; the C ABI maps them to errno codes
// the way IpcArrayStream::Wrap does (src/include/ipc/array_stream.hpp:29-48).
struct Exception : std::runtime_error {
  int code;
  Exception(int code_p, const std::string& msg) : std::runtime_error(msg), code(code_p) {}
};
struct IOException : Exception {
  explicit IOException(const std::string& m) : Exception(MI_EIO, m) {}
};
struct InternalException : Exception {
  explicit InternalException(const std::string& m) : Exception(MI_EINVAL, m) {}
};
struct InvalidInputException : Exception {
  explicit InvalidInputException(const std::string& m) : Exception(MI_EINVAL, m) {}
};
struct BinderException : Exception {
  explicit BinderException(const std::string& m) : Exception(MI_EINVAL, m) {}
};
struct NotImplementedException : Exception {
  explicit NotImplementedException(const std::string& m) : Exception(MI_ENOTSUP, m) {}
};
struct ConversionException : Exception {
  explicit ConversionException(const std::string& m) : Exception(MI_ERANGE, m) {}
};

enum class MessageType : int32_t { UNINITIALIZED = 0, SCHEMA = 1, DICTIONARY_BATCH = 2, RECORD_BATCH = 3, TENSOR = 4, SPARSE_TENSOR = 5 };
const char* MessageTypeString(MessageType t);  // base_stream_reader.cpp:296-313

struct ArrowField {
  std::string name;
  std::string timezone;
  int32_t type = MI_AT_NONE;
  int32_t bit_width = 0;
  bool is_signed = false;
  int32_t precision = 0;
  int32_t scale = 0;
  int32_t unit = 0;
  int32_t byte_width = 0;
  bool nullable = true;
  bool has_dictionary = false;
  int64_t dict_id = 0;
  int32_t dict_index_bit_width = 32;
  bool dict_index_signed = true;
  bool dict_ordered = false;
  std::vector<ArrowField> children;
  std::vector<std::pair<std::string, std::string>> metadata;

  // derived
  std::string Format() const;    // Arrow C data interface format string
  std::string DuckType() const;  // DuckDB logical type name
  // Transcode plan for the value type (ignoring dictionary encoding when value_only)
  bool Plan(int32_t* kind, int64_t* param, int32_t* out_width, int32_t* n_buffers, bool value_only = false) const;
  bool Supported(std::string* why) const;  // this field and all its descendants can be decoded by the path
  int64_t CountFields() const;   // IPCStreamReader::CountFields (base_stream_reader.cpp:271-277)
  int64_t CountBuffers() const;  // buffers this field and its children own in a RecordBatch body
};

struct ArrowSchemaModel {
  std::vector<ArrowField> fields;
  std::vector<std::pair<std::string, std::string>> metadata;
  int32_t endianness = 0;  // 0 little, 1 big
  uint32_t features = 0;
};

struct MessageHeader {
  MessageType type = MessageType::UNINITIALIZED;
  int32_t version = 0;
  int64_t body_length = 0;
};

struct RecordBatchMeta {
  int64_t length = 0;
  std::vector<std::pair<int64_t, int64_t>> nodes;    // {length, null_count}
  std::vector<mi_buffer_span> buffers;               // {offset, length}
  int32_t compression = -1;                          // -1 none, 0 LZ4_FRAME, 1 ZSTD
  std::vector<int64_t> variadic_counts;
  bool is_dictionary = false;
  int64_t dict_id = -1;
  bool is_delta = false;
};

struct FooterBlock {
  int64_t offset;
  int32_t meta_len;
  int64_t body_len;
};

// ---- decode (throw IOException on malformed metadata, like THROW_NOT_OK(IOException, ...)) ----
MessageHeader DecodeMessageHeader(const uint8_t* meta, int64_t meta_len);
ArrowSchemaModel DecodeSchema(const uint8_t* meta, int64_t meta_len);
RecordBatchMeta DecodeRecordBatch(const uint8_t* meta, int64_t meta_len);
bool DecodeFooter(const uint8_t* file_tail, int64_t tail_len, int64_t file_size, std::vector<FooterBlock>* dict_blocks,
                  std::vector<FooterBlock>* batch_blocks);

// ---- encode: complete encapsulated messages (8-byte prefix + flatbuffer padded so that the total is a multiple
// of 8, like ArrowIpcEncoderFinalizeBuffer(encoder, /*encapsulate*/ true, ...)) ----
std::vector<uint8_t> EncodeSchemaMessage(const ArrowSchemaModel& schema);
std::vector<uint8_t> EncodeRecordBatchMessage(int64_t length, const std::vector<std::pair<int64_t, int64_t>>& nodes,
                                              const std::vector<mi_buffer_span>& buffers, int64_t body_length);

// DuckDB logical type name ("BIGINT", "DECIMAL(15,2)", ...) -> Arrow field as ArrowConverter::ToArrowSchema
// exports it (arrow_stream_writer.cpp:22-24). Throws NotImplementedException for types outside the path.
ArrowField FieldFromDuckType(const std::string& name, const std::string& duck_type);

void FillCField(const ArrowField& f, int32_t flat_index, mi_field* out);

}  // namespace miarrow
