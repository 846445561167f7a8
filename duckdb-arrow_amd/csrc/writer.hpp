// writer.hpp -- the encode direction: DuckDB vectors -> Arrow IPC messages.
//
// Mirrors the reference's writer classes (same names and call structure):
//   ColumnDataCollectionSerializer   src/writer/column_data_collection_serializer.cpp  (Init / SerializeSchema /
//                                    Serialize / Flush / GetHeader / GetBody)
//   ArrowStreamWriter                src/writer/arrow_stream_writer.cpp (InitSchema / InitOutputFile / WriteSchema /
//                                    Flush / Finalize / NumberOfRowGroups / FileSize)
//   COPY ... (FORMAT ARROWS) sink    src/writer/write_arrow_stream.cpp:141-174 (ArrowWriteSink / Combine / Finalize)
// ArrowConverter::ToArrowArray / ArrowAppender (DuckDB core; call site column_data_collection_serializer.cpp:85) is the
// part that runs on the GPU: the buffered chunks are staged in pinned memory, DMA'd to HBM, encoded by the K7 kernels
// straight into the IPC body layout (every buffer padded to 8 bytes, column_data_collection_serializer.cpp:86-92 ->
// ArrowIpcEncoderEncodeSimpleRecordBatch) and DMA'd back as one body.
#pragma once

#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "engine.hpp"
#include "ipc_format.hpp"

namespace miarrow {

//! The buffered rows of one future record batch (the reference's local ColumnDataCollection,
//! write_arrow_stream.cpp:43-52): per FIELD NODE (depth first; a flat column is one node) a pinned staging array in
//! DuckDB layout.  Nested vectors are flattened while they are appended, the way ArrowAppender walks them: struct and
//! fixed-size-list children take every parent row, a list's child rows are gathered in list order (NULL lists add none)
//! and the list node itself stages its list_entry_t rows (the GPU turns their lengths into the int32 Arrow offsets).
class ChunkCollection {
 public:
  //! fields as exported (LargeUtf8 / LargeBinary / LargeList nodes get int64 offsets)
  ChunkCollection(Context* ctx, const std::vector<ArrowField>& fields);
  ~ChunkCollection();
  void Append(const mi_data_chunk& chunk);
  void Reset();
  int64_t Count() const { return count; }
  int64_t SizeInBytes() const { return size_in_bytes; }

  struct Column {
    int32_t enc_kind = 0;    // MI_K_ENC_* of a leaf; MI_K_ENC_LIST32 for a list / map; 0 for struct / fixed-size list
    int32_t arrow_type = 0;  // MI_AT_*
    int64_t param = 0;       // vector element width (or decimal physical width); fixed_size_list: list size
    int32_t width = 0;       // bytes per staged row (list: 16 = one list_entry_t)
    int32_t depth = 0;
    int64_t count = 0;       // rows buffered in this node
    uint8_t* data = nullptr;      size_t data_cap = 0;       // pinned
    uint64_t* validity = nullptr; size_t validity_cap = 0;   // pinned words
    uint8_t* heap = nullptr;      size_t heap_cap = 0;       // pinned payload of long strings (pointer = heap offset)
    int64_t heap_used = 0;
    int64_t payload_bytes = 0;    // sum of valid string lengths = size of the Arrow data buffer (list: child rows gathered)
    bool has_nulls = false;
    bool large_offsets = false;   // int64 Arrow offsets (arrow_large_buffer_size)
    // long-string payloads are staged in `heap` and the staged string_t rows point at heap offsets (ptr_base 0).  When the
    // source vector declares the allocation its long strings live in (mi_vector.heap: the Arrow data buffer of a scanned
    // record batch) and they ascend inside it, the bytes from the first to the last long string of an appended slice are
    // staged with ONE copy; otherwise string by string.  Nothing outside a declared allocation or a string is ever read.
    uint64_t ptr_base = 0;        // pointer value of heap[0] as the staged string_t rows see it (always 0)
    std::vector<int32_t> children;
    bool IsList() const { return arrow_type == MI_AT_LIST || arrow_type == MI_AT_LARGE_LIST || arrow_type == MI_AT_MAP; }
    bool IsGroup() const { return arrow_type == MI_AT_STRUCT || arrow_type == MI_AT_FIXED_LIST; }
  };
  std::vector<Column> columns;     // every field node, depth first
  std::vector<int32_t> roots;      // node of top-level column c

 private:
  int32_t AddField(const ArrowField& f, int32_t depth);
  void AppendNode(int32_t ni, const mi_vector& v, int64_t start, int64_t n);
  void Reserve(Column& c, int64_t rows, int64_t extra_heap);
  Context* ctx;
  int64_t count = 0;
  int64_t size_in_bytes = 0;
};

class ColumnDataCollectionSerializer {
 public:
  //! own_stream: the encode runs on a stream of its own (one serializer per sink thread: their H2D / K7 / D2H overlap)
  explicit ColumnDataCollectionSerializer(Context* ctx, bool own_stream = false);
  ~ColumnDataCollectionSerializer();
  void Init(const ArrowSchemaModel* schema);
  void SerializeSchema();
  //! Serializes the collection as ONE record batch (header + body). Returns the number of messages (0 when empty).
  idx_t Serialize(ChunkCollection& buffer);
  const std::vector<uint8_t>& GetHeader() const { return header; }
  const uint8_t* GetBody() const { return h_body; }
  int CurrentBody() const { return cur_body; }
  //! The next Serialize() writes its body into the other pinned buffer (the stream writer hands the current one to its
  //! I/O thread); returns the index of the buffer that was current
  int SwapBody() {
    const int was = cur_body;
    h_bodies[cur_body] = h_body;
    h_body_caps[cur_body] = h_body_cap;
    cur_body ^= 1;
    h_body = h_bodies[cur_body];
    h_body_cap = h_body_caps[cur_body];
    return was;
  }
  int64_t GetBodySize() const { return body_size; }
  int64_t LastBytesRead() const { return plan ? plan->bytes_read : 0; }

 private:
  Context* ctx;
  const ArrowSchemaModel* schema = nullptr;
  std::vector<uint8_t> header;
  uint8_t* h_body = nullptr;  size_t h_body_cap = 0;   // pinned; the current one of two
  uint8_t* h_bodies[2] = {nullptr, nullptr};
  size_t h_body_caps[2] = {0, 0};
  int cur_body = 0;
  uint8_t* d_body = nullptr;  size_t d_body_cap = 0;
  uint8_t* d_in = nullptr;    size_t d_in_cap = 0;
  int64_t body_size = 0;
  std::unique_ptr<Plan> plan;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
};

class ArrowStreamWriter {
 public:
  ArrowStreamWriter(Context* ctx, const std::string& file_path, const std::vector<ArrowField>& fields,
                    const std::vector<std::pair<std::string, std::string>>& metadata);
  ~ArrowStreamWriter();
  void WriteSchema();
  void Flush(ChunkCollection& buffer);
  void Finalize();
  // ---- thread-safe half, for sink threads that serialize their own row groups (write_arrow_stream.cpp:141-159: every
  // DuckDB thread has a local state; here each also owns a serializer) ----
  //! Claims the next `bytes` of the file for one record-batch message and counts the row group
  int64_t ReserveRowGroup(size_t bytes);
  //! pwrite of a claimed range (any thread, no lock)
  void WriteAt(int64_t offset, const uint8_t* p, size_t n);
  void CountEmptyFlush() { std::lock_guard<std::mutex> lk(io_mu); ++row_group_count; }
  Context* GetContext() const { return ctx; }
  idx_t NumberOfRowGroups() const { return row_group_count; }
  idx_t FileSize() const { return total_written; }
  const ArrowSchemaModel& Schema() const { return schema; }

 private:
  void InitSchema(const std::vector<ArrowField>& fields, const std::vector<std::pair<std::string, std::string>>& metadata);
  void InitOutputFile(const std::string& file_path);
  void WriteData(const uint8_t* p, size_t n);   // appends at the end of what has been claimed so far

  // record-batch messages are written by an I/O thread while the sink stages the next row group; two body buffers
  // alternate (the reference writes synchronously inside Flush, arrow_stream_writer.cpp:66-77)
  struct WriteJob {
    std::vector<uint8_t> header;
    const uint8_t* body = nullptr;
    size_t body_size = 0;
    int buffer = -1;
    int64_t offset = 0;   // claimed position of the message in the file
  };
  void IoLoop();
  void WaitBufferFree(int buffer);
  void DrainIo();
  std::thread io_thread;
  std::mutex io_mu;
  std::condition_variable io_cv;
  std::deque<WriteJob> io_jobs;
  bool buffer_busy[2] = {false, false};
  bool io_stop = false;
  std::exception_ptr io_error;

  Context* ctx;
  ArrowSchemaModel schema;
  ColumnDataCollectionSerializer serializer;
  std::string file_name;
  int fd = -1;
  idx_t row_group_count = 0;
  idx_t total_written = 0;      // bytes handed to the file (queued writes included: rotation decisions see them at once)
  bool finalized = false;
};

}  // namespace miarrow
