// ipc_stream_reader.hpp -- host half of the scan path: Arrow IPC message framing.
//
// Takes the place of the reference's reader classes (same class names, same error strings):
//   IPCStreamReader        src/include/ipc/stream_reader/base_stream_reader.hpp:44-126, base_stream_reader.cpp
//   IPCFileStreamReader    src/ipc/stream_reader/ipc_file_stream_reader.cpp
//   IPCBufferStreamReader  src/ipc/stream_reader/ipc_buffer_stream_reader.cpp
// What differs by design: metadata is parsed by ipc_format.cpp instead of nanoarrow, GetNextBatch yields a flat
// buffer table (DecodedBatch) instead of an ArrowArray, and the file reader can read message bodies straight into
// caller-provided (pinned) memory so the body is copied exactly once on its way to HBM.
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "ipc_format.hpp"
#include "zstd_format.hpp"

namespace miarrow {

//! Runs fn(i), i in [0, n), on the process-wide I/O pool (MI_IO_THREADS, default 8) + the calling thread; rethrows the
//! first failure.  Callers on different threads share the pool.
void ParallelFor(int n, const std::function<void(int)>& fn);
int IoThreads();
//! Grows the pool to at least n threads (bounded by the host's cores); multi-device scans ask for 8 per device
void EnsureIoThreads(int n);
// NUMA locality of the host side (engine.hpp, Context::BindThisThread).  BindThisThreadToNode pins the calling thread to
// `cpus` (within what it may use) and makes its allocations prefer `node`; from then on the tasks it gives the I/O pool
// (parallel preads of a body, host decompression) run under the same binding: a pool worker adopts the binding of the batch
// of tasks it takes.  PreferNode(node) / PreferNode(-1): only the allocation policy of the calling thread.
void BindThisThreadToNode(int node, const std::vector<int>& cpus);
void PreferNode(int node);




struct ArrowIpcMessagePrefix {  // base_stream_reader.hpp:39-42
  uint32_t continuation_token;
  int32_t metadata_size;
};

//! == ArrowIPCBuffer (src/include/table_function/scan_arrow_ipc.hpp:19-23)
struct ArrowIPCBuffer {
  ArrowIPCBuffer(uint64_t ptr_p, uint64_t size_p) : ptr(ptr_p), size(size_p) {}
  uint64_t ptr;
  uint64_t size;
};

//! One field node of a record batch (depth-first), with every buffer it owns.
struct DecodedNode {
  const ArrowField* field = nullptr;
  int32_t parent = -1;
  int32_t depth = 0;
  int64_t length = 0;
  int64_t null_count = 0;
  bool value_only = false;               // dictionary batch: decode with the value type, not as indices
  std::vector<mi_buffer_span> spans;     // validity, buffer 1, buffer 2, ... (views: + variadic data buffers)
  std::vector<int32_t> children;         // indices into DecodedBatch::nodes
};

//! A record-batch body whose LZ4_FRAME / ZSTD buffers are still compressed (IPCStreamReader::SetDeferLz4): the frames were
//! walked on the host (frame header, block headers), the bytes are decompressed in HBM by the K8 kernels (kernels_lz4.hip).
struct DeferredLz4Body {
  struct Buffer {
    int64_t comp_off = 0, comp_len = 0;   // raw: the bytes themselves; else the frame, inside the compressed body
    int64_t out_off = 0, out_len = 0;     // place in the decompressed body
    bool raw = false;                     // stored uncompressed (length prefix -1)
    uint32_t first_block = 0, n_blocks = 0, block_max = 0;
  };
  struct Block {
    uint32_t comp_off = 0, comp_size = 0, buffer = 0, stored = 0;
    uint32_t seq_cap = 0;                 // ZSTD: sequence descriptors the block needs (LZ4: derived from comp_size)
  };
  const uint8_t* comp = nullptr;          // the compressed body as it was read (kept alive by DecodedBatch::owner)
  int64_t comp_size = 0;
  std::vector<Buffer> buffers;            // the needed, non-empty buffers of the message
  std::vector<Block> blocks;              // every block of every non-raw buffer, buffer by buffer
  int32_t codec = 0;                      // 0 LZ4_FRAME, 1 ZSTD
  std::vector<zstd::BlockInfo> zblocks;   // ZSTD: one per entry of `blocks`
  uint32_t literal_scratch = 0;           // ZSTD: bytes of decoded literals (BlockInfo::lit_pos of non-raw literals counts from 0)
};
//! Walks one ZSTD frame (one IPC buffer) from its headers.  false = a frame the device path does not take (dictionary,
//! content checksum, several frames, anything malformed): the host decompressor handles it and reports what is wrong.
bool WalkZstdFrame(const uint8_t* body, int64_t frame_off, int64_t frame_len, uint32_t buffer_index, int64_t declared_len,
                   DeferredLz4Body::Buffer* buf, std::vector<DeferredLz4Body::Block>* blocks, std::vector<zstd::BlockInfo>* infos,
                   uint32_t* literal_scratch);

//! What GetNextBatch produces: the buffers of every (projected) top-level column of one message.
struct DecodedBatch {
  int64_t length = 0;
  const uint8_t* body = nullptr;
  int64_t body_size = 0;
  int64_t body_file_offset = 0;
  bool is_dictionary = false;
  int64_t dict_id = -1;
  bool is_delta = false;
  int32_t compression = -1;
  std::vector<int32_t> column_field;     // top-level field index per output column
  std::vector<int64_t> null_count;       // per output column
  std::vector<int64_t> column_length;    // per output column (== length for top-level fields)
  std::vector<mi_buffer_span> buffers;   // 3 per output column: validity, buf1, buf2
  std::vector<DecodedNode> nodes;        // the projected columns with their descendants, depth-first
  std::vector<int32_t> column_node;      // per output column: its node
  //! Keeps the body alive (file reader: shared ownership like shared_ptr<AllocatedData>, base_stream_reader.cpp:286-294)
  std::shared_ptr<void> owner;
  //! set: `body` is NULL, body_size and every span describe the DECOMPRESSED layout, the bytes are still compressed
  std::shared_ptr<const DeferredLz4Body> deferred;
};

struct BatchIndexEntry {
  int64_t prefix_offset;
  int32_t meta_len;
  int32_t type;
  int64_t body_offset;
  int64_t body_len;
  int64_t n_rows;
};

//! Base IPC Reader
class IPCStreamReader {
 public:
  virtual ~IPCStreamReader() = default;

  //! Gets the output schema, which is the file schema with projection pushdown being considered
  const ArrowSchemaModel& GetOutputSchema();
  //! Gets the base schema with no projection pushdown
  const ArrowSchemaModel& GetBaseSchema();
  //! Gets the next batch; false at end of stream.  accept_dictionaries: also return DictionaryBatch messages
  //! (the reference accepts RecordBatch only, base_stream_reader.cpp:86-96)
  bool GetNextBatch(DecodedBatch* out, bool accept_dictionaries = false, bool skip_record_batch_body = false);
  //! Sets the projection pushdown for this reader
  void SetColumnProjection(const std::vector<std::string>& column_names);
  bool HasProjection() const { return !projected_fields.empty(); }
  //! Drops the reader's own reference to the body of the message it returned last (the DecodedBatch keeps its own);
  //! a caller that recycles body buffers needs this when it stops pulling from a reader
  void ReleaseCurrentBody() {
    cur_owner.reset();
    compressed_owner.reset();
    cur_ptr = nullptr;
    cur_size = 0;
  }
  //! Byte ranges of a record-batch body that hold the buffers of the projected columns (merged when closer than
  //! `gap`); empty = everything (no projection, compressed body, or malformed metadata: the full validation decides).
  std::vector<char> NeededBuffers(const RecordBatchMeta& meta) const;
  std::vector<std::pair<int64_t, int64_t>> ProjectedBodyRanges(const RecordBatchMeta& meta, int64_t body_length, int64_t gap) const;
  const std::vector<int64_t>& ProjectedFlatFields() const { return projected_fields; }

  MessageType ReadNextMessage(std::vector<MessageType> expected_types, bool end_of_stream_ok = true);
  virtual MessageType ReadNextMessage() = 0;
  virtual double GetProgress() { return 0; }
  //! Header-only walk of the remaining input: batch boundaries for record-batch sharding
  virtual const std::vector<BatchIndexEntry>& BuildIndex() = 0;

  //! Where message bodies are placed (file reader only). Default: an internal 64-byte aligned heap block per message.
  using BodyAllocator = std::function<std::shared_ptr<void>(size_t bytes, MessageType type, uint8_t** ptr)>;
  void SetBodyAllocator(BodyAllocator a) { body_allocator = std::move(a); }

  //! LZ4_FRAME record batches (not dictionary batches, not big-endian streams) are handed out still compressed, with the
  //! frame / block tables a GPU decompressor needs (DecodedBatch::deferred); everything else is decompressed here as before
  void SetDeferLz4(bool on) { defer_lz4 = on; }
  //! the same for ZSTD record batches (frames with a dictionary id or a content checksum stay with the host library)
  void SetDeferZstd(bool on) { defer_zstd = on; }

  static int64_t CountFields(const ArrowField& field) { return field.CountFields(); }
  static constexpr uint32_t kContinuationToken = 0xFFFFFFFF;

 protected:
  //! With the prefix in message_prefix: checks the metadata size, then header and body through the two virtual seams
  //! below (what the reference does in DecodeMetadata + DecodeMessage, base_stream_reader.cpp:214-236)
  MessageType FinishMessage();
  //! Reads and parses the flatbuffer header (message_header_size = prefix + metadata); true = end-of-stream marker
  virtual bool DecodeHeader(idx_t message_header_size) = 0;
  //! Makes the message body available at cur_ptr / cur_size
  virtual void DecodeBody() = 0;

  //! Parses the current header into `message` (ENODATA == metadata_size 0 => returns false)
  bool ParseHeader(const uint8_t* header_with_prefix, idx_t size);
  //! Slices cur_ptr/cur_size into per-column buffers, with the size checks of NANOARROW_VALIDATION_LEVEL_FULL that do
  //! not need the data (offset monotonicity is checked on the device by the string kernel)
  void SliceBatch(const RecordBatchMeta& meta, DecodedBatch* out);
  //! Replaces cur_ptr/cur_size with the decompressed body and rewrites meta->buffers (ZSTD, per buffer; the CPU step the
  //! reference performs in DuckDBDecompressZstd, base_stream_reader.cpp:11-32)
  void DecompressBody(RecordBatchMeta* meta);
  bool defer_lz4 = false, defer_zstd = false;
  std::shared_ptr<const DeferredLz4Body> cur_deferred;   // set by DecompressBody when the current body stays compressed
  //! Big-endian stream: every multi-byte number of the body is swapped in place (after decompression), so the rest of the
  //! path sees little-endian buffers (what nanoarrow's decoder does for the reference, base_stream_reader.cpp:68-69)
  void SwapBodyEndianness(const RecordBatchMeta& meta);
  std::shared_ptr<void> compressed_owner;

  MessageHeader message;               // the decoder's message_type / body_size_bytes
  const uint8_t* message_meta = nullptr;  // flatbuffer of the current message
  int64_t message_meta_len = 0;

  std::vector<int64_t> projected_fields;   // flattened field index per projected column
  std::vector<int32_t> projected_columns;  // top-level field index per projected column
  ArrowSchemaModel projected_schema;
  ArrowSchemaModel base_schema;
  bool have_base_schema = false;

  //! Information on current buffer
  const uint8_t* cur_ptr = nullptr;
  int64_t cur_size = 0;
  int64_t cur_body_offset = 0;
  std::shared_ptr<void> cur_owner;

  bool finished = false;
  bool skip_record_batch_body = false;  // sharded scans: batches owned by another rank are stepped over unread
  ArrowIpcMessagePrefix message_prefix{};
  BodyAllocator body_allocator;
  std::vector<BatchIndexEntry> index;
  bool index_built = false;
};

//! Reads from a file (stream format, or the stream embedded in the file format)
class IPCFileStreamReader : public IPCStreamReader {
 public:
  explicit IPCFileStreamReader(const std::string& path);
  ~IPCFileStreamReader() override;

  MessageType ReadNextMessage() override;
  double GetProgress() override;
  void PopulateNames(std::vector<std::string>& names);
  const std::vector<BatchIndexEntry>& BuildIndex() override;
  //! Positions the reader on a message found by BuildIndex (record-batch sharding)
  void Seek(int64_t prefix_offset);
  int64_t FileSize() const { return file_size; }
  //! true when BuildIndex() came from the IPC file footer instead of a header walk
  bool IndexFromFooterUsed() const { return index_from_footer; }

 protected:
  const uint8_t* ReadData(uint8_t* ptr, idx_t size);
  bool DecodeHeader(idx_t message_header_size) override;
  void DecodeBody() override;
  bool ReadPrefix();
  void SkipBodyPadding();
  bool IndexFromFooter();

 private:
  bool index_from_footer = false;
  int fd = -1;
  std::string path;
  int64_t file_size = 0;
  int64_t offset = 0;  // BufferedFileReader::CurrentOffset
  std::vector<uint8_t> message_header;
};

//! Reads from caller-owned memory, zero copy
class IPCBufferStreamReader : public IPCStreamReader {
 public:
  explicit IPCBufferStreamReader(std::vector<ArrowIPCBuffer> buffers);

  MessageType ReadNextMessage() override;
  const std::vector<BatchIndexEntry>& BuildIndex() override;
  double GetProgress() override;

 protected:
  const uint8_t* ReadData(idx_t size);
  bool DecodeHeader(idx_t message_header_size) override;
  void DecodeBody() override;

 private:
  //! positions `view` on the next byte of the buffer list nobody has read yet; false when there is none
  bool SeekUnreadByte();
  struct View {
    const uint8_t* ptr = nullptr;
    bool opened = false;            // false until the first buffer is opened
    int64_t size = 0;
    int64_t pos = 0;
  };
  std::vector<ArrowIPCBuffer> buffers;
  View view;
  idx_t view_index = 0;             // which buffer `view` is a window of
  const uint8_t* prefix_at = nullptr;  // where the current message's prefix lies in the caller's memory
};

}  // namespace miarrow
