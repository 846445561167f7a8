// writer.cpp -- see writer.hpp.
#include "writer.hpp"
#include "ipc_stream_reader.hpp"
#include "scan_operator.hpp"

#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cctype>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <functional>

namespace miarrow {

int WrapC(const std::function<void()>& f);  // c_api.cpp

namespace {
// MI_WRITER_TIMING=1: cumulative seconds per stage of the COPY sink, printed when a writer is finalized
struct SinkTimers {
  double append = 0, serialize = 0, write = 0;
  bool on = std::getenv("MI_WRITER_TIMING") != nullptr;
  std::mutex mu;  // several sink threads add their stage times
};
SinkTimers& Timers() {
  static SinkTimers t;
  return t;
}
struct ScopedTimer {
  double* acc;
  std::chrono::steady_clock::time_point t0;
  explicit ScopedTimer(double* a) : acc(Timers().on ? a : nullptr) { if (acc) t0 = std::chrono::steady_clock::now(); }
  ~ScopedTimer() {
    if (!acc) return;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::lock_guard<std::mutex> lk(Timers().mu);
    *acc += dt;
  }
};
constexpr size_t kBufferAlign = 64;  // Arrow's recommended buffer alignment; any multiple of 8 is valid IPC
size_t RoundUp(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <typename T>
void GrowPinned(T** p, size_t* cap_bytes, size_t need_bytes, size_t keep_bytes) {
  if (need_bytes <= *cap_bytes && *p) return;
  // a quarter of headroom: row groups of one table differ by a few percent, and growing means hipHostFree / hipFree, which
  // wait for the device to go idle -- with the other sink threads' row groups in flight a stall of milliseconds
  size_t ncap = RoundUp(std::max(need_bytes + need_bytes / 4, *cap_bytes + *cap_bytes / 2), 1 << 16);
  void* np = nullptr;
  MI_HIP_CHECK(hipHostMalloc(&np, ncap, hipHostMallocDefault));
  if (*p) {
    std::memcpy(np, *p, std::min(keep_bytes, *cap_bytes));
    MI_HIP_CHECK(hipHostFree(*p));
  }
  *p = static_cast<T*>(np);
  *cap_bytes = ncap;
}

void GrowDevice(uint8_t** p, size_t* cap, size_t need) {
  if (need <= *cap && *p) return;
  if (*p) MI_HIP_CHECK(hipFree(*p));
  *p = nullptr;
  *cap = RoundUp(std::max(need + need / 4, *cap + *cap / 2), 1 << 16);
  MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), *cap));
}

// DuckDB logical type -> how the vector is laid out and which K7 kernel encodes it
void EncodePlanFor(const ArrowField& f, int32_t* enc_kind, int64_t* param, int32_t* width) {
  switch (f.type) {
    case MI_AT_BOOL: *enc_kind = MI_K_ENC_BOOL; *param = 1; *width = 1; return;
    case MI_AT_INT: *enc_kind = MI_K_ENC_COPY; *param = f.bit_width / 8; *width = f.bit_width / 8; return;
    case MI_AT_FLOAT: *enc_kind = MI_K_ENC_COPY; *param = f.precision == 1 ? 4 : 8; *width = static_cast<int32_t>(*param); return;
    case MI_AT_DATE: *enc_kind = MI_K_ENC_COPY; *param = 4; *width = 4; return;
    case MI_AT_TIME: case MI_AT_TIMESTAMP: *enc_kind = MI_K_ENC_COPY; *param = 8; *width = 8; return;
    case MI_AT_DECIMAL:
      if (f.precision <= 4) { *enc_kind = MI_K_ENC_DEC128; *param = 2; *width = 2; }
      else if (f.precision <= 9) { *enc_kind = MI_K_ENC_DEC128; *param = 4; *width = 4; }
      else if (f.precision <= 18) { *enc_kind = MI_K_ENC_DEC128; *param = 8; *width = 8; }
      else { *enc_kind = MI_K_ENC_COPY; *param = 16; *width = 16; }
      return;
    case MI_AT_UTF8: case MI_AT_BINARY: case MI_AT_LARGE_UTF8: case MI_AT_LARGE_BINARY:
      *enc_kind = MI_K_ENC_STR32; *param = 0; *width = 16; return;
    default: throw NotImplementedException("Arrow type " + f.Format() + " is not encoded by the MI355X writer path");
  }
}

// append `n` bits of `src` starting at bit `spos` (src NULL = all ones) to `dst` at bit position `pos`
void AppendBits(uint64_t* dst, int64_t pos, const uint64_t* src, int64_t spos, int64_t n) {
  for (int64_t i = 0; i < n;) {
    const int64_t dw = (pos + i) >> 6;
    const int dsh = static_cast<int>((pos + i) & 63);
    const int64_t take = std::min<int64_t>(64 - dsh, n - i);
    uint64_t bits;
    if (!src) {
      bits = ~0ull;
    } else {
      const int64_t sw = (spos + i) >> 6;
      const int ssh = static_cast<int>((spos + i) & 63);
      bits = src[sw] >> ssh;
      if (ssh && ssh + take > 64) bits |= src[sw + 1] << (64 - ssh);
    }
    const uint64_t mask = take == 64 ? ~0ull : ((1ull << take) - 1ull);
    dst[dw] = (dst[dw] & ~(mask << dsh)) | ((bits & mask) << dsh);
    i += take;
  }
}
inline bool BitAt(const uint64_t* v, int64_t i) { return !v || ((v[i >> 6] >> (i & 63)) & 1); }
}  // namespace

// ------------------------------------------------------------------------------------------------ collection
ChunkCollection::ChunkCollection(Context* ctx_p, const std::vector<ArrowField>& fields) : ctx(ctx_p) {
  for (auto& f : fields) roots.push_back(AddField(f, 0));
}

int32_t ChunkCollection::AddField(const ArrowField& f, int32_t depth) {
  const int32_t idx = static_cast<int32_t>(columns.size());
  columns.emplace_back();
  {
    Column& c = columns.back();
    c.arrow_type = f.type;
    c.depth = depth;
    switch (f.type) {
      case MI_AT_STRUCT: break;
      case MI_AT_FIXED_LIST: c.param = f.byte_width; break;
      case MI_AT_LIST: case MI_AT_LARGE_LIST: case MI_AT_MAP: c.enc_kind = MI_K_ENC_LIST32; c.param = 0; c.width = 16; break;
      default: EncodePlanFor(f, &c.enc_kind, &c.param, &c.width); break;
    }
    c.large_offsets = f.type == MI_AT_LARGE_UTF8 || f.type == MI_AT_LARGE_BINARY || f.type == MI_AT_LARGE_LIST;
  }
  for (auto& ch : f.children) {
    const int32_t k = AddField(ch, depth + 1);
    columns[static_cast<size_t>(idx)].children.push_back(k);
  }
  return idx;
}

ChunkCollection::~ChunkCollection() {
  for (auto& c : columns) {
    if (c.data) (void)hipHostFree(c.data);
    if (c.validity) (void)hipHostFree(c.validity);
    if (c.heap) (void)hipHostFree(c.heap);
  }
}

void ChunkCollection::Reserve(Column& c, int64_t rows, int64_t extra_heap) {
  ctx->Bind();
  if (c.width > 0)
    GrowPinned(&c.data, &c.data_cap, static_cast<size_t>(rows) * static_cast<size_t>(c.width) + 64,
               static_cast<size_t>(c.count) * static_cast<size_t>(c.width));
  const size_t old_vcap = c.validity_cap;
  GrowPinned(&c.validity, &c.validity_cap, static_cast<size_t>((rows + 63) / 64) * 8 + 16, static_cast<size_t>((c.count + 63) / 64) * 8);
  if (c.validity_cap != old_vcap) {
    const size_t used = static_cast<size_t>((c.count + 63) / 64) * 8;
    std::memset(reinterpret_cast<uint8_t*>(c.validity) + used, 0xFF, c.validity_cap - used);
  }
  if (extra_heap > 0)
    GrowPinned(&c.heap, &c.heap_cap, static_cast<size_t>(c.heap_used + extra_heap) + 64, static_cast<size_t>(c.heap_used));
}

void ChunkCollection::Append(const mi_data_chunk& chunk) {
  ScopedTimer timer(&Timers().append);
  if (chunk.n_columns != static_cast<int32_t>(roots.size()))
    throw InvalidInputException("DataChunk has " + std::to_string(chunk.n_columns) + " columns, the writer expects " + std::to_string(roots.size()));
  const int64_t n = chunk.size;
  if (n <= 0) return;
  if (n > MI_VECTOR_SIZE * 1024) throw InvalidInputException("DataChunk too large");
  for (size_t ci = 0; ci < roots.size(); ci++) AppendNode(roots[ci], chunk.columns[ci], 0, n);
  count += n;
}

// rows [start, start + n) of vector `v` -> node `ni` (and, for nested types, the rows they own in the child nodes)
void ChunkCollection::AppendNode(int32_t ni, const mi_vector& v, int64_t start, int64_t n) {
  if (n <= 0) return;
  const int64_t vbit = static_cast<int64_t>(v.validity_shift) + start;  // bit of v.validity that belongs to row `start`
  {
    Column& c = columns[static_cast<size_t>(ni)];
    if (!c.IsGroup() && !v.data) throw InvalidInputException("DataChunk column " + std::to_string(ni) + " has no data");
    if ((c.IsList() || c.IsGroup()) && (v.n_children < (c.arrow_type == MI_AT_STRUCT ? static_cast<int32_t>(c.children.size()) : 1) || !v.children))
      throw InvalidInputException("nested vector without child vectors");
  }
  // strings: one pass over the rows decides how the long-string payloads of this slice are staged
  int64_t extra_heap = 0, payload = 0;
  bool one_copy = false;
  uint64_t region_lo = 0, region_hi = 0;
  if (columns[static_cast<size_t>(ni)].enc_kind == MI_K_ENC_STR32) {
    const mi_string_t* s = static_cast<const mi_string_t*>(v.data) + start;
    const uint64_t heap_lo = reinterpret_cast<uint64_t>(v.heap), heap_hi = heap_lo + static_cast<uint64_t>(v.heap_size > 0 ? v.heap_size : 0);
    bool ascending_inside = v.heap != nullptr;
    int64_t long_bytes = 0;
    uint64_t prev_end = 0;
    for (int64_t i = 0; i < n; i++) {
      if (!BitAt(v.validity, vbit + i)) continue;
      const uint32_t len = s[i].value.inlined.length;
      payload += len;
      if (len <= 12) continue;
      const uint64_t p = s[i].value.pointer.ptr;
      long_bytes += len;
      if (region_lo == 0) region_lo = p;
      if (p < prev_end || p < heap_lo || p > heap_hi || len > heap_hi - p) ascending_inside = false;
      prev_end = p + len;
    }
    region_hi = prev_end;
    if (long_bytes > 0) {
      // one copy of [first long string, end of the last) when that range is known to be one allocation and is not much
      // larger than what it is needed for (short and NULL rows in between own at most 12 bytes each there)
      one_copy = ascending_inside && region_hi - region_lo <= static_cast<uint64_t>(long_bytes) + 16ull * static_cast<uint64_t>(n);
      extra_heap = one_copy ? static_cast<int64_t>(region_hi - region_lo) : long_bytes;
    }
  }
  Reserve(columns[static_cast<size_t>(ni)], columns[static_cast<size_t>(ni)].count + n, extra_heap);
  Column& c = columns[static_cast<size_t>(ni)];  // (children are appended after this block: `columns` never grows here)
  AppendBits(c.validity, c.count, v.validity, vbit, n);
  if (v.validity && !c.has_nulls) {
    for (int64_t i = 0; i < n; i++)
      if (!BitAt(v.validity, vbit + i)) { c.has_nulls = true; break; }
  }
  if (c.IsGroup()) {
    const int64_t mult = c.arrow_type == MI_AT_FIXED_LIST ? c.param : 1;
    c.count += n;
    for (size_t k = 0; k < c.children.size(); k++) AppendNode(c.children[k], v.children[k], start * mult, n * mult);
    return;
  }
  if (c.IsList()) {
    // the list_entry_t rows are staged as they are (the GPU turns the lengths of the valid rows into Arrow offsets);
    // the child rows of every valid list are gathered here, in list order (ArrowListData::Append)
    const uint64_t* ent = static_cast<const uint64_t*>(v.data) + 2 * start;
    std::memcpy(c.data + static_cast<size_t>(c.count) * 16, ent, static_cast<size_t>(n) * 16);
    int64_t run_start = 0, run_len = 0;
    const int32_t child = c.children[0];
    c.count += n;
    size_in_bytes += n * 16;
    for (int64_t i = 0; i < n; i++) {
      if (!BitAt(v.validity, vbit + i)) continue;
      const int64_t o = static_cast<int64_t>(ent[2 * i]), l = static_cast<int64_t>(ent[2 * i + 1]);
      if (l <= 0) continue;
      if (run_len > 0 && o == run_start + run_len) {
        run_len += l;
      } else {
        if (run_len > 0) AppendNode(child, v.children[0], run_start, run_len);
        run_start = o;
        run_len = l;
      }
      columns[static_cast<size_t>(ni)].payload_bytes += l;   // child rows so far = the last Arrow offset
      if (columns[static_cast<size_t>(ni)].payload_bytes > 0x7FFFFFFFll && !columns[static_cast<size_t>(ni)].large_offsets)
        throw InvalidInputException("Arrow Appender: The maximum combined list offset for regular list buffers is 2147483647 but the offset of " +
                                    std::to_string(columns[static_cast<size_t>(ni)].payload_bytes) +
                                    " exceeds this.\n* SET arrow_large_buffer_size=true to use large list buffers");
    }
    if (run_len > 0) AppendNode(child, v.children[0], run_start, run_len);
    return;
  }
  std::memcpy(c.data + static_cast<size_t>(c.count) * static_cast<size_t>(c.width),
              static_cast<const uint8_t*>(v.data) + static_cast<size_t>(start) * static_cast<size_t>(c.width),
              static_cast<size_t>(n) * static_cast<size_t>(c.width));
  if (c.enc_kind == MI_K_ENC_STR32) {
    c.payload_bytes += payload;
    if (extra_heap > 0) {
      mi_string_t* dst = reinterpret_cast<mi_string_t*>(c.data) + c.count;
      if (one_copy) std::memcpy(c.heap + c.heap_used, reinterpret_cast<const void*>(static_cast<uintptr_t>(region_lo)), static_cast<size_t>(extra_heap));
      int64_t at = c.heap_used;
      for (int64_t i = 0; i < n; i++) {   // the staged rows point at heap offsets
        if (!BitAt(v.validity, vbit + i)) continue;
        const uint32_t len = dst[i].value.inlined.length;
        if (len <= 12) continue;
        const uint64_t p = dst[i].value.pointer.ptr;
        if (one_copy) {
          dst[i].value.pointer.ptr = static_cast<uint64_t>(c.heap_used) + (p - region_lo);
        } else {
          std::memcpy(c.heap + at, reinterpret_cast<const void*>(static_cast<uintptr_t>(p)), len);
          dst[i].value.pointer.ptr = static_cast<uint64_t>(at);
          at += len;
        }
      }
      c.heap_used += extra_heap;
    }
    size_in_bytes += extra_heap;
  }
  size_in_bytes += n * c.width;
  c.count += n;
}

void ChunkCollection::Reset() {
  for (auto& c : columns) {
    c.count = 0;
    c.heap_used = 0;
    c.ptr_base = 0;
    c.payload_bytes = 0;
    c.has_nulls = false;
    if (c.validity) std::memset(c.validity, 0xFF, c.validity_cap);
  }
  count = 0;
  size_in_bytes = 0;
}

// ------------------------------------------------------------------------------------------------ serializer
ColumnDataCollectionSerializer::ColumnDataCollectionSerializer(Context* ctx_p, bool own_stream) : ctx(ctx_p) {
  stream = ctx->stream;
  if (own_stream) {
    ctx->Bind();
    MI_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    owns_stream = true;
  }
}

ColumnDataCollectionSerializer::~ColumnDataCollectionSerializer() {
  h_bodies[cur_body] = h_body;
  for (auto* b : h_bodies)
    if (b) (void)hipHostFree(b);
  plan.reset();
  if (d_body) (void)hipFree(d_body);
  if (d_in) (void)hipFree(d_in);
  if (owns_stream && stream) (void)hipStreamDestroy(stream);
}

void ColumnDataCollectionSerializer::Init(const ArrowSchemaModel* schema_p) { schema = schema_p; }

void ColumnDataCollectionSerializer::SerializeSchema() {
  header = EncodeSchemaMessage(*schema);
  body_size = 0;
}

idx_t ColumnDataCollectionSerializer::Serialize(ChunkCollection& buffer) {
  ScopedTimer timer(&Timers().serialize);
  header.clear();
  body_size = 0;
  const int64_t n_top = buffer.Count();
  if (n_top == 0) return 0;
  ctx->Bind();
  if (!plan) plan = std::make_unique<Plan>(ctx);
  if (n_top > 0x7FFFFFFFll) throw InvalidInputException("record batch too large");

  // body layout: field nodes depth first, per node validity, then (offsets) / data -- the order
  // ArrowIpcEncoderEncodeSimpleRecordBatch walks the ArrowArray tree; device staging layout beside it
  const size_t n_nodes = buffer.columns.size();
  std::vector<mi_buffer_span> spans;
  struct InOff { size_t data, validity, heap; size_t first_span; };
  std::vector<InOff> in_off(n_nodes);
  size_t body_off = 0, in_bytes = 0;
  auto add_span = [&](int64_t len) {
    spans.push_back(mi_buffer_span{static_cast<int64_t>(body_off), len});
    body_off += RoundUp(static_cast<size_t>(len), kBufferAlign);
  };
  for (size_t ci = 0; ci < n_nodes; ci++) {
    auto& c = buffer.columns[ci];
    const int64_t n = c.count;
    if (n > 0x7FFFFFFFll) throw InvalidInputException("record batch too large");
    in_off[ci].first_span = spans.size();
    add_span((n + 7) / 8);  // validity: always emitted (ArrowAppender::FinalizeChild)
    const int64_t off_width = c.large_offsets ? 8 : 4;
    if (c.IsList()) {
      add_span((n + 1) * off_width);
    } else if (!c.IsGroup()) {
      switch (c.enc_kind) {
        case MI_K_ENC_COPY: add_span(n * c.param); break;
        case MI_K_ENC_DEC128: add_span(n * 16); break;
        case MI_K_ENC_BOOL: add_span((n + 7) / 8); break;
        case MI_K_ENC_STR32:
          if (c.payload_bytes > 0x7FFFFFFFll && !c.large_offsets) {
            throw InvalidInputException(
                "Arrow Appender: The maximum total string size for regular string buffers is 2147483647 but the offset of " +
                std::to_string(c.payload_bytes) + " exceeds this.\n* SET arrow_large_buffer_size=true to use large string buffers");
          }
          add_span((n + 1) * off_width);
          add_span(c.payload_bytes);
          break;
        default: break;
      }
    }
    const int64_t staged_rows = n;
    in_off[ci].data = in_bytes;
    in_bytes += RoundUp(static_cast<size_t>(staged_rows) * static_cast<size_t>(c.width) + 16, 256);
    in_off[ci].validity = in_bytes;
    in_bytes += RoundUp(static_cast<size_t>((n + 63) / 64) * 8 + 8, 256);
    in_off[ci].heap = in_bytes;
    in_bytes += RoundUp(static_cast<size_t>(c.heap_used) + 16, 256);
  }
  body_size = static_cast<int64_t>(body_off);
  GrowDevice(&d_in, &d_in_cap, in_bytes + 256);
  GrowDevice(&d_body, &d_body_cap, body_off + 256);
  { size_t zero = 0; GrowPinned(&h_body, &h_body_cap, body_off + 256, zero); }

  hipStream_t s = stream;
  MI_HIP_CHECK(hipMemsetAsync(d_body, 0, body_off, s));  // the padding bytes of every buffer are zero
  std::vector<mi_col_task> tasks;
  std::vector<int32_t> validity_task(n_nodes, -1);  // task whose NULL counter belongs to node ci
  for (size_t ci = 0; ci < n_nodes; ci++) {
    auto& c = buffer.columns[ci];
    const int64_t n = c.count;
    if (n == 0) continue;  // zero-length buffers, no work
    const size_t sp = in_off[ci].first_span;
    const int64_t staged_rows = n;
    if (c.width > 0)
      MI_HIP_CHECK(hipMemcpyAsync(d_in + in_off[ci].data, c.data, static_cast<size_t>(staged_rows) * static_cast<size_t>(c.width), hipMemcpyHostToDevice, s));
    if (c.has_nulls)
      MI_HIP_CHECK(hipMemcpyAsync(d_in + in_off[ci].validity, c.validity, static_cast<size_t>((n + 63) / 64) * 8, hipMemcpyHostToDevice, s));
    if (c.heap_used)
      MI_HIP_CHECK(hipMemcpyAsync(d_in + in_off[ci].heap, c.heap, static_cast<size_t>(c.heap_used), hipMemcpyHostToDevice, s));
    mi_col_task t;
    std::memset(&t, 0, sizeof(t));
    t.nrows = n;
    t.validity = c.has_nulls ? d_in + in_off[ci].validity : nullptr;
    t.out_validity = d_body + spans[sp].offset;
    validity_task[ci] = static_cast<int32_t>(tasks.size());
    if (c.IsGroup()) {
      // struct / fixed_size_list: the node's own bitmap + NULL count
      t.kind = MI_K_ENC_VALIDITY;
      t.buf1 = d_in + in_off[ci].validity;
      t.out_data = d_body + spans[sp].offset;
      tasks.push_back(t);
      continue;
    }
    t.flags = c.large_offsets ? 1 : 0;
    if (c.IsList()) {  // list / map: bitmap + int32 (or int64) offsets from the staged list_entry_t rows
      t.kind = MI_K_ENC_LIST32;
      t.buf1 = d_in + in_off[ci].data;
      t.out_data = d_body + spans[sp + 1].offset;
      tasks.push_back(t);
      continue;
    }
    t.kind = c.enc_kind;
    t.param = c.param;
    t.buf1 = d_in + in_off[ci].data;
    t.out_data = d_body + spans[sp + 1].offset;
    if (c.enc_kind == MI_K_ENC_STR32) {
      t.buf2 = d_in + in_off[ci].heap;
      t.buf2_len = c.payload_bytes;
      t.ptr_base = c.ptr_base;
      t.out_aux = d_body + spans[sp + 2].offset;
    }
    tasks.push_back(t);
  }
  plan->Set(tasks.data(), static_cast<int32_t>(tasks.size()), s);
  plan->Launch(s);
  MI_HIP_CHECK(hipMemcpyAsync(h_body, d_body, body_off, hipMemcpyDeviceToHost, s));
  ThrowForStatus(plan->Status());  // synchronises the stream
  std::vector<int64_t> null_counts = plan->NullCounts(/*reset*/ true);
  std::vector<std::pair<int64_t, int64_t>> nodes;
  for (size_t ci = 0; ci < n_nodes; ci++)
    nodes.emplace_back(buffer.columns[ci].count, validity_task[ci] >= 0 ? null_counts[static_cast<size_t>(validity_task[ci])] : 0);
  header = EncodeRecordBatchMessage(n_top, nodes, spans, body_size);
  return 1;
}

// ------------------------------------------------------------------------------------------------ stream writer
ArrowStreamWriter::ArrowStreamWriter(Context* ctx_p, const std::string& file_path, const std::vector<ArrowField>& fields,
                                     const std::vector<std::pair<std::string, std::string>>& metadata)
    : ctx(ctx_p), serializer(ctx_p), file_name(file_path) {
  InitSchema(fields, metadata);
  if (!file_path.empty()) InitOutputFile(file_path);
}

ArrowStreamWriter::~ArrowStreamWriter() {
  {
    std::lock_guard<std::mutex> lk(io_mu);
    io_stop = true;
  }
  io_cv.notify_all();
  if (io_thread.joinable()) io_thread.join();  // queued batches are still written
  if (fd >= 0) ::close(fd);
}

void ArrowStreamWriter::InitSchema(const std::vector<ArrowField>& fields,
                                   const std::vector<std::pair<std::string, std::string>>& metadata) {
  schema.fields = fields;
  schema.metadata = metadata;  // kv_metadata COPY option (arrow_stream_writer.cpp:26-44)
  serializer.Init(&schema);
}

void ArrowStreamWriter::InitOutputFile(const std::string& file_path) {
  // FILE_FLAGS_WRITE | FILE_FLAGS_FILE_CREATE_NEW (arrow_stream_writer.cpp:49-53): always a fresh file
  fd = ::open(file_path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) throw IOException("Cannot open file \"" + file_path + "\": " + std::strerror(errno));
  // Record batches are pwritten at claimed offsets by whichever thread serialized them.  (On tmpfs the page cache IS the
  // file and every new page is allocated, charged and zeroed under the write: measured on the MI355X box ~5-6 GB/s into one
  // file whether 1, 2, 4 or 6 threads pwrite and whether they write() or store through a shared mapping -- the end-to-end
  // COPY of SF10 is bound by that, not by staging or the K7 kernels: tools/copy_bench.py)
}

void ArrowStreamWriter::WriteAt(int64_t offset, const uint8_t* p, size_t n) {
  ScopedTimer timer(&Timers().write);
  size_t done = 0;
  while (done < n) {
    ssize_t w = ::pwrite(fd, p + done, n - done, static_cast<off_t>(offset + static_cast<int64_t>(done)));
    if (w < 0) {
      if (errno == EINTR) continue;
      throw IOException("Could not write to file \"" + file_name + "\": " + std::strerror(errno));
    }
    done += static_cast<size_t>(w);
  }
}

int64_t ArrowStreamWriter::ReserveRowGroup(size_t bytes) {
  std::lock_guard<std::mutex> lk(io_mu);
  const int64_t at = static_cast<int64_t>(total_written);
  total_written += bytes;
  ++row_group_count;
  return at;
}

void ArrowStreamWriter::WriteData(const uint8_t* p, size_t n) {
  int64_t at;
  {
    std::lock_guard<std::mutex> lk(io_mu);
    at = static_cast<int64_t>(total_written);
    total_written += n;
  }
  WriteAt(at, p, n);
}

void ArrowStreamWriter::WriteSchema() {
  serializer.SerializeSchema();
  WriteData(serializer.GetHeader().data(), serializer.GetHeader().size());
}

void ArrowStreamWriter::IoLoop() {
  while (true) {
    WriteJob job;
    {
      std::unique_lock<std::mutex> lk(io_mu);
      io_cv.wait(lk, [&] { return io_stop || !io_jobs.empty(); });
      if (io_jobs.empty()) return;  // stop requested and nothing left
      job = std::move(io_jobs.front());
      io_jobs.pop_front();
    }
    try {
      if (!io_error) {
        WriteAt(job.offset, job.header.data(), job.header.size());
        WriteAt(job.offset + static_cast<int64_t>(job.header.size()), job.body, job.body_size);
      }
    } catch (...) {
      std::lock_guard<std::mutex> lk(io_mu);
      if (!io_error) io_error = std::current_exception();
    }
    {
      std::lock_guard<std::mutex> lk(io_mu);
      if (job.buffer >= 0) buffer_busy[job.buffer] = false;
    }
    io_cv.notify_all();
  }
}

void ArrowStreamWriter::WaitBufferFree(int buffer) {
  std::unique_lock<std::mutex> lk(io_mu);
  io_cv.wait(lk, [&] { return !buffer_busy[buffer]; });
  if (io_error) std::rethrow_exception(io_error);
}

void ArrowStreamWriter::DrainIo() {
  std::unique_lock<std::mutex> lk(io_mu);
  io_cv.wait(lk, [&] { return io_jobs.empty() && !buffer_busy[0] && !buffer_busy[1]; });
  if (io_error) std::rethrow_exception(io_error);
}

void ArrowStreamWriter::Flush(ChunkCollection& buffer) {
  // Serialize() writes into the serializer's current body buffer: it must not be in the I/O thread's hands any more
  WaitBufferFree(serializer.CurrentBody());
  if (serializer.Serialize(buffer) == 0) {
    buffer.Reset();
    CountEmptyFlush();  // the reference counts the flush even when the collection was empty (arrow_stream_writer.cpp:66-71)
    return;
  }
  buffer.Reset();
  WriteJob job;
  job.header = serializer.GetHeader();
  job.body = serializer.GetBody();
  job.body_size = static_cast<size_t>(serializer.GetBodySize());
  job.offset = ReserveRowGroup(job.header.size() + job.body_size);
  {
    std::lock_guard<std::mutex> lk(io_mu);
    job.buffer = serializer.SwapBody();
    buffer_busy[job.buffer] = true;
    io_jobs.push_back(std::move(job));
    if (!io_thread.joinable()) io_thread = std::thread([this] { IoLoop(); });
  }
  io_cv.notify_all();
}

void ArrowStreamWriter::Finalize() {
  if (finalized) return;
  DrainIo();
  const uint8_t end_of_stream[] = {0xFF, 0xFF, 0xFF, 0xFF, 0x00, 0x00, 0x00, 0x00};
  WriteData(end_of_stream, sizeof(end_of_stream));
  ::close(fd);
  fd = -1;
  finalized = true;
  if (Timers().on)
    std::fprintf(stderr, "[mi_writer] append %.3f s, serialize (H2D + K7 + D2H) %.3f s, write (I/O thread) %.3f s\n", Timers().append,
                 Timers().serialize, Timers().write);
}

}  // namespace miarrow

// ------------------------------------------------------------------------------------------------ C ABI
using namespace miarrow;

namespace miarrow {
Context* ContextOf(mi_ctx* c);
}

namespace miarrow {
ArrowScan* SingleScanOf(mi_scan* s);  // scan_operator.cpp
}

struct mi_writer {
  Context* ctx = nullptr;
  mi_write_options opts;
  std::vector<ArrowField> fields;
  std::unique_ptr<ArrowStreamWriter> writer;        // COPY TO file
  std::unique_ptr<ChunkCollection> buffer;
  // to_arrow_ipc mode
  ArrowSchemaModel schema;
  std::unique_ptr<ColumnDataCollectionSerializer> serializer;
  std::vector<uint8_t> blob;
};

namespace {
std::string LowerStr(std::string s) {
  for (auto& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
  return s;
}
std::string UpperStr(std::string s) {
  for (auto& c : s) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
  return s;
}

// DBConfig::ParseMemoryLimit-style sizes: "100", "1KB", "2 MiB", "1gb"
int64_t ParseMemory(const std::string& v) {
  char* end = nullptr;
  double num = std::strtod(v.c_str(), &end);
  if (end == v.c_str() || num < 0) throw InvalidInputException("Could not parse memory size '" + v + "'");
  std::string unit;
  for (const char* p = end; *p; p++)
    if (!std::isspace(static_cast<unsigned char>(*p))) unit += static_cast<char>(std::tolower(static_cast<unsigned char>(*p)));
  double mult = 1;
  if (unit.empty() || unit == "b" || unit == "byte" || unit == "bytes") mult = 1;
  else if (unit == "kb" || unit == "k") mult = 1000.0;
  else if (unit == "mb" || unit == "m") mult = 1000.0 * 1000;
  else if (unit == "gb" || unit == "g") mult = 1000.0 * 1000 * 1000;
  else if (unit == "tb" || unit == "t") mult = 1000.0 * 1000 * 1000 * 1000;
  else if (unit == "kib") mult = 1024.0;
  else if (unit == "mib") mult = 1024.0 * 1024;
  else if (unit == "gib") mult = 1024.0 * 1024 * 1024;
  else if (unit == "tib") mult = 1024.0 * 1024 * 1024 * 1024;
  else throw InvalidInputException("Unknown unit for memory size: '" + unit + "'");
  return static_cast<int64_t>(num * mult);
}

uint64_t ParseU64(const std::string& name, const std::string& v) {
  char* end = nullptr;
  errno = 0;
  unsigned long long x = std::strtoull(v.c_str(), &end, 10);
  if (end == v.c_str() || *end != 0 || errno != 0 || (!v.empty() && v[0] == '-'))
    throw InvalidInputException("Could not convert string '" + v + "' to UINT64 for option " + UpperStr(name));
  return x;
}

// arrow_large_buffer_size: VARCHAR / BLOB / LIST export with 64-bit offsets (ArrowConverter::ToArrowSchema with
// ArrowOffsetSize::LARGE); MAP keeps int32 offsets (the Arrow format has no large map)
void MakeLarge(ArrowField& f) {
  if (f.type == MI_AT_UTF8) f.type = MI_AT_LARGE_UTF8;
  else if (f.type == MI_AT_BINARY) f.type = MI_AT_LARGE_BINARY;
  else if (f.type == MI_AT_LIST) f.type = MI_AT_LARGE_LIST;
  for (auto& c : f.children) MakeLarge(c);
}

std::vector<ArrowField> FieldsFromC(const mi_field* fields, int32_t n_fields, bool large = false) {
  if (!fields || n_fields <= 0) throw InvalidInputException("writer needs at least one column");
  std::vector<ArrowField> out;
  for (int32_t i = 0; i < n_fields; i++) {
    out.push_back(FieldFromDuckType(fields[i].name, fields[i].duck_type));
    if (large) MakeLarge(out.back());
  }
  return out;
}
}  // namespace

extern "C" {

int mi_write_options_init(mi_write_options* o) {
  return WrapC([&] {
    if (!o) throw InvalidInputException("mi_write_options_init: NULL");
    std::memset(o, 0, sizeof(*o));
    o->row_group_size = 122880;
    o->preserve_insertion_order = 1;
  });
}

int mi_write_options_set(mi_write_options* o, const char* name, const char* value) {
  return WrapC([&] {
    if (!o || !name) throw InvalidInputException("mi_write_options_set: NULL");
    const std::string loption = LowerStr(name);
    if (!value) throw BinderException(UpperStr(loption) + " requires exactly one argument");
    if (loption == "row_group_size" || loption == "chunk_size") {
      if (o->row_group_size_set) throw BinderException("ROW_GROUP_SIZE and ROW_GROUP_SIZE_BYTES are mutually exclusive");
      o->row_group_size = static_cast<int64_t>(ParseU64(loption, value));
      o->row_group_size_set = 1;
    } else if (loption == "row_group_size_bytes") {
      o->row_group_size_bytes = ParseMemory(value);
      o->row_group_size_bytes_set = 1;
    } else if (loption == "row_groups_per_file") {
      o->row_groups_per_file = static_cast<int64_t>(ParseU64(loption, value));
    }
    // other options are not ours: the bind loop ignores them (write_arrow_stream.cpp:62-105)
  });
}

int mi_write_options_add_kv(mi_write_options* o, const char* key, const char* value, int32_t value_len) {
  return WrapC([&] {
    if (!o || !key || !value) throw InvalidInputException("mi_write_options_add_kv: NULL");
    if (o->n_kv_metadata >= MI_MAX_KV_METADATA) throw InvalidInputException("too many kv_metadata entries");
    if (value_len < 0) value_len = static_cast<int32_t>(std::strlen(value));
    if (std::strlen(key) >= sizeof(o->kv_keys[0]) || static_cast<size_t>(value_len) > sizeof(o->kv_values[0]))
      throw InvalidInputException("kv_metadata entry too long");
    std::snprintf(o->kv_keys[o->n_kv_metadata], sizeof(o->kv_keys[0]), "%s", key);
    std::memcpy(o->kv_values[o->n_kv_metadata], value, static_cast<size_t>(value_len));
    o->kv_value_lens[o->n_kv_metadata] = value_len;
    o->n_kv_metadata++;
  });
}

int mi_write_options_finalize(mi_write_options* o) {
  return WrapC([&] {
    if (!o) throw InvalidInputException("mi_write_options_finalize: NULL");
    if (o->row_group_size_bytes_set) {
      if (o->preserve_insertion_order) {
        throw BinderException(
            "ROW_GROUP_SIZE_BYTES does not work while preserving insertion order. Use \"SET "
            "preserve_insertion_order=false;\" to disable preserving insertion order.");
      }
    } else {
      // We always set a max row group size bytes so we don't use too much memory
      o->row_group_size_bytes = o->row_group_size * 1024;
    }
  });
}

int mi_encode_schema(const mi_field* fields, int32_t n_fields, uint8_t* out, int64_t cap, int64_t* size) {
  return WrapC([&] {
    if (!size) throw InvalidInputException("mi_encode_schema: NULL argument");
    ArrowSchemaModel schema;
    schema.fields = FieldsFromC(fields, n_fields);
    const std::vector<uint8_t> msg = EncodeSchemaMessage(schema);
    *size = static_cast<int64_t>(msg.size());
    if (out && cap >= *size) std::memcpy(out, msg.data(), msg.size());
  });
}

int mi_writer_open(mi_ctx* ctx, const char* path, const mi_field* fields, int32_t n_fields, const mi_write_options* opts,
                   mi_writer** out) {
  return WrapC([&] {
    if (!ctx || !path || !out) throw InvalidInputException("mi_writer_open: NULL argument");
    auto w = std::make_unique<mi_writer>();
    w->ctx = ContextOf(ctx);
    if (opts) {
      w->opts = *opts;
    } else {
      mi_write_options_init(&w->opts);
      mi_write_options_finalize(&w->opts);
    }
    if (w->opts.row_group_size_bytes <= 0) w->opts.row_group_size_bytes = w->opts.row_group_size * 1024;
    w->fields = FieldsFromC(fields, n_fields, w->opts.arrow_large_buffer_size != 0);
    std::vector<std::pair<std::string, std::string>> kv;
    for (int32_t i = 0; i < w->opts.n_kv_metadata; i++)
      kv.emplace_back(w->opts.kv_keys[i], std::string(w->opts.kv_values[i], static_cast<size_t>(w->opts.kv_value_lens[i])));
    w->buffer = std::make_unique<ChunkCollection>(w->ctx, w->fields);
    w->writer = std::make_unique<ArrowStreamWriter>(w->ctx, path, w->fields, kv);
    w->writer->WriteSchema();
    *out = w.release();
  });
}

int mi_writer_sink(mi_writer* w, const mi_data_chunk* chunk) {
  return WrapC([&] {
    if (!w || !chunk || !w->writer) throw InvalidInputException("mi_writer_sink: bad argument");
    // append data to the local (buffered) chunk collection; flush when it exceeds the row / byte budget
    w->buffer->Append(*chunk);
    if (w->buffer->Count() >= w->opts.row_group_size || w->buffer->SizeInBytes() >= w->opts.row_group_size_bytes) {
      w->writer->Flush(*w->buffer);
    }
  });
}

// ---- per-thread sink state (ArrowWriteInitializeLocal / Sink / Combine, write_arrow_stream.cpp:141-174): every sink
// thread buffers its own chunks AND serializes its own row groups (H2D + K7 + D2H on a stream of its own), so staging,
// encoding and writing of different row groups overlap; only the claim of the file range is serialised.
}  // extern "C"

struct mi_writer_local {
  mi_writer* w = nullptr;
  std::unique_ptr<ChunkCollection> buffer;
  std::unique_ptr<ColumnDataCollectionSerializer> serializer;
  //! serializes the buffered rows as one record batch and writes it at the next free position of the file
  void FlushRowGroup(const std::function<void()>& before_claim = nullptr, const std::function<void()>& after_claim = nullptr) {
    ArrowStreamWriter& out = *w->writer;
    if (serializer->Serialize(*buffer) == 0) {
      buffer->Reset();
      if (before_claim) before_claim();
      out.CountEmptyFlush();
      if (after_claim) after_claim();
      return;
    }
    buffer->Reset();
    const auto& header = serializer->GetHeader();
    const size_t body = static_cast<size_t>(serializer->GetBodySize());
    if (before_claim) before_claim();   // ordered sinks wait for their turn here
    const int64_t at = out.ReserveRowGroup(header.size() + body);
    if (after_claim) after_claim();
    out.WriteAt(at, header.data(), header.size());
    out.WriteAt(at + static_cast<int64_t>(header.size()), serializer->GetBody(), body);
  }
};

namespace {
std::unique_ptr<mi_writer_local> MakeLocal(mi_writer* w) {
  auto l = std::make_unique<mi_writer_local>();
  l->w = w;
  l->buffer = std::make_unique<ChunkCollection>(w->ctx, w->fields);
  l->serializer = std::make_unique<ColumnDataCollectionSerializer>(w->ctx, /*own_stream*/ true);
  l->serializer->Init(&w->writer->Schema());
  return l;
}

int SinkThreads() {
  const char* v = std::getenv("MI_WRITER_THREADS");
  if (v) return std::max(1, std::min(16, std::atoi(v)));
  const int hw = static_cast<int>(std::thread::hardware_concurrency());
  return std::max(1, std::min(6, hw / 3));
}

// COPY (FROM read_arrow(...)) TO 'file': the pump DuckDB's executor is between a scan and a copy sink, with the batch
// copy's re-partitioning (PhysicalBatchCopyToFile hands prepare_batch collections of desired_batch_size = row_group_size
// rows, write_arrow_stream.cpp:225-245).  The pump thread pulls whole record batches from the scan and cuts the stream of
// their 2048-row chunks into row groups exactly where the one-thread sink would flush (after the chunk that reaches
// row_group_size rows / row_group_size_bytes); each row group goes to one of T sink threads which appends its chunks,
// encodes it and writes it -- claims of the file range happen in row-group order, so the file equals the one-thread file.
void PumpScanParallel(mi_writer* w, ArrowScan* scan, int threads, int64_t* rows_out, const BatchRef* first = nullptr) {
  struct Piece { int batch; int32_t w0, w1; };        // windows [w0, w1) of held batch `batch`
  // `spilled`: rows of this row group the pump has already staged itself (see spill_cur below); the worker that takes the
  // job appends the remaining pieces to it and flushes it instead of its own state
  struct Job { std::vector<Piece> pieces; int64_t seq = 0; std::unique_ptr<mi_writer_local> spilled; };
  struct Held { BatchRef ref; int pieces_open = 0; bool fully_cut = false; };
  if (!first) scan->EnsurePipelineDepth(threads + 4);   // with a batch already acquired the caller has done it
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Job> jobs;
  std::vector<Held> held;             // index = batch token
  std::deque<int> to_release;         // tokens whose last piece was appended (released by the pump thread: it owns the scan)
  int64_t next_claim = 0;             // sequence number of the row group that may claim its file range next
  bool done = false;
  std::exception_ptr error;
  std::vector<std::thread> workers;
  auto fail = [&](std::exception_ptr e) {
    std::lock_guard<std::mutex> lk(mu);
    if (!error) error = e;
    cv.notify_all();
  };
  for (int t = 0; t < threads; t++) {
    workers.emplace_back([&] {
      try {
        auto local = MakeLocal(w);
        ChunkStorage storage;
        mi_data_chunk chunk;
        while (true) {
          Job job;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return error || done || !jobs.empty(); });
            if (error) return;
            if (jobs.empty()) return;  // done
            job = std::move(jobs.front());
            jobs.pop_front();
          }
          mi_writer_local* sink = job.spilled ? job.spilled.get() : local.get();
          for (const Piece& pc : job.pieces) {
            BatchRef ref;
            {
              std::lock_guard<std::mutex> lk(mu);
              ref = held[static_cast<size_t>(pc.batch)].ref;
            }
            for (int32_t wi = pc.w0; wi < pc.w1; wi++) {
              scan->BuildChunk(ref, wi, &storage, &chunk);
              sink->buffer->Append(chunk);
            }
            std::lock_guard<std::mutex> lk(mu);
            Held& h = held[static_cast<size_t>(pc.batch)];
            if (--h.pieces_open == 0 && h.fully_cut) {
              to_release.push_back(pc.batch);
              cv.notify_all();
            }
          }
          sink->FlushRowGroup(
              [&] {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return error || next_claim == job.seq; });
              },
              [&] {
                std::lock_guard<std::mutex> lk(mu);
                next_claim = job.seq + 1;
                cv.notify_all();
              });
        }
      } catch (...) {
        fail(std::current_exception());
      }
    });
  }
  int64_t rows = 0, seq = 0;
  try {
    Job cur;
    int64_t cur_rows = 0, cur_bytes = 0;
    const int64_t row_bytes = std::max<int64_t>(1, [&] {   // staged bytes per row, for row_group_size_bytes
      int64_t b = 0;
      for (auto& c : scan->OutputColumns()) {
        int32_t kind, wd, nb;
        int64_t param;
        if (!c.is_filename && !c.is_hive && c.field.Plan(&kind, &param, &wd, &nb)) b += wd; else b += 16;
      }
      return b;
    }());
    int64_t n_released = 0;           // batches given back to the scan so far
    auto dispatch = [&] {
      if (cur.pieces.empty() && !cur.spilled) return;
      cur.seq = seq++;
      {
        std::lock_guard<std::mutex> lk(mu);
        jobs.push_back(std::move(cur));
      }
      cv.notify_all();
      cur = Job();
      cur_rows = cur_bytes = 0;
    };
    auto release_ready = [&](bool wait) {
      std::unique_lock<std::mutex> lk(mu);
      if (wait) cv.wait(lk, [&] { return error || !to_release.empty(); });
      if (error) std::rethrow_exception(error);
      while (!to_release.empty()) {
        const int tok = to_release.front();
        to_release.pop_front();
        const BatchRef ref = held[static_cast<size_t>(tok)].ref;
        lk.unlock();
        scan->ReleaseBatch(ref);
        n_released++;
        lk.lock();
      }
    };
    // A row group that spans more record batches than the scan has slots: every slot is held by a piece of the row group
    // still being cut, which no sink thread will see before it is full.  The pump then stages those rows itself (a sink
    // state of its own that travels with the job) and gives the slots back; the worker that gets the job appends the rest.
    ChunkStorage spill_storage;
    mi_data_chunk spill_chunk;
    auto spill_cur = [&] {
      if (!cur.spilled) cur.spilled = MakeLocal(w);
      for (const Piece& pc : cur.pieces) {
        const BatchRef ref = held[static_cast<size_t>(pc.batch)].ref;   // only the pump thread grows `held`
        for (int32_t wi = pc.w0; wi < pc.w1; wi++) {
          scan->BuildChunk(ref, wi, &spill_storage, &spill_chunk);
          cur.spilled->buffer->Append(spill_chunk);
        }
        bool give_back;
        {
          std::lock_guard<std::mutex> lk(mu);
          Held& h = held[static_cast<size_t>(pc.batch)];
          give_back = --h.pieces_open == 0 && h.fully_cut;
        }
        if (give_back) {
          scan->ReleaseBatch(ref);
          n_released++;
        }
      }
      cur.pieces.clear();
    };
    while (true) {
      release_ready(false);
      BatchRef ref;
      if (first) {
        ref = *first;
        first = nullptr;
      } else if (!scan->AcquireBatch(&ref)) {
        if (scan->Exhausted()) break;
        // every slot is held.  Batches whose pieces all went to sink threads come back by themselves; the ones that only
        // the undispatched row group refers to never would
        std::vector<int> cur_toks;
        for (const Piece& pc : cur.pieces)
          if (std::find(cur_toks.begin(), cur_toks.end(), pc.batch) == cur_toks.end()) cur_toks.push_back(pc.batch);
        bool with_workers;
        {
          std::lock_guard<std::mutex> lk(mu);
          with_workers = static_cast<int64_t>(held.size()) - n_released - static_cast<int64_t>(cur_toks.size()) > 0;
        }
        if (with_workers) release_ready(true);
        else if (!cur.pieces.empty()) spill_cur();
        else throw InternalException("COPY pump: no record batch can be acquired and none is held");
        continue;
      }
      scan->EnsureHostVectors(ref);
      rows += ref.chunk_rows;
      int tok;
      {
        std::lock_guard<std::mutex> lk(mu);
        held.push_back(Held{ref, 0, false});
        tok = static_cast<int>(held.size() - 1);
      }
      if (ref.n_windows == 0) {
        std::lock_guard<std::mutex> lk(mu);
        held[static_cast<size_t>(tok)].fully_cut = true;
        to_release.push_back(tok);
        continue;
      }
      int32_t w0 = 0;
      for (int32_t wi = 0; wi < ref.n_windows; wi++) {
        const int64_t n = std::min<int64_t>(MI_VECTOR_SIZE, ref.chunk_rows - static_cast<int64_t>(wi) * MI_VECTOR_SIZE);
        cur_rows += n;
        cur_bytes += n * row_bytes;
        const bool full = cur_rows >= w->opts.row_group_size || cur_bytes >= w->opts.row_group_size_bytes;
        if (full || wi + 1 == ref.n_windows) {
          {
            std::lock_guard<std::mutex> lk(mu);
            held[static_cast<size_t>(tok)].pieces_open++;
            if (wi + 1 == ref.n_windows) held[static_cast<size_t>(tok)].fully_cut = true;
          }
          cur.pieces.push_back(Piece{tok, w0, wi + 1});
          w0 = wi + 1;
          if (full) dispatch();
        }
      }
    }
    dispatch();   // the tail row group (ArrowWriteCombine)
  } catch (...) {
    fail(std::current_exception());
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    done = true;
  }
  cv.notify_all();
  for (auto& t : workers) t.join();
  if (Timers().on)
    std::fprintf(stderr, "[mi_writer] pump with %d sink threads (thread-seconds): append %.3f, serialize (H2D + K7 + D2H) %.3f, write %.3f\n", threads,
                 Timers().append, Timers().serialize, Timers().write);
  // give every batch back before reporting
  for (int tok : to_release) scan->ReleaseBatch(held[static_cast<size_t>(tok)].ref);
  if (error) std::rethrow_exception(error);
  if (rows_out) *rows_out = rows;
}

// ---- the fused COPY (FROM read_arrow(...)) TO 'file': decode and encode both run on the GPU, so the decoded vectors never
// have to leave HBM.  For every row group that lies inside one record batch of the scan the K7 kernels read the scan slot's
// vectors where the K1-K4 kernels wrote them (string payloads: the HBM copy of the Arrow data buffer the string_t rows
// point into) and write the IPC body; only that body travels back (one D2H) and is written by an I/O thread.  Per row this
// takes the host out of the loop except for pread -> H2D and D2H -> pwrite: no D2H of the vectors, no staging copy, no H2D
// of the staged rows (DESIGN.md section 10 has the byte counts).  Row groups are cut exactly where the one-thread sink cuts
// them (after the 2048-row chunk that reaches row_group_size); rows of a row group that straddles two record batches take
// the host path (EnsureHostVectors + ChunkCollection) on the pump thread, so the file is byte-identical to the
// one-thread file in every case.
struct FusedEncoder {
  uint8_t* d_body = nullptr;  size_t d_cap = 0;
  uint8_t* h_body = nullptr;  size_t h_cap = 0;
  int64_t* h_nulls = nullptr; size_t h_nulls_cap = 0;   // pinned copy of the plan's NULL counters
  uint32_t* h_status = nullptr;
  std::unique_ptr<Plan> plan;
  hipEvent_t encoded = nullptr, done = nullptr;
  bool busy = false;
  // the row group in flight
  int64_t nrows = 0, body_size = 0;
  std::vector<mi_buffer_span> spans;
  std::vector<int32_t> first_span;
};

bool FusedSinkPossible(mi_writer* w, ArrowScan* scan) {
  if (std::getenv("MI_WRITER_NO_FUSED")) return false;
  if (scan->HasFilter() || w->buffer->Count() != 0) return false;
  const auto& cols = scan->OutputColumns();
  if (cols.size() != w->buffer->roots.size() || w->buffer->columns.size() != cols.size()) return false;   // flat schema only
  for (size_t c = 0; c < cols.size(); c++) {
    if (cols[c].is_filename || cols[c].is_hive) return false;
    const auto& wc = w->buffer->columns[static_cast<size_t>(w->buffer->roots[c])];
    if (!wc.children.empty()) return false;
    if (wc.enc_kind != MI_K_ENC_COPY && wc.enc_kind != MI_K_ENC_DEC128 && wc.enc_kind != MI_K_ENC_BOOL && wc.enc_kind != MI_K_ENC_STR32) return false;
  }
  return true;
}

void PumpScanFused(mi_writer* w, ArrowScan* scan, const BatchRef& first, int64_t rows_per_group, int64_t* rows_out) {
  Context* ctx = w->ctx;
  ArrowStreamWriter& out = *w->writer;
  ctx->Bind();
  auto local = MakeLocal(w);             // host path of row groups that straddle record batches
  constexpr int kEncoders = 4;
  std::vector<FusedEncoder> enc(kEncoders);
  hipStream_t enc_stream = nullptr, back_stream = nullptr;
  // whatever ends this function, the device objects created below go away (the encoders' plans free themselves)
  struct Cleanup {
    std::vector<FusedEncoder>& enc;
    hipStream_t& enc_stream;
    hipStream_t& back_stream;
    ~Cleanup() {
      if (enc_stream) (void)hipStreamSynchronize(enc_stream);
      if (back_stream) (void)hipStreamSynchronize(back_stream);
      for (auto& e : enc) {
        e.plan.reset();
        if (e.d_body) (void)hipFree(e.d_body);
        if (e.h_body) (void)hipHostFree(e.h_body);
        if (e.h_nulls) (void)hipHostFree(e.h_nulls);
        if (e.h_status) (void)hipHostFree(e.h_status);
        if (e.encoded) (void)hipEventDestroy(e.encoded);
        if (e.done) (void)hipEventDestroy(e.done);
      }
      if (enc_stream) (void)hipStreamDestroy(enc_stream);
      if (back_stream) (void)hipStreamDestroy(back_stream);
    }
  } cleanup{enc, enc_stream, back_stream};
  MI_HIP_CHECK(hipStreamCreateWithFlags(&enc_stream, hipStreamNonBlocking));
  MI_HIP_CHECK(hipStreamCreateWithFlags(&back_stream, hipStreamNonBlocking));
  for (auto& e : enc) {
    e.plan = std::make_unique<Plan>(ctx);
    MI_HIP_CHECK(hipEventCreateWithFlags(&e.encoded, hipEventDisableTiming));
    MI_HIP_CHECK(hipEventCreateWithFlags(&e.done, hipEventDisableTiming));
    MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&e.h_status), 64, hipHostMallocDefault));
  }
  ChunkStorage storage;
  mi_data_chunk chunk;

  struct WriteJob {
    int enc = -1;                        // fused encoder; -1: `header` / `body` are ready (host-serialized row group)
    int tok = -1;                        // batch token whose open count drops once the GPU has read it
    std::vector<uint8_t> header;
    const uint8_t* body = nullptr;
    size_t body_size = 0;
  };
  struct Held { BatchRef ref; int open = 0; bool fully_cut = false; bool released = false; };
  std::mutex mu;
  std::condition_variable cv;
  std::deque<WriteJob> jobs;
  std::vector<Held> held;
  std::deque<int> to_release;
  bool stop = false;
  int64_t jobs_written = 0, jobs_queued = 0;
  std::exception_ptr error;
  const size_t n_cols = scan->NumOutputColumns();

  std::thread io([&] {
    try {
      ctx->Bind();
      while (true) {
        WriteJob job;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return stop || error || !jobs.empty(); });
          if (error || jobs.empty()) return;
          job = std::move(jobs.front());
          jobs.pop_front();
        }
        if (job.enc >= 0) {
          FusedEncoder& e = enc[static_cast<size_t>(job.enc)];
          MI_HIP_CHECK(hipEventSynchronize(e.done));
          {
            std::lock_guard<std::mutex> lk(mu);   // the GPU is done with the scan slot
            Held& h = held[static_cast<size_t>(job.tok)];
            if (--h.open == 0 && h.fully_cut && !h.released) {
              h.released = true;
              to_release.push_back(job.tok);
            }
          }
          cv.notify_all();
          ThrowForStatus(e.h_status[0]);
          const std::vector<int64_t> nulls = e.plan->MapNullCounts(e.h_nulls);
          std::vector<std::pair<int64_t, int64_t>> nodes;
          for (size_t c = 0; c < n_cols; c++) nodes.emplace_back(e.nrows, nulls[c]);
          job.header = EncodeRecordBatchMessage(e.nrows, nodes, e.spans, e.body_size);
          job.body = e.h_body;
          job.body_size = static_cast<size_t>(e.body_size);
        }
        const int64_t at = out.ReserveRowGroup(job.header.size() + job.body_size);
        out.WriteAt(at, job.header.data(), job.header.size());
        out.WriteAt(at + static_cast<int64_t>(job.header.size()), job.body, job.body_size);
        {
          std::lock_guard<std::mutex> lk(mu);
          if (job.enc >= 0) enc[static_cast<size_t>(job.enc)].busy = false;
          ++jobs_written;
        }
        cv.notify_all();
      }
    } catch (...) {
      std::lock_guard<std::mutex> lk(mu);
      if (!error) error = std::current_exception();
      cv.notify_all();
    }
  });

  auto release_ready = [&](bool wait) {
    std::unique_lock<std::mutex> lk(mu);
    if (wait) cv.wait(lk, [&] { return error || !to_release.empty(); });
    if (error) std::rethrow_exception(error);
    while (!to_release.empty()) {
      const int tok = to_release.front();
      to_release.pop_front();
      const BatchRef ref = held[static_cast<size_t>(tok)].ref;
      lk.unlock();
      scan->ReleaseBatch(ref);
      lk.lock();
    }
  };
  auto queue_job = [&](WriteJob&& job) {
    {
      std::lock_guard<std::mutex> lk(mu);
      jobs.push_back(std::move(job));
      ++jobs_queued;
    }
    cv.notify_all();
  };
  // valid string bytes of rows [r0, r0 + m): the size of the Arrow data buffer the encoder will fill
  auto payload_of = [](const DeviceColumnView& v, int64_t r0, int64_t m) -> int64_t {
    auto off = [&](int64_t i) -> int64_t {
      if (v.offset_width == 8) { int64_t x; std::memcpy(&x, v.h_offsets + i * 8, 8); return x; }
      int32_t x; std::memcpy(&x, v.h_offsets + i * 4, 4); return x;
    };
    if (v.null_count == 0 || !v.h_validity) return off(r0 + m) - off(r0);
    int64_t total = 0;
    for (int64_t i = r0; i < r0 + m; i++)
      if ((v.h_validity[i >> 3] >> (i & 7)) & 1) total += off(i + 1) - off(i);
    return total;
  };
  // can rows of this batch be encoded where they lie?
  auto views_of = [&](const BatchRef& ref, std::vector<DeviceColumnView>* views) -> bool {
    views->resize(n_cols);
    for (size_t c = 0; c < n_cols; c++) {
      DeviceColumnView& v = (*views)[c];
      scan->DeviceColumn(ref, c, &v);
      if (!v.flat) return false;
      const auto& wc = w->buffer->columns[static_cast<size_t>(w->buffer->roots[c])];
      const bool is_string = v.kind == MI_K_STR32 || v.kind == MI_K_STR64;
      if (wc.enc_kind == MI_K_ENC_STR32) {
        if (!is_string) return false;
      } else if (is_string || v.width != wc.width || v.kind == MI_K_STRVIEW || v.kind == MI_K_FIXED_BINARY) {
        return false;
      }
    }
    return true;
  };
  auto encode_on_gpu = [&](int tok, const std::vector<DeviceColumnView>& views, int64_t r0, int64_t m) {
    // a free encoder (at most kEncoders row groups between the kernels and the file)
    int ei = -1;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] {
        if (error) return true;
        for (int i = 0; i < kEncoders; i++)
          if (!enc[static_cast<size_t>(i)].busy) { ei = i; return true; }
        return false;
      });
      if (error) std::rethrow_exception(error);
      enc[static_cast<size_t>(ei)].busy = true;
      held[static_cast<size_t>(tok)].open++;
    }
    FusedEncoder& e = enc[static_cast<size_t>(ei)];
    e.nrows = m;
    e.spans.clear();
    e.first_span.assign(n_cols, 0);
    size_t body_off = 0;
    auto add_span = [&](int64_t len) {
      e.spans.push_back(mi_buffer_span{static_cast<int64_t>(body_off), len});
      body_off += RoundUp(static_cast<size_t>(len), kBufferAlign);
    };
    std::vector<int64_t> payload(n_cols, 0);
    for (size_t c = 0; c < n_cols; c++) {
      const auto& wc = w->buffer->columns[static_cast<size_t>(w->buffer->roots[c])];
      e.first_span[c] = static_cast<int32_t>(e.spans.size());
      add_span((m + 7) / 8);
      switch (wc.enc_kind) {
        case MI_K_ENC_COPY: add_span(m * wc.param); break;
        case MI_K_ENC_DEC128: add_span(m * 16); break;
        case MI_K_ENC_BOOL: add_span((m + 7) / 8); break;
        default: {
          payload[c] = payload_of(views[c], r0, m);
          if (payload[c] > 0x7FFFFFFFll && !wc.large_offsets)
            throw InvalidInputException(
                "Arrow Appender: The maximum total string size for regular string buffers is 2147483647 but the offset of " +
                std::to_string(payload[c]) + " exceeds this.\n* SET arrow_large_buffer_size=true to use large string buffers");
          add_span((m + 1) * (wc.large_offsets ? 8 : 4));
          add_span(payload[c]);
        }
      }
    }
    e.body_size = static_cast<int64_t>(body_off);
    GrowDevice(&e.d_body, &e.d_cap, body_off + 256);
    { size_t zero = 0; GrowPinned(&e.h_body, &e.h_cap, body_off + 256, zero); }
    MI_HIP_CHECK(hipMemsetAsync(e.d_body, 0, body_off, enc_stream));
    std::vector<mi_col_task> tasks(n_cols);
    for (size_t c = 0; c < n_cols; c++) {
      const auto& wc = w->buffer->columns[static_cast<size_t>(w->buffer->roots[c])];
      const DeviceColumnView& v = views[c];
      const size_t sp = static_cast<size_t>(e.first_span[c]);
      mi_col_task& t = tasks[c];
      std::memset(&t, 0, sizeof(t));
      t.nrows = m;
      t.kind = wc.enc_kind;
      t.param = wc.param;
      t.flags = wc.large_offsets ? 1 : 0;
      t.buf1 = v.d_data + static_cast<size_t>(r0) * static_cast<size_t>(v.width);
      t.validity = v.d_validity ? v.d_validity + static_cast<size_t>(r0 / 64) * 8 : nullptr;   // r0 is a multiple of 2048
      t.out_validity = e.d_body + e.spans[sp].offset;
      t.out_data = e.d_body + e.spans[sp + 1].offset;
      if (wc.enc_kind == MI_K_ENC_STR32) {
        t.buf2 = v.d_heap;
        t.buf2_len = payload[c];
        t.ptr_base = v.ptr_base;
        t.out_aux = e.d_body + e.spans[sp + 2].offset;
      }
    }
    e.plan->Set(tasks.data(), static_cast<int32_t>(tasks.size()), enc_stream);
    MI_HIP_CHECK(hipMemsetAsync(e.plan->d_status, 0, sizeof(uint32_t), enc_stream));
    if (e.plan->n_null_counts) MI_HIP_CHECK(hipMemsetAsync(e.plan->d_null_counts, 0, static_cast<size_t>(e.plan->n_null_counts) * 8, enc_stream));
    e.plan->Launch(enc_stream);
    MI_HIP_CHECK(hipEventRecord(e.encoded, enc_stream));
    MI_HIP_CHECK(hipStreamWaitEvent(back_stream, e.encoded, 0));
    MI_HIP_CHECK(hipMemcpyAsync(e.h_body, e.d_body, body_off, hipMemcpyDeviceToHost, back_stream));
    { size_t zero = 0; GrowPinned(&e.h_nulls, &e.h_nulls_cap, static_cast<size_t>(e.plan->n_null_counts + 1) * 8, zero); }
    if (e.plan->n_null_counts)
      MI_HIP_CHECK(hipMemcpyAsync(e.h_nulls, e.plan->d_null_counts, static_cast<size_t>(e.plan->n_null_counts) * 8, hipMemcpyDeviceToHost, back_stream));
    MI_HIP_CHECK(hipMemcpyAsync(e.h_status, e.plan->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, back_stream));
    MI_HIP_CHECK(hipEventRecord(e.done, back_stream));
    WriteJob job;
    job.enc = ei;
    job.tok = tok;
    queue_job(std::move(job));
  };
  // the host path: serialise the rows buffered in `local` and hand them to the I/O thread; its body buffer is reused by
  // the next host-path row group, so wait until it is written (row groups that straddle batches are the exception)
  auto flush_host = [&] {
    if (local->serializer->Serialize(*local->buffer) == 0) {
      local->buffer->Reset();
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return error || jobs_written == jobs_queued; });
      out.CountEmptyFlush();
      return;
    }
    local->buffer->Reset();
    WriteJob job;
    job.header = local->serializer->GetHeader();
    job.body = local->serializer->GetBody();
    job.body_size = static_cast<size_t>(local->serializer->GetBodySize());
    queue_job(std::move(job));
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return error || jobs_written == jobs_queued; });
    if (error) std::rethrow_exception(error);
  };

  int64_t rows = 0;
  std::exception_ptr pump_error;
  try {
    const int64_t group_rows = (rows_per_group + MI_VECTOR_SIZE - 1) / MI_VECTOR_SIZE * MI_VECTOR_SIZE;
    int64_t carry = 0;                      // rows buffered on the host path
    bool have_first = true;
    std::vector<DeviceColumnView> views;
    while (true) {
      release_ready(false);
      BatchRef ref;
      if (have_first) {
        ref = first;
        have_first = false;
      } else if (!scan->AcquireBatch(&ref)) {
        if (scan->Exhausted()) break;
        release_ready(true);
        continue;
      }
      rows += ref.chunk_rows;
      int tok;
      {
        std::lock_guard<std::mutex> lk(mu);
        held.push_back(Held{ref, 0, false, false});
        tok = static_cast<int>(held.size() - 1);
      }
      const int64_t n = ref.chunk_rows;
      const bool on_gpu = n > 0 && views_of(ref, &views);
      int64_t r = 0;
      while (r < n) {
        if (carry == 0 && on_gpu && n - r >= rows_per_group) {
          const int64_t m = std::min(group_rows, n - r);
          encode_on_gpu(tok, views, r, m);
          r += m;
          continue;
        }
        // host path, chunk by chunk until the row group is full or the batch ends
        scan->EnsureHostVectors(ref);
        const int32_t wi = static_cast<int32_t>(r / MI_VECTOR_SIZE);
        scan->BuildChunk(ref, wi, &storage, &chunk);
        local->buffer->Append(chunk);
        carry += chunk.size;
        r += chunk.size;
        if (carry >= rows_per_group) {
          flush_host();
          carry = 0;
        }
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        Held& h = held[static_cast<size_t>(tok)];
        h.fully_cut = true;
        if (h.open == 0 && !h.released) {
          h.released = true;
          to_release.push_back(tok);
        }
      }
    }
    if (carry > 0) flush_host();   // the tail row group (ArrowWriteCombine)
  } catch (...) {
    pump_error = std::current_exception();
    std::lock_guard<std::mutex> lk(mu);
    if (!error) error = pump_error;
  }
  {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return error || jobs_written == jobs_queued; });
    stop = true;
  }
  cv.notify_all();
  io.join();
  (void)hipStreamSynchronize(enc_stream);   // the GPU is done with every scan slot before the batches go back
  (void)hipStreamSynchronize(back_stream);
  for (size_t tok = 0; tok < held.size(); tok++)
    if (!held[tok].released || std::find(to_release.begin(), to_release.end(), static_cast<int>(tok)) != to_release.end()) scan->ReleaseBatch(held[tok].ref);
  if (error) std::rethrow_exception(error);
  if (rows_out) *rows_out = rows;
}
}  // namespace

extern "C" {

int mi_writer_local_create(mi_writer* w, mi_writer_local** out) {
  return WrapC([&] {
    if (!w || !w->writer || !out) throw InvalidInputException("mi_writer_local_create: bad argument");
    *out = MakeLocal(w).release();
  });
}

int mi_writer_local_sink(mi_writer_local* l, const mi_data_chunk* chunk) {
  return WrapC([&] {
    if (!l || !chunk) throw InvalidInputException("mi_writer_local_sink: bad argument");
    l->buffer->Append(*chunk);
    if (l->buffer->Count() >= l->w->opts.row_group_size || l->buffer->SizeInBytes() >= l->w->opts.row_group_size_bytes) l->FlushRowGroup();
  });
}

int mi_writer_local_combine(mi_writer_local* l) {
  return WrapC([&] {
    if (!l) throw InvalidInputException("mi_writer_local_combine: NULL");
    if (l->buffer->Count() > 0) l->FlushRowGroup();
  });
}

void mi_writer_local_destroy(mi_writer_local* l) { delete l; }

int mi_writer_sink_scan(mi_writer* w, mi_scan* scan, int64_t* rows) {
  if (!w || !w->writer || !scan) return WrapC([] { throw InvalidInputException("mi_writer_sink_scan: bad argument"); });
  ArrowScan* single = SingleScanOf(scan);
  const int threads = SinkThreads();
  if (single && single->HostConsumer() && w->buffer->Count() == 0 && !single->Initialized()) single->Init({});
  if (single && single->HostConsumer() && w->buffer->Count() == 0 && (threads > 1 || FusedSinkPossible(w, single))) {
    return WrapC([&] {
      // rows after which the one-thread sink flushes (row_group_size, or row_group_size_bytes at the staged row width)
      int64_t row_bytes = 0;
      for (auto& c : single->OutputColumns()) {
        int32_t kind, wd, nb;
        int64_t param;
        row_bytes += (!c.is_filename && !c.is_hive && c.field.Plan(&kind, &param, &wd, &nb)) ? wd : 16;
      }
      row_bytes = std::max<int64_t>(1, row_bytes);
      const int64_t group = std::max<int64_t>(1, std::min(w->opts.row_group_size, (w->opts.row_group_size_bytes + row_bytes - 1) / row_bytes));
      // record batches at least one row group long are encoded where they lie in HBM; smaller ones go through the sink
      // threads.  Decided on the first batch (the scan keeps its vectors on the device until then).
      const bool fused = FusedSinkPossible(w, single);
      single->EnsurePipelineDepth(std::max(1, threads) + 4);
      // whatever happens below, the scan hands out host vectors again afterwards
      struct Restore { ArrowScan* s; ~Restore() { s->KeepVectorsOnDevice(false); } } restore{single};
      single->KeepVectorsOnDevice(fused);
      BatchRef first;
      if (!single->AcquireBatch(&first)) {
        if (rows) *rows = 0;
        return;
      }
      if (fused && first.chunk_rows >= group) {
        PumpScanFused(w, single, first, group, rows);
        return;
      }
      single->KeepVectorsOnDevice(false);
      PumpScanParallel(w, single, std::max(1, threads), rows, &first);
    });
  }
  int64_t n = 0;
  mi_data_chunk ch;
  while (true) {
    int rc = mi_scan_next(scan, &ch);
    if (rc != MI_OK) return rc;
    if (ch.size == 0) break;
    rc = mi_writer_sink(w, &ch);
    if (rc != MI_OK) return rc;
    n += ch.size;
  }
  if (rows) *rows = n;
  return MI_OK;
}

int mi_writer_finalize(mi_writer* w) {
  return WrapC([&] {
    if (!w || !w->writer) throw InvalidInputException("mi_writer_finalize: bad argument");
    if (w->buffer->Count() > 0) w->writer->Flush(*w->buffer);  // ArrowWriteCombine
    w->writer->Finalize();                                     // ArrowWriteFinalize
  });
}

void mi_writer_close(mi_writer* w) { delete w; }

int64_t mi_writer_row_groups(const mi_writer* w) { return (w && w->writer) ? static_cast<int64_t>(w->writer->NumberOfRowGroups()) : 0; }
int64_t mi_writer_file_size(const mi_writer* w) { return (w && w->writer) ? static_cast<int64_t>(w->writer->FileSize()) : 0; }

int mi_writer_rotate_next_file(const mi_writer* w, int64_t file_size_bytes) {
  if (!w || !w->writer) return 0;
  if (file_size_bytes >= 0 && static_cast<int64_t>(w->writer->FileSize()) > file_size_bytes) return 1;
  if (w->opts.row_groups_per_file > 0 && static_cast<int64_t>(w->writer->NumberOfRowGroups()) >= w->opts.row_groups_per_file) return 1;
  return 0;
}

int mi_ipc_serializer_create(mi_ctx* ctx, const mi_field* fields, int32_t n_fields, mi_writer** out) {
  return WrapC([&] {
    if (!ctx || !out) throw InvalidInputException("mi_ipc_serializer_create: NULL argument");
    auto w = std::make_unique<mi_writer>();
    w->ctx = ContextOf(ctx);
    mi_write_options_init(&w->opts);
    w->fields = FieldsFromC(fields, n_fields);
    w->schema.fields = w->fields;
    w->buffer = std::make_unique<ChunkCollection>(w->ctx, w->fields);
    w->serializer = std::make_unique<ColumnDataCollectionSerializer>(w->ctx);
    w->serializer->Init(&w->schema);
    *out = w.release();
  });
}

int mi_ipc_serialize_schema(mi_writer* w, const uint8_t** blob, int64_t* size) {
  return WrapC([&] {
    if (!w || !w->serializer || !blob || !size) throw InvalidInputException("mi_ipc_serialize_schema: bad argument");
    w->serializer->SerializeSchema();
    w->blob = w->serializer->GetHeader();
    *blob = w->blob.data();
    *size = static_cast<int64_t>(w->blob.size());
  });
}

int mi_writer_append_message(mi_writer* w, const uint8_t* blob, int64_t size) {
  return WrapC([&] {
    if (!w || !w->writer || (!blob && size) || size < 0) throw InvalidInputException("mi_writer_append_message: bad argument");
    if (size == 0) {   // an empty collection still counts as a flushed row group (ArrowStreamWriter::Flush)
      w->writer->CountEmptyFlush();
      return;
    }
    const int64_t at = w->writer->ReserveRowGroup(static_cast<size_t>(size));
    w->writer->WriteAt(at, blob, static_cast<size_t>(size));
  });
}

int mi_ipc_serialize_chunks(mi_writer* w, const mi_data_chunk* chunks, int32_t n_chunks, const uint8_t** blob, int64_t* size) {
  return WrapC([&] {
    if (!w || !w->serializer || !blob || !size || (!chunks && n_chunks)) throw InvalidInputException("mi_ipc_serialize_chunks: bad argument");
    w->buffer->Reset();
    for (int32_t i = 0; i < n_chunks; i++) w->buffer->Append(chunks[i]);
    w->blob.clear();
    if (w->serializer->Serialize(*w->buffer)) {
      // header || body concatenated, like SerializeArray (to_arrow_ipc.cpp:72-87)
      const auto& h = w->serializer->GetHeader();
      w->blob.resize(h.size() + static_cast<size_t>(w->serializer->GetBodySize()));
      std::memcpy(w->blob.data(), h.data(), h.size());
      std::memcpy(w->blob.data() + h.size(), w->serializer->GetBody(), static_cast<size_t>(w->serializer->GetBodySize()));
    }
    w->buffer->Reset();
    *blob = w->blob.data();
    *size = static_cast<int64_t>(w->blob.size());
  });
}

}  // extern "C"
