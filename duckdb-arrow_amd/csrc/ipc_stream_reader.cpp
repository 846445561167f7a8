// ipc_stream_reader.cpp -- see ipc_stream_reader.hpp.
#include "ipc_stream_reader.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <unordered_map>
#include <unordered_set>

namespace miarrow {

static std::string Lower(std::string s) {
  for (auto& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
  return s;
}

// Same observable behaviour as DuckDB's QueryResult::DeduplicateColumns (used at base_stream_reader.cpp:177 and
// arrow_file_scan.cpp:19): case-insensitive; a repeated name gets "_<n>" appended, n counting repetitions and
// skipping suffixes that are already taken.
void DeduplicateColumns(std::vector<std::string>& names) {
  std::unordered_map<std::string, idx_t> seen;
  for (auto& name : names) {
    std::string low = Lower(name);
    auto it = seen.find(low);
    if (it == seen.end()) {
      seen[low] = 1;
      continue;
    }
    std::string candidate = name + "_" + std::to_string(seen[low]);
    while (seen.find(Lower(candidate)) != seen.end()) {
      seen[low]++;
      candidate = name + "_" + std::to_string(seen[low]);
    }
    name = candidate;
    seen[Lower(candidate)] = 1;
  }
}

// ------------------------------------------------------------------------------------------------ helpers
namespace {
struct SerializationException : std::runtime_error {
  SerializationException() : std::runtime_error("not enough data in file to deserialize result") {}
};

std::shared_ptr<void> DefaultBodyAlloc(size_t bytes, MessageType, uint8_t** ptr) {
  void* p = nullptr;
  if (posix_memalign(&p, 256, bytes ? bytes : 8) != 0 || p == nullptr) throw std::bad_alloc();
  *ptr = static_cast<uint8_t*>(p);
  return std::shared_ptr<void>(p, [](void* q) { std::free(q); });
}
}  // namespace

// ------------------------------------------------------------------------------------------------ base reader
const ArrowSchemaModel& IPCStreamReader::GetBaseSchema() {
  if (have_base_schema) return base_schema;
  ReadNextMessage({MessageType::SCHEMA}, /*end_of_stream_ok*/ false);
  base_schema = DecodeSchema(message_meta, message_meta_len);
  if (base_schema.features & (1u << 1)) {
    throw IOException("This stream uses unsupported feature DICTIONARY_REPLACEMENT");
  }
  // Big-endian streams (Schema.endianness = Big): metadata and message prefixes are little-endian whatever the producer,
  // only the buffers hold big-endian values.  The reference hands this to nanoarrow, which swaps while it decodes
  // (ArrowIpcDecoderSetEndianness, base_stream_reader.cpp:68-69); here the body of every message is swapped in place right
  // after it is read (SwapBodyEndianness), so that everything downstream -- kernels, C stream export, planner -- sees the
  // little-endian layout it expects.
  if (base_schema.endianness != 0 && base_schema.endianness != 1)
    throw IOException("Unknown Schema.endianness " + std::to_string(base_schema.endianness));
  have_base_schema = true;
  return base_schema;
}

const ArrowSchemaModel& IPCStreamReader::GetOutputSchema() {
  if (HasProjection()) return projected_schema;
  return GetBaseSchema();
}

// Projection pushdown by column name (the seam of base_stream_reader.cpp:146-212; error texts are the reference's).
// Names are the deduplicated top-level names; a name that two columns still share after deduplication cannot be
// addressed.  Per projected column the reader keeps its top-level index and its depth-first flattened field index
// (children count as fields of their own), which is how the decoder of the reference addresses columns.
void IPCStreamReader::SetColumnProjection(const std::vector<std::string>& column_names) {
  if (column_names.empty()) throw InternalException("Can't request zero fields projected from IpcStreamReader");
  GetBaseSchema();
  const size_t n_top = base_schema.fields.size();
  std::vector<std::string> names(n_top);
  std::vector<int64_t> flat_start(n_top + 1, 0);  // flattened index of every top-level field = fields before it
  for (size_t i = 0; i < n_top; i++) {
    names[i] = base_schema.fields[i].name;
    flat_start[i + 1] = flat_start[i] + CountFields(base_schema.fields[i]);
  }
  DeduplicateColumns(names);
  auto locate = [&](const std::string& wanted) -> size_t {
    size_t hit = n_top, hits = 0;
    for (size_t i = 0; i < n_top; i++) {
      if (names[i] != wanted) continue;
      if (hits++ == 0) hit = i;
    }
    if (hits > 1) throw InternalException("Field '" + wanted + "' refers to a duplicate column name in IPC file schema");
    if (hits == 0) throw InternalException("Field '" + wanted + "' does not exist in IPC file schema");
    return hit;
  };
  ArrowSchemaModel picked;
  picked.endianness = base_schema.endianness;
  picked.metadata = base_schema.metadata;
  std::vector<int64_t> flat;
  std::vector<int32_t> top;
  for (const std::string& wanted : column_names) {
    const size_t i = locate(wanted);
    top.push_back(static_cast<int32_t>(i));
    flat.push_back(flat_start[i]);
    picked.fields.push_back(base_schema.fields[i]);
  }
  // nothing changes unless every name resolved
  projected_columns.swap(top);
  projected_fields.swap(flat);
  projected_schema = std::move(picked);
}

// The prefix has been read into message_prefix: header (flatbuffer) next, then the body.  UNINITIALIZED = the
// end-of-stream marker.  (The reference splits this into DecodeMetadata + DecodeMessage, base_stream_reader.cpp:214-236;
// its BSWAP of the length only happens on big-endian hosts, which gfx950 hosts are not.)
MessageType IPCStreamReader::FinishMessage() {
  const int64_t metadata_size = message_prefix.metadata_size;
  if (metadata_size < 0) throw IOException("Expected metadata size >= 0 but got " + std::to_string(metadata_size));
  const bool end_of_stream = DecodeHeader(static_cast<idx_t>(metadata_size) + sizeof(message_prefix));
  if (end_of_stream) return MessageType::UNINITIALIZED;
  DecodeBody();
  return message.type;
}

bool IPCStreamReader::ParseHeader(const uint8_t* header_with_prefix, idx_t size) {
  // ArrowIpcDecoderDecodeHeader: metadata size 0 is the end-of-stream marker => ENODATA
  if (message_prefix.metadata_size == 0) return false;
  message_meta = header_with_prefix + sizeof(message_prefix);
  message_meta_len = static_cast<int64_t>(size - sizeof(message_prefix));
  message = DecodeMessageHeader(message_meta, message_meta_len);
  return true;
}

MessageType IPCStreamReader::ReadNextMessage(std::vector<MessageType> expected_types, bool end_of_stream_ok) {
  const MessageType got = ReadNextMessage();
  const bool at_end = got == MessageType::UNINITIALIZED;
  if (at_end && end_of_stream_ok) return got;
  if (std::find(expected_types.begin(), expected_types.end(), got) != expected_types.end()) return got;
  std::string wanted;
  for (MessageType t : expected_types) {
    if (!wanted.empty()) wanted += " or ";
    wanted += MessageTypeString(t);
  }
  throw IOException("Expected " + wanted + " Arrow IPC message but got " + (at_end ? "end of stream" : MessageTypeString(got)));
}

bool IPCStreamReader::GetNextBatch(DecodedBatch* out, bool accept_dictionaries, bool skip_body) {
  GetBaseSchema();
  std::vector<MessageType> expected = {MessageType::RECORD_BATCH};
  if (accept_dictionaries) expected.push_back(MessageType::DICTIONARY_BATCH);
  skip_record_batch_body = skip_body;
  MessageType message_type;
  try {
    message_type = ReadNextMessage(expected);
  } catch (...) {
    skip_record_batch_body = false;
    throw;
  }
  skip_record_batch_body = false;
  if (message_type == MessageType::UNINITIALIZED) return false;
  RecordBatchMeta meta = DecodeRecordBatch(message_meta, message_meta_len);
  if (skip_body && message_type == MessageType::RECORD_BATCH) {
    *out = DecodedBatch();
    out->length = meta.length;
    out->body_file_offset = cur_body_offset;
    out->compression = meta.compression;
    return true;
  }
  cur_deferred.reset();
  if (meta.compression != -1 && cur_size > 0) DecompressBody(&meta);
  if (base_schema.endianness == 1 && cur_size > 0) SwapBodyEndianness(meta);
  SliceBatch(meta, out);
  if (cur_deferred) {
    out->deferred = cur_deferred;
    out->body = nullptr;
    out->compression = 0;
    cur_deferred.reset();
  }
  return true;
}

// Large bodies are read with several concurrent pread()s: one thread copies out of the page cache at ~10 GB/s, far below
// what the H2D link takes, so the body is cut into slices read in parallel.  The pool is process wide (MI_IO_THREADS,
// default 8, grown by multi-device scans to 8 per device) and serves any number of callers at once: a Run() is a batch of
// tasks in one shared queue, the caller works on its own batch while it waits.
namespace {
struct IoAffinity {
  cpu_set_t cpus;
  int node = -1;
};
thread_local const IoAffinity* tls_io_affinity = nullptr;   // what this thread is bound to; its pool jobs ask the same of the workers
constexpr int kMpolDefault = 0, kMpolPreferred = 1;          // <linux/mempolicy.h>
void SetPreferredNode(int node) {
  if (node < 0) {
    (void)syscall(SYS_set_mempolicy, kMpolDefault, nullptr, 0);
    return;
  }
  unsigned long mask[16] = {0};
  if (node >= static_cast<int>(sizeof(mask) * 8)) return;
  mask[static_cast<size_t>(node) / (8 * sizeof(unsigned long))] |= 1ul << (static_cast<size_t>(node) % (8 * sizeof(unsigned long)));
  (void)syscall(SYS_set_mempolicy, kMpolPreferred, mask, sizeof(mask) * 8);
}
void ApplyIoAffinity(const IoAffinity* a) {
  cpu_set_t allowed, want;
  CPU_ZERO(&allowed);
  CPU_ZERO(&want);
  // a thread that was narrowed to another node before may widen again: ask for the process's CPUs first
  if (sched_getaffinity(getpid(), sizeof(allowed), &allowed) != 0) return;
  int n = 0;
  for (int c = 0; c < CPU_SETSIZE; c++)
    if (CPU_ISSET(c, &a->cpus) && CPU_ISSET(c, &allowed)) {
      CPU_SET(c, &want);
      n++;
    }
  if (n == 0) return;   // the process may not run on that node at all: stay
  (void)sched_setaffinity(0, sizeof(want), &want);
  SetPreferredNode(a->node);
  tls_io_affinity = a;
}
const IoAffinity* IoAffinityOf(int node, const std::vector<int>& cpus) {
  static std::mutex mu;
  static std::map<int, std::unique_ptr<IoAffinity>> by_node;   // a node's CPUs do not change: one object per node, never freed
  std::lock_guard<std::mutex> lk(mu);
  auto& slot = by_node[node];
  if (!slot) {
    slot = std::make_unique<IoAffinity>();
    CPU_ZERO(&slot->cpus);
    for (int c : cpus)
      if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &slot->cpus);
    slot->node = node;
  }
  return slot.get();
}

class IoPool {
 public:
  static IoPool& Get() {
    static IoPool pool;
    return pool;
  }
  int Threads() {
    std::lock_guard<std::mutex> lk(mu);
    return n_threads;
  }
  // CPUs the process may really use: the hardware's, or the cgroup's CPU quota when there is one (a container sees all 256
  // CPUs of the box and gets 16 CPUs' worth of time: threads beyond the quota only throttle one another -- with 12 and 16
  // I/O threads the SF10 host-consumer scan took 0.204 s, with 8 0.18 s)
  static int CpuBudget() {
    int hw = std::max(1, static_cast<int>(std::thread::hardware_concurrency()));
    long long quota = -1, period = 0;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota|max> <period>"
      char q[32] = {0};
      if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atoll(q);
      std::fclose(f);
    } else if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
      if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
      std::fclose(g);
      if (FILE* h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (std::fscanf(h, "%lld", &period) != 1) period = 0;
        std::fclose(h);
      }
    }
    if (quota > 0 && period > 0) hw = std::min<long long>(hw, std::max<long long>(1, quota / period));
    return hw;
  }
  void Ensure(int n) {
    std::lock_guard<std::mutex> lk(mu);
    const int cap = CpuBudget();
    n = std::min(n, std::max(cap / 2, 8));   // half of the budget: the pipeline threads, the HIP runtime's and the caller's need the rest
    while (n_threads < n) {
      workers.emplace_back([this] { Loop(); });
      n_threads++;
    }
  }
  // runs fn(i) for i in [0, n) on the pool + the calling thread; rethrows the first failure
  void Run(int n, const std::function<void(int)>& fn) {
    if (n <= 1 || Threads() <= 1) {
      for (int i = 0; i < n; i++) fn(i);
      return;
    }
    Job job;
    job.fn = &fn;
    job.n = n;
    job.pending = n;
    job.affinity = tls_io_affinity;   // the caller's binding, if it has one
    {
      std::lock_guard<std::mutex> lk(mu);
      jobs.push_back(&job);
    }
    cv.notify_all();
    Work(&job);  // the caller takes tasks of its own batch
    std::unique_lock<std::mutex> lk(mu);
    job.done_cv.wait(lk, [&] { return job.pending == 0; });
    if (job.error) std::rethrow_exception(job.error);
  }

 private:
  struct Job {
    const std::function<void(int)>* fn = nullptr;
    const IoAffinity* affinity = nullptr;
    int n = 0, next = 0, pending = 0;
    std::exception_ptr error;
    std::condition_variable done_cv;
  };
  IoPool() {
    const char* v = std::getenv("MI_IO_THREADS");
    const int n = v ? std::max(1, std::atoi(v)) : 8;
    n_threads = 1;  // the calling thread
    for (int i = 1; i < n; i++) {
      workers.emplace_back([this] { Loop(); });
      n_threads++;
    }
  }
  ~IoPool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : workers) t.join();
  }
  // takes tasks of `only` (or of the oldest batch with tasks left when NULL) until none is left
  void Work(Job* only) {
    while (true) {
      Job* job = nullptr;
      int i = 0;
      {
        std::lock_guard<std::mutex> lk(mu);
        if (only) {
          if (only->next < only->n) job = only;
        } else {
          for (Job* j : jobs)
            if (j->next < j->n) { job = j; break; }
        }
        if (!job) return;
        i = job->next++;
        if (job->next >= job->n) jobs.erase(std::find(jobs.begin(), jobs.end(), job));  // nothing left to hand out
      }
      if (job->affinity && job->affinity != tls_io_affinity) ApplyIoAffinity(job->affinity);   // ~2 us, once per change of caller
      std::exception_ptr err;
      try {
        (*job->fn)(i);
      } catch (...) {
        err = std::current_exception();
      }
      std::lock_guard<std::mutex> lk(mu);
      if (err && !job->error) job->error = err;
      if (--job->pending == 0) job->done_cv.notify_all();
    }
  }
  void Loop() {
    while (true) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !jobs.empty(); });
        if (stop) return;
      }
      Work(nullptr);
    }
  }
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::thread> workers;
  std::deque<Job*> jobs;  // batches that still have tasks to hand out
  int n_threads = 1;
  bool stop = false;
};
}  // namespace

void EnsureIoThreads(int n) { IoPool::Get().Ensure(n); }

void BindThisThreadToNode(int node, const std::vector<int>& cpus) {
  if (node < 0 || cpus.empty()) return;
  ApplyIoAffinity(IoAffinityOf(node, cpus));
}
void PreferNode(int node) { SetPreferredNode(node); }

// ------------------------------------------------------------------------------------------------ compression
// Body compression (Message.fbs BodyCompression, method BUFFER): every buffer is `int64 uncompressed_length` (-1 = the
// bytes that follow are stored raw) + one frame.  The reference decompresses ZSTD on the CPU with DuckDB's bundled zstd
// (DuckDBDecompressZstd, base_stream_reader.cpp:11-32) and registers no LZ4 function (:37-50); here the system's
// libzstd.so.1 is bound at run time (no headers in the image), LZ4_FRAME stays unsupported like in the reference.
namespace {
struct ZstdApi {
  size_t (*decompress)(void*, size_t, const void*, size_t) = nullptr;
  unsigned (*is_error)(size_t) = nullptr;
  const char* (*error_name)(size_t) = nullptr;
  unsigned long long (*frame_content_size)(const void*, size_t) = nullptr;  // optional
  bool ok = false;
};
const ZstdApi& Zstd() {
  static ZstdApi api = [] {
    ZstdApi a;
    void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
    if (h) {
      a.decompress = reinterpret_cast<size_t (*)(void*, size_t, const void*, size_t)>(dlsym(h, "ZSTD_decompress"));
      a.is_error = reinterpret_cast<unsigned (*)(size_t)>(dlsym(h, "ZSTD_isError"));
      a.error_name = reinterpret_cast<const char* (*)(size_t)>(dlsym(h, "ZSTD_getErrorName"));
      a.frame_content_size = reinterpret_cast<unsigned long long (*)(const void*, size_t)>(dlsym(h, "ZSTD_getFrameContentSize"));
      a.ok = a.decompress && a.is_error && a.error_name;
    }
    return a;
  }();
  return api;
}
// LZ4_FRAME (codec 0; what Feather V2 files use by default): the reference registers no LZ4 function, so it rejects these
// bodies; here the system's liblz4.so.1 frame API is bound at run time when it exists.
struct Lz4Api {
  size_t (*create)(void**, unsigned) = nullptr;
  size_t (*free_ctx)(void*) = nullptr;
  size_t (*decompress)(void*, void*, size_t*, const void*, size_t*, const void*) = nullptr;
  unsigned (*is_error)(size_t) = nullptr;
  const char* (*error_name)(size_t) = nullptr;
  bool ok = false;
};
const Lz4Api& Lz4() {
  static Lz4Api api = [] {
    Lz4Api a;
    void* h = dlopen("liblz4.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("liblz4.so", RTLD_NOW | RTLD_LOCAL);
    if (h) {
      a.create = reinterpret_cast<size_t (*)(void**, unsigned)>(dlsym(h, "LZ4F_createDecompressionContext"));
      a.free_ctx = reinterpret_cast<size_t (*)(void*)>(dlsym(h, "LZ4F_freeDecompressionContext"));
      a.decompress = reinterpret_cast<size_t (*)(void*, void*, size_t*, const void*, size_t*, const void*)>(dlsym(h, "LZ4F_decompress"));
      a.is_error = reinterpret_cast<unsigned (*)(size_t)>(dlsym(h, "LZ4F_isError"));
      a.error_name = reinterpret_cast<const char* (*)(size_t)>(dlsym(h, "LZ4F_getErrorName"));
      a.ok = a.create && a.free_ctx && a.decompress && a.is_error && a.error_name;
    }
    return a;
  }();
  return api;
}

// one LZ4 frame -> exactly n bytes
void Lz4DecompressFrame(const Lz4Api& z, uint8_t* dst, int64_t n, const uint8_t* src, int64_t src_len) {
  void* dctx = nullptr;
  size_t rc = z.create(&dctx, 100 /* LZ4F_VERSION */);
  if (z.is_error(rc)) throw IOException(std::string("LZ4F_createDecompressionContext failed: ") + z.error_name(rc));
  std::shared_ptr<void> guard(dctx, [&z](void* p) { z.free_ctx(p); });
  size_t produced = 0, consumed = 0;
  while (true) {
    size_t dst_size = static_cast<size_t>(n) - produced, src_size = static_cast<size_t>(src_len) - consumed;
    rc = z.decompress(dctx, dst + produced, &dst_size, src + consumed, &src_size, nullptr);
    if (z.is_error(rc)) {
      throw IOException("LZ4F_decompress([buffer with " + std::to_string(src_len) + " bytes] -> [buffer with " + std::to_string(n) +
                        " bytes]) failed with error '" + z.error_name(rc) + "'");
    }
    produced += dst_size;
    consumed += src_size;
    if (rc == 0) break;                             // frame complete
    if (dst_size == 0 && src_size == 0) break;      // no progress: truncated frame or output full
  }
  if (static_cast<int64_t>(produced) != n || rc != 0)
    throw IOException("Expected decompressed size of " + std::to_string(n) + " bytes but got " + std::to_string(produced) + " bytes");
}
}  // namespace

static void SubtreeBufferBounds(const ArrowField& f, const RecordBatchMeta& meta, size_t* node, size_t* variadic, bool value_only,
                                std::vector<int64_t>* out);

// list / map columns: the planner samples their offsets on the host (child windows of every chunk), so their record batches
// need the decompressed body in host memory
static bool HasListField(const ArrowField& f) {
  if (f.type == MI_AT_LIST || f.type == MI_AT_LARGE_LIST || f.type == MI_AT_MAP) return true;
  for (auto& c : f.children)
    if (HasListField(c)) return true;
  return false;
}

// Walks one LZ4 frame (lz4_Frame_format.md: magic, FLG, BD, [content size], [dict id], HC, blocks, end mark) without
// touching the block data: appends its blocks to `blocks`.  false = something the GPU path does not take (skippable or
// legacy frames, a dictionary id, a damaged header): the caller decompresses the record batch on the host instead, which
// also produces the reference's error text for damaged input.
static bool WalkLz4Frame(const uint8_t* body, int64_t frame_off, int64_t frame_len, uint32_t buffer_index,
                         DeferredLz4Body::Buffer* buf, std::vector<DeferredLz4Body::Block>* blocks) {
  const uint8_t* p = body + frame_off;
  int64_t at = 0;
  auto u32 = [&](int64_t o) { uint32_t v; std::memcpy(&v, p + o, 4); return v; };
  if (frame_len < 7 || u32(0) != 0x184D2204u) return false;
  const uint8_t flg = p[4], bd = p[5];
  if ((flg >> 6) != 1 || (flg & 0x02) || (flg & 0x01)) return false;   // version 01; reserved bit; dictionary id
  const bool block_checksum = (flg & 0x10) != 0, content_size = (flg & 0x08) != 0, content_checksum = (flg & 0x04) != 0;
  const int bsid = (bd >> 4) & 7;
  if (bsid < 4 || (bd & 0x8F)) return false;
  buf->block_max = 1u << (8 + 2 * bsid);   // 4: 64 KiB, 5: 256 KiB, 6: 1 MiB, 7: 4 MiB
  at = 6 + (content_size ? 8 : 0) + 1;     // + header checksum
  if (at > frame_len) return false;
  buf->first_block = static_cast<uint32_t>(blocks->size());
  while (true) {
    if (at + 4 > frame_len) return false;
    const uint32_t word = u32(at);
    at += 4;
    if (word == 0) break;   // end mark
    const uint32_t size = word & 0x7FFFFFFFu;
    if (size > buf->block_max || at + size + (block_checksum ? 4 : 0) > frame_len) return false;
    DeferredLz4Body::Block b;
    b.comp_off = static_cast<uint32_t>(frame_off + at);
    b.comp_size = size;
    b.buffer = buffer_index;
    b.stored = word >> 31;
    blocks->push_back(b);
    at += size + (block_checksum ? 4 : 0);
  }
  if (content_checksum && at + 4 > frame_len) return false;
  buf->n_blocks = static_cast<uint32_t>(blocks->size()) - buf->first_block;
  return true;
}

bool WalkZstdFrame(const uint8_t* body, int64_t frame_off, int64_t frame_len, uint32_t buffer_index, int64_t declared_len,
                   DeferredLz4Body::Buffer* buf, std::vector<DeferredLz4Body::Block>* blocks, std::vector<zstd::BlockInfo>* infos,
                   uint32_t* literal_scratch) {
  const uint8_t* p = body + frame_off;
  if (frame_len < 9 || p[0] != 0x28 || p[1] != 0xB5 || p[2] != 0x2F || p[3] != 0xFD) return false;
  const uint8_t fhd = p[4];
  const int fcs_flag = fhd >> 6;
  const bool single_segment = (fhd & 0x20) != 0;
  if ((fhd & 0x08) || (fhd & 0x04) || (fhd & 0x03)) return false;   // reserved bit; content checksum; dictionary id
  int64_t at = 5;
  uint64_t window = 0;
  if (!single_segment) {
    const uint8_t wd = p[at++];
    const uint64_t base = uint64_t(1) << (10 + (wd >> 3));
    window = base + (base >> 3) * (wd & 7u);
  }
  const int fcs_bytes = fcs_flag == 0 ? (single_segment ? 1 : 0) : fcs_flag == 1 ? 2 : fcs_flag == 2 ? 4 : 8;
  if (at + fcs_bytes > frame_len) return false;
  if (fcs_bytes) {
    uint64_t fcs = 0;
    std::memcpy(&fcs, p + at, static_cast<size_t>(fcs_bytes));   // little-endian host (the extension's platforms)
    if (fcs_bytes == 2) fcs += 256;
    if (fcs != static_cast<uint64_t>(declared_len)) return false;   // the host path words the error
    if (single_segment) window = fcs;
    at += fcs_bytes;
  }
  const uint64_t block_max = std::min<uint64_t>(std::max<uint64_t>(window, 1), zstd::kBlockMax);
  buf->block_max = zstd::kBlockMax;
  buf->first_block = static_cast<uint32_t>(blocks->size());
  const uint32_t none = ~0u;
  uint32_t last_huf = none, last_tbl[3] = {none, none, none};
  for (bool last = false; !last;) {
    if (at + 3 > frame_len) return false;
    const uint32_t h = static_cast<uint32_t>(p[at]) | (static_cast<uint32_t>(p[at + 1]) << 8) | (static_cast<uint32_t>(p[at + 2]) << 16);
    at += 3;
    last = (h & 1u) != 0;
    const uint32_t type = (h >> 1) & 3u, size = h >> 3;
    if (type == 3) return false;
    const uint32_t stored = type == 1 ? 1u : size;
    if ((type != 1 && size > block_max) || (type == 1 && size > block_max) || at + stored > frame_len) return false;
    const uint32_t self = static_cast<uint32_t>(blocks->size());
    DeferredLz4Body::Block b;
    b.comp_off = static_cast<uint32_t>(frame_off + at);
    b.comp_size = stored;
    b.buffer = buffer_index;
    b.stored = type == 0;
    zstd::BlockInfo z;
    std::memset(&z, 0, sizeof(z));
    z.comp_off = b.comp_off;
    z.comp_size = stored;
    z.type = type;
    z.huf_src = z.ll_src = z.of_src = z.ml_src = self;
    const uint8_t* c = p + at;
    if (type == 1) {
      z.regen = size;
      z.lit_pos = *literal_scratch;   // its one byte, written to the scratch like a literal
      *literal_scratch += 1;
      b.seq_cap = 256;
    } else if (type == 2) {
      if (size < 2) return false;
      // literals section header
      z.lit_type = c[0] & 3u;
      const uint32_t fmt = (c[0] >> 2) & 3u;
      if (z.lit_type < 2) {
        if (!(fmt & 1u)) { z.lit_hdr = 1; z.lit_regen = c[0] >> 3; }
        else if (fmt == 1) { z.lit_hdr = 2; z.lit_regen = (c[0] >> 4) | (static_cast<uint32_t>(c[1]) << 4); }
        else {
          if (size < 3) return false;
          z.lit_hdr = 3;
          z.lit_regen = (c[0] >> 4) | (static_cast<uint32_t>(c[1]) << 4) | (static_cast<uint32_t>(c[2]) << 12);
        }
        z.lit_comp = z.lit_type == 0 ? z.lit_regen : 1;
        z.lit_streams = 1;
      } else {
        if (size < 5) return false;
        const uint64_t v = static_cast<uint64_t>(c[0]) | (static_cast<uint64_t>(c[1]) << 8) | (static_cast<uint64_t>(c[2]) << 16) |
                           (static_cast<uint64_t>(c[3]) << 24) | (static_cast<uint64_t>(c[4]) << 32);
        if (fmt <= 1) { z.lit_hdr = 3; z.lit_regen = (v >> 4) & 0x3FFu; z.lit_comp = (v >> 14) & 0x3FFu; }
        else if (fmt == 2) { z.lit_hdr = 4; z.lit_regen = (v >> 4) & 0x3FFFu; z.lit_comp = (v >> 18) & 0x3FFFu; }
        else { z.lit_hdr = 5; z.lit_regen = (v >> 4) & 0x3FFFFu; z.lit_comp = (v >> 22) & 0x3FFFFu; }
        z.lit_streams = fmt == 0 ? 1 : 4;
        if (z.lit_type == 3) {
          if (last_huf == none) return false;
          z.huf_src = last_huf;
        } else {
          last_huf = self;
        }
        if (z.lit_comp == 0 || z.lit_regen == 0) return false;
      }
      if (z.lit_regen > zstd::kBlockMax || static_cast<uint64_t>(z.lit_hdr) + z.lit_comp + 1 > size) return false;
      if (z.lit_type == 0) {
        z.lit_pos = b.comp_off + z.lit_hdr;
      } else {
        z.lit_pos = *literal_scratch;
        *literal_scratch += (z.lit_regen + 3u) & ~3u;
      }
      // sequences section: the count, then (count > 0) the modes of the three tables
      z.seq_pos = z.lit_hdr + z.lit_comp;
      const uint8_t* q = c + z.seq_pos;
      const uint32_t left = size - z.seq_pos;
      if (q[0] == 0) { z.seq_hdr = 1; z.nseq = 0; }
      else if (q[0] < 128) { z.seq_hdr = 1; z.nseq = q[0]; }
      else if (q[0] < 255) {
        if (left < 2) return false;
        z.seq_hdr = 2;
        z.nseq = ((static_cast<uint32_t>(q[0]) - 128u) << 8) + q[1];
      } else {
        if (left < 3) return false;
        z.seq_hdr = 3;
        z.nseq = static_cast<uint32_t>(q[1]) + (static_cast<uint32_t>(q[2]) << 8) + 0x7F00u;
      }
      if (z.nseq == 0) {
        if (left != z.seq_hdr) return false;
      } else {
        if (left < z.seq_hdr + 2) return false;
        const uint32_t modes = q[z.seq_hdr];
        if (modes & 3u) return false;
        uint32_t* src[3] = {&z.ll_src, &z.of_src, &z.ml_src};
        for (int t = 0; t < 3; t++) {
          if (((modes >> (6 - 2 * t)) & 3u) == 3u) {
            if (last_tbl[t] == none) return false;
            *src[t] = last_tbl[t];
          } else {
            last_tbl[t] = self;
          }
        }
      }
      b.seq_cap = 256u * ((z.nseq + 1u + 255u) / 256u);
    }
    blocks->push_back(b);
    infos->push_back(z);
    at += stored;
  }
  if (at != frame_len) return false;   // a second frame, a skippable frame, trailing bytes: the host library's business
  buf->n_blocks = static_cast<uint32_t>(blocks->size()) - buf->first_block;
  return true;
}

void IPCStreamReader::DecompressBody(RecordBatchMeta* meta) {
  if (meta->compression != 0 && meta->compression != 1) throw IOException("Unknown BodyCompression codec " + std::to_string(meta->compression));
  const bool lz4 = meta->compression == 0;
  const ZstdApi& z = Zstd();
  const Lz4Api& l4 = Lz4();
  if (lz4 && !l4.ok)
    throw NotImplementedException("LZ4_FRAME compressed IPC body but liblz4.so.1 is not available on this host (the reference registers a ZSTD decompressor only)");
  if (!lz4 && !z.ok) throw NotImplementedException("ZSTD compressed IPC body but libzstd.so.1 is not available on this host");
  // pass 1: uncompressed sizes -> layout of the new body (every buffer 64-byte aligned); buffers of columns outside
  // the projection are neither read (DecodeBody) nor decompressed
  const std::vector<char> needed = NeededBuffers(*meta);
  const size_t nbuf = meta->buffers.size();
  std::vector<int64_t> bound;
  {
    size_t node = 0, variadic = 0;
    if (meta->is_dictionary) {
      std::function<const ArrowField*(const ArrowField&)> find = [&](const ArrowField& f) -> const ArrowField* {
        if (f.has_dictionary && f.dict_id == meta->dict_id) return &f;
        for (auto& c : f.children)
          if (const ArrowField* hit = find(c)) return hit;
        return nullptr;
      };
      for (auto& f : base_schema.fields)
        if (const ArrowField* hit = find(f)) { SubtreeBufferBounds(*hit, *meta, &node, &variadic, true, &bound); break; }
    } else {
      for (auto& f : base_schema.fields) SubtreeBufferBounds(f, *meta, &node, &variadic, false, &bound);
    }
    if (bound.size() != nbuf) bound.assign(nbuf, int64_t(1) << 40);  // metadata the walk cannot follow: validation reports it
  }
  std::vector<int64_t> ulen(nbuf, 0), opos(nbuf, 0);
  int64_t total = 0;
  for (size_t i = 0; i < nbuf; i++) {
    mi_buffer_span& b = meta->buffers[i];
    opos[i] = total;
    if (!needed.empty() && !needed[i]) {
      b.length = 0;
      continue;
    }
    if (b.length == 0) continue;
    if (b.length < 8 || !SpanInside(b.offset, b.length, cur_size))
      throw InternalException("Compressed buffer " + std::to_string(i) + " lies outside the message body");
    int64_t declared;
    std::memcpy(&declared, cur_ptr + b.offset, 8);
    const int64_t n = declared == -1 ? b.length - 8 : declared;
    if (n < 0) throw IOException("Compressed buffer " + std::to_string(i) + " declares a negative uncompressed length");
    if (n > bound[i] || total > (int64_t(1) << 41))
      throw IOException("Compressed buffer " + std::to_string(i) + " declares an uncompressed length of " + std::to_string(n) +
                        " bytes, more than its field node (" + std::to_string(bound[i]) + " bytes at most) can hold");
    if (!lz4 && declared != -1 && z.frame_content_size) {
      // the ZSTD frame header carries the content size too: a length prefix that disagrees with it is rejected before
      // anything is allocated for it (the reference finds out after decompressing: base_stream_reader.cpp:24-29)
      const unsigned long long fcs = z.frame_content_size(cur_ptr + b.offset + 8, static_cast<size_t>(b.length - 8));
      if (fcs < 0xFFFFFFFFFFFFFFFEull && fcs != static_cast<unsigned long long>(n))
        throw IOException("Expected decompressed size of " + std::to_string(n) + " bytes but got " + std::to_string(fcs) + " bytes");
    }
    ulen[i] = n;
    total += (n + 63) & ~static_cast<int64_t>(63);
  }
  // GPU consumers (SetDeferLz4): keep the body compressed and hand out the frame / block tables instead
  bool deferrable = (lz4 ? defer_lz4 : defer_zstd) && !meta->is_dictionary && base_schema.endianness == 0 && total < (int64_t(1) << 31) - 64 &&
                    cur_size < (int64_t(1) << 31) - 64;
  if (deferrable)
    for (auto& f : (HasProjection() ? projected_schema.fields : base_schema.fields))
      if (HasListField(f)) deferrable = false;
  if (deferrable) {
    auto d = std::make_shared<DeferredLz4Body>();
    d->comp = cur_ptr;
    d->comp_size = cur_size;
    d->codec = lz4 ? 0 : 1;
    bool ok = true;
    for (size_t i = 0; i < nbuf && ok; i++) {
      const mi_buffer_span& b = meta->buffers[i];
      if (b.length == 0 || ulen[i] == 0) continue;
      DeferredLz4Body::Buffer f;
      int64_t declared;
      std::memcpy(&declared, cur_ptr + b.offset, 8);
      f.raw = declared == -1;
      f.comp_off = b.offset + 8;
      f.comp_len = b.length - 8;
      f.out_off = opos[i];
      f.out_len = ulen[i];
      if (!f.raw)
        ok = lz4 ? WalkLz4Frame(cur_ptr, f.comp_off, f.comp_len, static_cast<uint32_t>(d->buffers.size()), &f, &d->blocks)
                 : WalkZstdFrame(cur_ptr, f.comp_off, f.comp_len, static_cast<uint32_t>(d->buffers.size()), ulen[i], &f, &d->blocks,
                                 &d->zblocks, &d->literal_scratch);
      d->buffers.push_back(f);
    }
    if (ok) {
      for (size_t i = 0; i < nbuf; i++) {
        mi_buffer_span& b = meta->buffers[i];
        b.offset = opos[i];
        b.length = (b.length == 0) ? 0 : ulen[i];
      }
      cur_deferred = d;
      cur_size = total;          // SliceBatch checks the spans against the decompressed layout; it never reads the body
      meta->compression = -1;
      return;
    }
  }
  uint8_t* out = nullptr;
  std::shared_ptr<void> owner = body_allocator ? body_allocator(static_cast<size_t>(total + 64), message.type, &out)
                                               : DefaultBodyAlloc(static_cast<size_t>(total + 64), message.type, &out);
  // pass 2: one frame per buffer, independent of each other -> the I/O pool's threads share them
  const uint8_t* in = cur_ptr;
  IoPool::Get().Run(static_cast<int>(nbuf), [&](int bi) {
    const size_t i = static_cast<size_t>(bi);
    mi_buffer_span& b = meta->buffers[i];
    if (b.length == 0) {
      b.offset = opos[i];
      return;
    }
    const uint8_t* src = in + b.offset;
    int64_t declared;
    std::memcpy(&declared, src, 8);
    const int64_t n = ulen[i];
    uint8_t* dst = out + opos[i];
    if (declared == -1) {
      std::memcpy(dst, src + 8, static_cast<size_t>(n));
    } else if (lz4) {
      Lz4DecompressFrame(l4, dst, n, src + 8, b.length - 8);
    } else {
      const size_t code = z.decompress(dst, static_cast<size_t>(n), src + 8, static_cast<size_t>(b.length - 8));
      if (z.is_error(code)) {
        throw IOException("ZSTD_decompress([buffer with " + std::to_string(b.length - 8) + " bytes] -> [buffer with " +
                          std::to_string(n) + " bytes]) failed with error '" + z.error_name(code) + "'");
      }
      if (static_cast<int64_t>(code) != n) {
        throw IOException("Expected decompressed size of " + std::to_string(n) + " bytes but got " + std::to_string(code) + " bytes");
      }
    }
    const int64_t padded = (n + 63) & ~static_cast<int64_t>(63);
    std::memset(dst + n, 0, static_cast<size_t>(padded - n));
    b.offset = opos[i];
    b.length = n;
  });
  compressed_owner = cur_owner;  // released with the next message
  cur_owner = owner;
  cur_ptr = out;
  cur_size = total;
  meta->compression = -1;
}

// Upper bound of the UNCOMPRESSED size of every RecordBatch.buffers entry, from the field nodes alone: a compressed buffer
// declares its own uncompressed length, and that number sizes an allocation (pinned, for scans) before a byte is decoded --
// a few damaged bytes must not be able to ask for terabytes.  Validity, fixed-width data and offsets are bounded by the
// node's row count; string data by the offset width (2 GiB for int32 offsets); what the walk cannot follow keeps 2^40.
static void SubtreeBufferBounds(const ArrowField& f, const RecordBatchMeta& meta, size_t* node, size_t* variadic, bool value_only,
                                std::vector<int64_t>* out) {
  constexpr int64_t kLoose = int64_t(1) << 40;
  const int64_t n = *node < meta.nodes.size() ? std::max<int64_t>(0, std::min<int64_t>(meta.nodes[*node].first, kLoose)) : kLoose;
  (*node)++;
  auto rows = [&](int64_t per_row, int64_t extra_rows = 0) {
    int64_t b = 0;
    if (per_row <= 0 || __builtin_mul_overflow(n + extra_rows, per_row, &b) || b > kLoose) return kLoose;
    return b + 64;
  };
  const int64_t bitmap = (n + 7) / 8 + 64;
  if (f.has_dictionary && !value_only) {
    out->push_back(bitmap);
    out->push_back(rows(f.dict_index_bit_width / 8));
    return;
  }
  switch (f.type) {
    case MI_AT_NULL: break;
    case MI_AT_STRUCT: case MI_AT_FIXED_LIST: out->push_back(bitmap); break;
    case MI_AT_UTF8: case MI_AT_BINARY: out->push_back(bitmap); out->push_back(rows(4, 1)); out->push_back((int64_t(1) << 31) + 64); break;
    case MI_AT_LARGE_UTF8: case MI_AT_LARGE_BINARY: out->push_back(bitmap); out->push_back(rows(8, 1)); out->push_back(kLoose); break;
    case MI_AT_LIST: case MI_AT_MAP: out->push_back(bitmap); out->push_back(rows(4, 1)); break;
    case MI_AT_LARGE_LIST: out->push_back(bitmap); out->push_back(rows(8, 1)); break;
    case MI_AT_UTF8_VIEW: case MI_AT_BINARY_VIEW: {
      out->push_back(bitmap);
      out->push_back(rows(16));
      const int64_t vc = *variadic < meta.variadic_counts.size() ? meta.variadic_counts[(*variadic)++] : 0;
      for (int64_t k = 0; k < vc && k < (1 << 20); k++) out->push_back(kLoose);
      break;
    }
    case MI_AT_UNION: out->push_back(rows(1)); if (f.unit == 1) out->push_back(rows(4)); break;
    case MI_AT_BOOL: out->push_back(bitmap); out->push_back(bitmap); break;
    default: {
      int32_t kind, w, nb;
      int64_t param;
      out->push_back(bitmap);
      int64_t width = 16;  // the widest fixed-width value of the format (decimal256 aside, which is not decoded)
      if (f.Plan(&kind, &param, &w, &nb, true)) {
        switch (kind) {
          case MI_K_COPY: case MI_K_FIXED_BINARY: width = param; break;
          case MI_K_DEC128: case MI_K_INTERVAL_MDN: width = 16; break;
          case MI_K_NARROW: width = param & 0xFF; break;
          case MI_K_HALF_FLOAT: width = 2; break;
          case MI_K_MUL_I32: case MI_K_INTERVAL_MONTHS: width = 4; break;
          default: width = 8; break;
        }
      } else if (f.type == MI_AT_DECIMAL) {
        width = 32;
      }
      out->push_back(rows(width));
      break;
    }
  }
  for (auto& c : f.children) SubtreeBufferBounds(c, meta, node, variadic, false, out);
}

// Number of RecordBatch.buffers entries a field subtree owns (same rules as the walk in SliceBatch)
static bool CountSubtreeBuffers(const ArrowField& f, const RecordBatchMeta& meta, size_t* variadic, size_t* buffers) {
  if (f.has_dictionary) {
    *buffers += 2;
    return true;
  }
  switch (f.type) {
    case MI_AT_NULL: break;
    case MI_AT_STRUCT: case MI_AT_FIXED_LIST: *buffers += 1; break;
    case MI_AT_UTF8: case MI_AT_BINARY: case MI_AT_LARGE_UTF8: case MI_AT_LARGE_BINARY: *buffers += 3; break;
    case MI_AT_UTF8_VIEW: case MI_AT_BINARY_VIEW: {
      if (*variadic >= meta.variadic_counts.size()) return false;
      const int64_t vc = meta.variadic_counts[(*variadic)++];
      if (vc < 0 || vc > (1 << 20)) return false;
      *buffers += 2 + static_cast<size_t>(vc);
      break;
    }
    case MI_AT_UNION: *buffers += f.unit == 1 ? 2 : 1; break;
    default: *buffers += 2; break;
  }
  for (auto& c : f.children)
    if (!CountSubtreeBuffers(c, meta, variadic, buffers)) return false;
  return true;
}

// Per RecordBatch.buffers entry: does a projected column own it?  Empty = all of them (no projection, dictionary batch,
// or metadata the walk cannot follow -- the full validation reports that).
std::vector<char> IPCStreamReader::NeededBuffers(const RecordBatchMeta& meta) const {
  std::vector<char> need;
  if (!HasProjection() || meta.is_dictionary) return need;
  std::vector<char> wanted(base_schema.fields.size(), 0);
  for (int32_t c : projected_columns) wanted[static_cast<size_t>(c)] = 1;
  need.assign(meta.buffers.size(), 0);
  size_t buf = 0, variadic = 0;
  for (size_t i = 0; i < base_schema.fields.size(); i++) {
    size_t n = 0;
    if (!CountSubtreeBuffers(base_schema.fields[i], meta, &variadic, &n) || buf + n > meta.buffers.size()) return {};
    if (wanted[i])
      for (size_t k = buf; k < buf + n; k++) need[k] = 1;
    buf += n;
  }
  return need;
}

std::vector<std::pair<int64_t, int64_t>> IPCStreamReader::ProjectedBodyRanges(const RecordBatchMeta& meta, int64_t body_length,
                                                                              int64_t gap) const {
  std::vector<std::pair<int64_t, int64_t>> ranges;
  const std::vector<char> needed = NeededBuffers(meta);
  if (needed.empty()) return ranges;
  std::vector<std::pair<int64_t, int64_t>> need;
  for (size_t k = 0; k < meta.buffers.size(); k++) {
    if (!needed[k]) continue;
    const mi_buffer_span& b = meta.buffers[k];
    if (b.length <= 0) continue;
    if (!SpanInside(b.offset, b.length, body_length)) return {};  // malformed: read everything, validation reports it
    need.emplace_back(b.offset, b.offset + ((b.length + 7) & ~int64_t(7)));  // + the 8-byte padding kernels may touch
  }
  std::sort(need.begin(), need.end());
  for (auto& r : need) {
    const int64_t hi = std::min(r.second, body_length);
    if (!ranges.empty() && r.first <= ranges.back().second + gap) ranges.back().second = std::max(ranges.back().second, hi);
    else ranges.emplace_back(r.first, hi);
  }
  if (ranges.empty()) ranges.emplace_back(0, 0);  // nothing to read at all (projection of empty buffers)
  return ranges;
}

// ------------------------------------------------------------------------------------------------ big-endian bodies
namespace {
enum class Swap : int { NONE = 0, W2 = 2, W4 = 4, W8 = 8, W16 = 16, W32 = 32, MONTH_DAY_NANO = 100, VIEW = 101 };

void SwapElements(uint8_t* p, int64_t bytes, Swap how) {
  switch (how) {
    case Swap::NONE: return;
    case Swap::W2: { uint16_t* v = reinterpret_cast<uint16_t*>(p); for (int64_t i = 0; i < bytes / 2; i++) v[i] = __builtin_bswap16(v[i]); return; }
    case Swap::W4: { uint32_t* v = reinterpret_cast<uint32_t*>(p); for (int64_t i = 0; i < bytes / 4; i++) v[i] = __builtin_bswap32(v[i]); return; }
    case Swap::W8: { uint64_t* v = reinterpret_cast<uint64_t*>(p); for (int64_t i = 0; i < bytes / 8; i++) v[i] = __builtin_bswap64(v[i]); return; }
    case Swap::W16: case Swap::W32: {  // one wide integer: the whole value is reversed
      const int w = static_cast<int>(how);
      for (int64_t i = 0; i + w <= bytes; i += w) std::reverse(p + i, p + i + w);
      return;
    }
    case Swap::MONTH_DAY_NANO:  // {int32 months, int32 days, int64 nanoseconds}
      for (int64_t i = 0; i + 16 <= bytes; i += 16) {
        uint32_t a, b;
        uint64_t c;
        std::memcpy(&a, p + i, 4);
        std::memcpy(&b, p + i + 4, 4);
        std::memcpy(&c, p + i + 8, 8);
        a = __builtin_bswap32(a);
        b = __builtin_bswap32(b);
        c = __builtin_bswap64(c);
        std::memcpy(p + i, &a, 4);
        std::memcpy(p + i + 4, &b, 4);
        std::memcpy(p + i + 8, &c, 8);
      }
      return;
    case Swap::VIEW:  // {int32 length, 12 inline bytes} or {int32 length, 4 prefix bytes, int32 buffer, int32 offset}
      for (int64_t i = 0; i + 16 <= bytes; i += 16) {
        uint32_t len;
        std::memcpy(&len, p + i, 4);
        len = __builtin_bswap32(len);
        std::memcpy(p + i, &len, 4);
        if (static_cast<int32_t>(len) > 12) {
          uint32_t bi, bo;
          std::memcpy(&bi, p + i + 8, 4);
          std::memcpy(&bo, p + i + 12, 4);
          bi = __builtin_bswap32(bi);
          bo = __builtin_bswap32(bo);
          std::memcpy(p + i + 8, &bi, 4);
          std::memcpy(p + i + 12, &bo, 4);
        }
      }
      return;
  }
}

// How every RecordBatch.buffers entry of a field subtree is stored (Arrow columnar format, "Endianness"): only multi-byte
// numbers are affected -- bitmaps, boolean data, string / binary payloads and fixed_size_binary values are byte sequences.
void SubtreeSwaps(const ArrowField& f, const RecordBatchMeta& meta, size_t* variadic, bool value_only, std::vector<Swap>* out) {
  auto of_width = [](int bytes) {
    switch (bytes) {
      case 2: return Swap::W2;
      case 4: return Swap::W4;
      case 8: return Swap::W8;
      case 16: return Swap::W16;
      case 32: return Swap::W32;
      default: return Swap::NONE;
    }
  };
  if (f.has_dictionary && !value_only) {
    out->push_back(Swap::NONE);
    out->push_back(of_width(f.dict_index_bit_width / 8));
    return;
  }
  switch (f.type) {
    case MI_AT_NULL: break;
    case MI_AT_STRUCT: case MI_AT_FIXED_LIST: out->push_back(Swap::NONE); break;
    case MI_AT_UTF8: case MI_AT_BINARY: case MI_AT_LIST: case MI_AT_MAP:
      out->push_back(Swap::NONE);
      out->push_back(Swap::W4);
      if (f.type == MI_AT_UTF8 || f.type == MI_AT_BINARY) out->push_back(Swap::NONE);
      break;
    case MI_AT_LARGE_UTF8: case MI_AT_LARGE_BINARY: case MI_AT_LARGE_LIST:
      out->push_back(Swap::NONE);
      out->push_back(Swap::W8);
      if (f.type != MI_AT_LARGE_LIST) out->push_back(Swap::NONE);
      break;
    case MI_AT_UTF8_VIEW: case MI_AT_BINARY_VIEW: {
      out->push_back(Swap::NONE);
      out->push_back(Swap::VIEW);
      const int64_t vc = *variadic < meta.variadic_counts.size() ? meta.variadic_counts[(*variadic)++] : 0;
      for (int64_t k = 0; k < vc && k < (1 << 20); k++) out->push_back(Swap::NONE);
      break;
    }
    case MI_AT_UNION: out->push_back(Swap::NONE); if (f.unit == 1) out->push_back(Swap::W4); break;
    case MI_AT_BOOL: case MI_AT_FIXED_BINARY: out->push_back(Swap::NONE); out->push_back(Swap::NONE); break;
    case MI_AT_INT: out->push_back(Swap::NONE); out->push_back(of_width(f.bit_width / 8)); break;
    case MI_AT_FLOAT: out->push_back(Swap::NONE); out->push_back(of_width(f.precision == 0 ? 2 : f.precision == 1 ? 4 : 8)); break;
    case MI_AT_DECIMAL: out->push_back(Swap::NONE); out->push_back(of_width(f.bit_width / 8)); break;
    case MI_AT_DATE: out->push_back(Swap::NONE); out->push_back(f.unit == 0 ? Swap::W4 : Swap::W8); break;
    case MI_AT_TIME: out->push_back(Swap::NONE); out->push_back(of_width(f.bit_width / 8)); break;
    case MI_AT_TIMESTAMP: case MI_AT_DURATION: out->push_back(Swap::NONE); out->push_back(Swap::W8); break;
    case MI_AT_INTERVAL:
      out->push_back(Swap::NONE);
      out->push_back(f.unit == 2 ? Swap::MONTH_DAY_NANO : Swap::W4);  // year_month: int32; day_time: two int32
      break;
    default: out->push_back(Swap::NONE); out->push_back(Swap::NONE); break;
  }
  for (auto& c : f.children) SubtreeSwaps(c, meta, variadic, false, out);
}
}  // namespace

void IPCStreamReader::SwapBodyEndianness(const RecordBatchMeta& meta) {
  std::vector<Swap> how;
  size_t variadic = 0;
  if (meta.is_dictionary) {
    std::function<const ArrowField*(const ArrowField&)> find = [&](const ArrowField& f) -> const ArrowField* {
      if (f.has_dictionary && f.dict_id == meta.dict_id) return &f;
      for (auto& c : f.children)
        if (const ArrowField* hit = find(c)) return hit;
      return nullptr;
    };
    for (auto& f : base_schema.fields)
      if (const ArrowField* hit = find(f)) { SubtreeSwaps(*hit, meta, &variadic, true, &how); break; }
  } else {
    for (auto& f : base_schema.fields) SubtreeSwaps(f, meta, &variadic, false, &how);
  }
  if (how.size() != meta.buffers.size()) return;  // metadata the walk cannot follow: the full validation reports it
  // the body must be ours to rewrite: caller-owned buffers (scan_arrow_ipc) are copied first
  if (!cur_owner) {
    uint8_t* copy = nullptr;
    std::shared_ptr<void> owner = DefaultBodyAlloc(static_cast<size_t>(cur_size) + 64, message.type, &copy);
    std::memcpy(copy, cur_ptr, static_cast<size_t>(cur_size));
    cur_owner = owner;
    cur_ptr = copy;
  }
  uint8_t* body = const_cast<uint8_t*>(cur_ptr);
  const std::vector<char> needed = NeededBuffers(meta);
  IoPool::Get().Run(static_cast<int>(how.size()), [&](int i) {
    const mi_buffer_span& b = meta.buffers[static_cast<size_t>(i)];
    if (how[static_cast<size_t>(i)] == Swap::NONE || b.length <= 0) return;
    if (!needed.empty() && !needed[static_cast<size_t>(i)]) return;    // never read from the file: nothing there to swap
    if (!SpanInside(b.offset, b.length, cur_size)) return;             // reported by SliceBatch
    SwapElements(body + b.offset, b.length, how[static_cast<size_t>(i)]);
  });
}

static std::string BufferSizeError(const std::string& column, int buffer, int64_t need, int64_t have) {
  return "Expected " + column + " buffer " + std::to_string(buffer) + " to have size >= " + std::to_string(need) +
         " bytes but found buffer with " + std::to_string(have) + " bytes";
}

void IPCStreamReader::SliceBatch(const RecordBatchMeta& meta, DecodedBatch* out) {
  out->length = meta.length;
  out->body = cur_ptr;
  out->body_size = cur_size;
  out->body_file_offset = cur_body_offset;
  out->is_dictionary = meta.is_dictionary;
  out->dict_id = meta.dict_id;
  out->is_delta = meta.is_delta;
  out->compression = meta.compression;
  out->owner = cur_owner;
  out->column_field.clear();
  out->null_count.clear();
  out->column_length.clear();
  out->buffers.clear();
  if (meta.compression != -1 && cur_size > 0) throw InternalException("compressed body reached SliceBatch");

  auto check_span = [&](const mi_buffer_span& s) {
    if (!SpanInside(s.offset, s.length, cur_size)) {
      throw InternalException("Buffer requires body offsets [" + std::to_string(s.offset) + ", " + std::to_string(s.offset) + " + " +
                              std::to_string(s.length) + ") but body has size " + std::to_string(cur_size));
    }
    if (s.offset % 8 != 0) throw InternalException("Buffer offset " + std::to_string(s.offset) + " is not 8-byte aligned");
  };

  // Depth-first walk over ALL fields keeps the node / buffer / variadic cursors of RecordBatch.{nodes,buffers,
  // variadicBufferCounts} in step; nodes are materialised only for the projected columns and their descendants.
  struct Cursor {
    size_t node = 0, buf = 0, variadic = 0;
  };
  std::function<int32_t(const ArrowField&, Cursor&, bool, int32_t, int32_t, bool)> walk =
      [&](const ArrowField& f, Cursor& cur, bool keep, int32_t parent, int32_t depth, bool value_only) -> int32_t {
    if (cur.node >= meta.nodes.size()) throw InternalException("RecordBatch has too few field nodes");
    const int64_t n = meta.nodes[cur.node].first;
    const int64_t nulls = meta.nodes[cur.node].second;
    cur.node++;
    if (n < 0) throw InternalException("Field node length is negative");
    // lengths come from the file: bound them before anything is multiplied by them (ArrowArrayViewValidate checks the
    // same relations), so a damaged RecordBatch cannot overflow a size computation and slip past the buffer checks
    if (n > (int64_t(1) << 40)) throw InternalException("Field node length " + std::to_string(n) + " is implausible");
    if (nulls < -1 || nulls > n) throw InternalException("Field node null_count " + std::to_string(nulls) + " is outside [0, length]");
    if (depth == 0 && !value_only && n != meta.length)
      throw InternalException("Expected array length " + std::to_string(meta.length) + " for column " + f.name + " but found " + std::to_string(n));
    if (keep && parent >= 0) {
      const DecodedNode& pn = out->nodes[static_cast<size_t>(parent)];
      const int32_t pt = pn.field->type;
      if (pt == MI_AT_STRUCT && n != pn.length)
        throw InternalException("Struct child " + f.name + " has length " + std::to_string(n) + ", its parent " + std::to_string(pn.length));
      if (pt == MI_AT_FIXED_LIST) {
        int64_t expect = 0;  // length <= 2^40 and listSize < 2^31: checked anyway, the product sizes buffers
        if (__builtin_mul_overflow(pn.length, static_cast<int64_t>(pn.field->byte_width), &expect) || n != expect)
          throw InternalException("Fixed-size list child " + f.name + " has length " + std::to_string(n) + ", expected " +
                                  std::to_string(pn.length) + " x " + std::to_string(pn.field->byte_width));
      }
    }
    const bool dict = f.has_dictionary && !value_only;
    size_t own;
    if (dict) {
      own = 2;
    } else {
      switch (f.type) {
        case MI_AT_NULL: own = 0; break;
        case MI_AT_STRUCT: case MI_AT_FIXED_LIST: own = 1; break;
        case MI_AT_UTF8: case MI_AT_BINARY: case MI_AT_LARGE_UTF8: case MI_AT_LARGE_BINARY: own = 3; break;
        case MI_AT_UTF8_VIEW: case MI_AT_BINARY_VIEW: {
          if (cur.variadic >= meta.variadic_counts.size()) throw InternalException("RecordBatch has too few variadicBufferCounts");
          const int64_t vc = meta.variadic_counts[cur.variadic++];
          if (vc < 0 || vc > (1 << 20)) throw InternalException("Invalid variadic buffer count");
          own = 2 + static_cast<size_t>(vc);
          break;
        }
        case MI_AT_UNION: own = f.unit == 1 ? 2 : 1; break;  // dense: types + offsets, sparse: types
        default: own = 2; break;
      }
    }
    if (cur.buf + own > meta.buffers.size()) throw InternalException("RecordBatch has too few buffers");
    int32_t idx = -1;
    if (keep) {
      idx = static_cast<int32_t>(out->nodes.size());
      out->nodes.emplace_back();
      DecodedNode nd;
      nd.field = &f;
      nd.parent = parent;
      nd.depth = depth;
      nd.length = n;
      nd.null_count = nulls;
      nd.value_only = value_only;
      for (size_t k = 0; k < own; k++) {
        nd.spans.push_back(meta.buffers[cur.buf + k]);
        check_span(nd.spans.back());
      }
      // size checks of ArrowArrayViewValidate (FULL), minus the data-dependent offsets walk (done on the device)
      int32_t kind, w, nb;
      int64_t param;
      if (f.Plan(&kind, &param, &w, &nb, value_only)) {
        const mi_buffer_span none{0, 0};
        const mi_buffer_span& s0 = own > 0 ? nd.spans[0] : none;
        const mi_buffer_span& s1 = own > 1 ? nd.spans[1] : none;
        if (s0.length != 0 && s0.length < (n + 7) / 8) throw InternalException(BufferSizeError(f.name, 0, (n + 7) / 8, s0.length));
        if (kind != MI_K_NULL && s0.length == 0 && n > 0 && nulls > 0)
          throw InternalException("Column " + f.name + " has null_count " + std::to_string(nulls) + " but no validity buffer");
        int64_t need1 = 0, per_row = 0, rows = n;
        switch (kind) {
          case MI_K_COPY: case MI_K_FIXED_BINARY: per_row = param; break;
          case MI_K_BOOL: need1 = (n + 7) / 8; break;
          case MI_K_DEC128: case MI_K_INTERVAL_MDN: case MI_K_STRVIEW: per_row = 16; break;
          case MI_K_DATE64: case MI_K_MUL_I64: case MI_K_DIV_I64: case MI_K_DURATION: per_row = 8; break;
          case MI_K_MUL_I32: case MI_K_INTERVAL_MONTHS: per_row = 4; break;
          case MI_K_STR32: case MI_K_LIST32: per_row = 4; rows = n > 0 ? n + 1 : 0; break;
          case MI_K_STR64: case MI_K_LIST64: per_row = 8; rows = n > 0 ? n + 1 : 0; break;
          case MI_K_DICT: case MI_K_NARROW: per_row = param & 0xFF; break;
          case MI_K_HALF_FLOAT: per_row = 2; break;
          default: break;
        }
        // rows <= 2^40 + 1 and widths come from the schema (validated, but up to 2^31 for fixed_size_binary): the
        // product is formed with an overflow check so that a wrapped size can never pass for a small one
        if (per_row < 0 || (per_row > 0 && __builtin_mul_overflow(rows, per_row, &need1)))
          throw InternalException("Column " + f.name + " needs more bytes than a buffer can hold (" + std::to_string(rows) + " x " + std::to_string(per_row) + ")");
        if (s1.length < need1) throw InternalException(BufferSizeError(f.name, 1, need1, s1.length));
      }
      out->nodes[static_cast<size_t>(idx)] = std::move(nd);
    }
    cur.buf += own;
    if (!dict) {
      for (auto& c : f.children) {
        const int32_t ci = walk(c, cur, keep, idx, depth + 1, false);
        if (keep) out->nodes[static_cast<size_t>(idx)].children.push_back(ci);
      }
    }
    return idx;
  };
  auto add_column = [&](int32_t top_index, int32_t node_idx) {
    const DecodedNode& nd = out->nodes[static_cast<size_t>(node_idx)];
    out->column_field.push_back(top_index);
    out->column_node.push_back(node_idx);
    out->null_count.push_back(nd.null_count);
    out->column_length.push_back(nd.length);
    for (size_t k = 0; k < 3; k++) out->buffers.push_back(k < nd.spans.size() ? nd.spans[k] : mi_buffer_span{0, 0});
  };
  out->nodes.clear();
  out->column_node.clear();

  if (meta.is_dictionary) {
    // one node: the dictionary values of the field that carries this id -- anywhere in the field tree (a list or struct
    // child may be dictionary-encoded too); `top` = the top-level column it belongs to
    const ArrowField* owner_field = nullptr;
    int32_t top = -1;
    std::function<const ArrowField*(const ArrowField&)> find = [&](const ArrowField& f) -> const ArrowField* {
      if (f.has_dictionary && f.dict_id == meta.dict_id) return &f;
      for (auto& c : f.children)
        if (const ArrowField* hit = find(c)) return hit;
      return nullptr;
    };
    for (size_t i = 0; i < base_schema.fields.size() && !owner_field; i++) {
      owner_field = find(base_schema.fields[i]);
      if (owner_field) top = static_cast<int32_t>(i);
    }
    if (!owner_field) throw IOException("DictionaryBatch refers to unknown dictionary id " + std::to_string(meta.dict_id));
    Cursor cur;
    add_column(top, walk(*owner_field, cur, true, -1, 0, /*value_only*/ true));
    return;
  }

  std::vector<int32_t> node_of_field(base_schema.fields.size(), -1);
  std::vector<char> wanted(base_schema.fields.size(), HasProjection() ? 0 : 1);
  for (int32_t c : projected_columns) wanted[static_cast<size_t>(c)] = 1;
  Cursor cur;
  for (size_t i = 0; i < base_schema.fields.size(); i++)
    node_of_field[i] = walk(base_schema.fields[i], cur, wanted[i] != 0, -1, 0, false);
  if (cur.node != meta.nodes.size()) {
    throw InternalException("Expected " + std::to_string(cur.node) + " field nodes in message but found " +
                            std::to_string(meta.nodes.size()));
  }
  if (HasProjection()) {
    for (int32_t c : projected_columns) add_column(c, node_of_field[static_cast<size_t>(c)]);
  } else {
    for (size_t i = 0; i < base_schema.fields.size(); i++) add_column(static_cast<int32_t>(i), node_of_field[i]);
  }
}

// ------------------------------------------------------------------------------------------------ file reader

IPCFileStreamReader::IPCFileStreamReader(const std::string& path_p) : path(path_p) {
  fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) {
    throw IOException("Cannot open file \"" + path + "\": " + std::strerror(errno));
  }
  struct stat st;
  if (fstat(fd, &st) != 0) {
    ::close(fd);
    fd = -1;
    throw IOException("Cannot stat file \"" + path + "\": " + std::strerror(errno));
  }
  file_size = st.st_size;
}

IPCFileStreamReader::~IPCFileStreamReader() {
  if (fd >= 0) ::close(fd);
}

void IPCFileStreamReader::PopulateNames(std::vector<std::string>& names) {
  GetBaseSchema();
  for (auto& f : base_schema.fields) names.push_back(f.name);
}

double IPCFileStreamReader::GetProgress() {
  if (file_size == 0) return 100;
  return (static_cast<double>(offset) / static_cast<double>(file_size)) * 100;
}

void ParallelFor(int n, const std::function<void(int)>& fn) { IoPool::Get().Run(n, fn); }
int IoThreads() { return IoPool::Get().Threads(); }

const uint8_t* IPCFileStreamReader::ReadData(uint8_t* ptr, idx_t size) {
  // BufferedFileReader::ReadData throws SerializationException when the file ends early
  constexpr idx_t kSlice = 256u << 10;  // smallest piece worth a thread hand-off
  if (size >= 4 * kSlice && IoPool::Get().Threads() > 1) {
    if (!SpanInside(offset, static_cast<int64_t>(size), file_size)) throw SerializationException();
    const int n = static_cast<int>(std::min<idx_t>((size + kSlice - 1) / kSlice, static_cast<idx_t>(2 * IoPool::Get().Threads())));
    const idx_t per = ((size + n - 1) / n + 4095) & ~static_cast<idx_t>(4095);
    const int64_t base = offset;
    IoPool::Get().Run(n, [&](int i) {
      idx_t lo = static_cast<idx_t>(i) * per, hi = std::min(size, lo + per);
      while (lo < hi) {
        ssize_t r = ::pread(fd, ptr + lo, hi - lo, static_cast<off_t>(base + static_cast<int64_t>(lo)));
        if (r < 0) {
          if (errno == EINTR) continue;
          throw IOException("Could not read from file \"" + path + "\": " + std::strerror(errno));
        }
        if (r == 0) throw SerializationException();
        lo += static_cast<idx_t>(r);
      }
    });
    offset += static_cast<int64_t>(size);
    return ptr;
  }
  idx_t done = 0;
  while (done < size) {
    ssize_t r = ::pread(fd, ptr + done, size - done, static_cast<off_t>(offset + static_cast<int64_t>(done)));
    if (r < 0) {
      if (errno == EINTR) continue;
      throw IOException("Could not read from file \"" + path + "\": " + std::strerror(errno));
    }
    if (r == 0) throw SerializationException();
    done += static_cast<idx_t>(r);
  }
  offset += static_cast<int64_t>(size);
  return ptr;
}

// Positions the reader on the next 8-byte boundary and reads one message prefix.  false = the file ends before a whole
// prefix could be read, which the reference treats as a clean end of stream (its aligned read throws
// SerializationException, ipc_file_stream_reader.cpp:103-129); the file size is known here, so no read is attempted.
bool IPCFileStreamReader::ReadPrefix() {
  const int64_t at = (offset + 7) & ~static_cast<int64_t>(7);
  if (at > file_size || file_size - at < static_cast<int64_t>(sizeof(message_prefix))) return false;
  offset = at;
  ReadData(reinterpret_cast<uint8_t*>(&message_prefix), sizeof(message_prefix));
  return true;
}

void IPCFileStreamReader::SkipBodyPadding() { offset = std::min(file_size, (offset + 7) & ~static_cast<int64_t>(7)); }

MessageType IPCFileStreamReader::ReadNextMessage() {
  static const char kFileMagic[8] = {'A', 'R', 'R', 'O', 'W', '1', 0, 0};
  while (!finished) {
    try {
      if (!ReadPrefix()) break;
    } catch (SerializationException&) {
      break;  // the file shrank under us: same outcome
    }
    // Arrow *file* format: the first 8 bytes are the magic, then comes an ordinary stream (ipc_file_stream_reader.cpp:107-119)
    if (offset == 8 && std::memcmp(kFileMagic, &message_prefix, 8) == 0) continue;
    if (message_prefix.continuation_token != kContinuationToken)
      throw IOException("Expected continuation token (0xFFFFFFFF) but got " + std::to_string(message_prefix.continuation_token));
    try {
      return FinishMessage();
    } catch (SerializationException& e) {
      throw IOException(std::string("SerializationException: ") + e.what());
    }
  }
  finished = true;
  return MessageType::UNINITIALIZED;
}

bool IPCFileStreamReader::DecodeHeader(const idx_t message_header_size) {
  // sizes come from the file: nothing is allocated for bytes the file cannot hold (BufferedFileReader::ReadData's
  // "not enough data in file to deserialize result")
  if (!SpanInside(offset, static_cast<int64_t>(message_prefix.metadata_size), file_size)) throw SerializationException();
  if (message_header.size() < message_header_size) message_header.resize(message_header_size);
  std::memcpy(message_header.data(), &message_prefix, sizeof(message_prefix));
  ReadData(message_header.data() + sizeof(message_prefix), static_cast<idx_t>(message_prefix.metadata_size));
  if (!ParseHeader(message_header.data(), message_header_size)) {
    finished = true;
    return true;
  }
  return false;
}

void IPCFileStreamReader::DecodeBody() {
  cur_owner.reset();
  cur_ptr = nullptr;
  cur_size = 0;
  if (message.body_length > 0) {
    SkipBodyPadding();
    cur_body_offset = offset;
    if (message.body_length > file_size - offset) throw SerializationException();  // before anything is allocated for it
    if (skip_record_batch_body && message.type == MessageType::RECORD_BATCH) {
      offset += message.body_length;  // step over the body without reading it
      return;
    }
    uint8_t* p = nullptr;
    bool compressed = false, stays_compressed = false;
    std::vector<std::pair<int64_t, int64_t>> ranges;
    if (message.type == MessageType::RECORD_BATCH || message.type == MessageType::DICTIONARY_BATCH) {
      const RecordBatchMeta meta = DecodeRecordBatch(message_meta, message_meta_len);
      compressed = meta.compression != -1;
      stays_compressed = ((defer_lz4 && meta.compression == 0) || (defer_zstd && meta.compression == 1)) && message.type == MessageType::RECORD_BATCH && base_schema.endianness == 0;
      // projection pushdown reaches the file: only the buffers of the projected columns are read (the reference reads
      // the whole body, ipc_file_stream_reader.cpp:71-89); what is skipped is never looked at
      if (message.type == MessageType::RECORD_BATCH) ranges = ProjectedBodyRanges(meta, message.body_length, 256 << 10);
    }
    // uncompressed bodies, and LZ4 bodies a GPU consumer decompresses itself, are copied to the device as they are: they
    // go where the consumer wants them (pinned staging); bodies the host decompresses only need to be readable here
    const bool to_device = !compressed || stays_compressed;
    cur_owner = (body_allocator && to_device) ? body_allocator(static_cast<size_t>(message.body_length), message.type, &p)
                                              : DefaultBodyAlloc(static_cast<size_t>(message.body_length), message.type, &p);
    if (ranges.empty()) {
      ReadData(p, static_cast<idx_t>(message.body_length));
    } else {
      if (!SpanInside(offset, message.body_length, file_size)) throw SerializationException();
      const int64_t body0 = offset;
      for (auto& r : ranges) {
        if (r.second <= r.first) continue;
        offset = body0 + r.first;
        ReadData(p + r.first, static_cast<idx_t>(r.second - r.first));
      }
      offset = body0 + message.body_length;
    }
    cur_ptr = p;
    cur_size = message.body_length;
  } else {
    cur_body_offset = offset;
  }
}

void IPCFileStreamReader::Seek(int64_t prefix_offset) {
  offset = prefix_offset;
  finished = false;
}

bool IPCFileStreamReader::IndexFromFooter() {
  // Arrow IPC *file*: "ARROW1\0\0" stream footer int32 footer_len "ARROW1".  The footer lists every dictionary and
  // record-batch block {offset, metaDataLength, bodyLength}: random access for sharding without walking the headers
  // (the reference notes this as future work: ipc_file_stream_reader.cpp:113-115, arrow_file_scan.cpp:36-40).
  if (file_size < 8 + 10) return false;
  uint8_t magic[8];
  if (::pread(fd, magic, 8, 0) != 8 || std::memcmp(magic, "ARROW1\0\0", 8) != 0) return false;
  uint8_t tail10[10];
  if (::pread(fd, tail10, 10, static_cast<off_t>(file_size - 10)) != 10 || std::memcmp(tail10 + 4, "ARROW1", 6) != 0) return false;
  int32_t flen;
  std::memcpy(&flen, tail10, 4);
  if (flen <= 0 || static_cast<int64_t>(flen) + 18 > file_size) return false;
  std::vector<uint8_t> tail(static_cast<size_t>(flen) + 10);
  if (::pread(fd, tail.data(), tail.size(), static_cast<off_t>(file_size - static_cast<int64_t>(tail.size()))) != static_cast<ssize_t>(tail.size()))
    return false;
  std::vector<FooterBlock> dict_blocks, batch_blocks;
  if (!DecodeFooter(tail.data(), static_cast<int64_t>(tail.size()), file_size, &dict_blocks, &batch_blocks)) return false;
  std::vector<uint8_t> meta;
  auto add = [&](const FooterBlock& b, MessageType type) {
    // block.metaDataLength covers the 8-byte prefix + the padded flatbuffer
    if (b.offset < 8 || b.meta_len < 8 || !SpanInside(b.offset, b.meta_len, file_size) || !SpanInside(b.offset + b.meta_len, b.body_len, file_size))
      throw IOException("Footer block out of bounds");
    BatchIndexEntry e{b.offset, b.meta_len - 8, static_cast<int32_t>(type), b.offset + b.meta_len, b.body_len, 0};
    meta.resize(static_cast<size_t>(b.meta_len - 8));
    if (::pread(fd, meta.data(), meta.size(), static_cast<off_t>(b.offset + 8)) != static_cast<ssize_t>(meta.size()))
      throw IOException("Could not read record batch metadata at offset " + std::to_string(b.offset));
    e.n_rows = DecodeRecordBatch(meta.data(), static_cast<int64_t>(meta.size())).length;
    index.push_back(e);
  };
  // stream order: dictionaries precede the batches that use them
  std::vector<std::pair<FooterBlock, MessageType>> all;
  for (auto& b : dict_blocks) all.emplace_back(b, MessageType::DICTIONARY_BATCH);
  for (auto& b : batch_blocks) all.emplace_back(b, MessageType::RECORD_BATCH);
  std::sort(all.begin(), all.end(), [](const auto& x, const auto& y) { return x.first.offset < y.first.offset; });
  for (auto& b : all)
    if (b.first.offset >= offset) add(b.first, b.second);
  return true;
}

const std::vector<BatchIndexEntry>& IPCFileStreamReader::BuildIndex() {
  if (index_built) return index;
  GetBaseSchema();
  if (IndexFromFooter()) {
    index_built = true;
    index_from_footer = true;
    return index;
  }
  index.clear();
  int64_t saved = offset;
  bool saved_finished = finished;
  // Walk headers only; bodies are skipped (the stream format has no footer index: SURVEY "Hard parts")
  std::vector<uint8_t> meta;
  int64_t pos = offset;
  while (true) {
    pos = (pos + 7) & ~static_cast<int64_t>(7);
    if (pos + 8 > file_size) break;
    ArrowIpcMessagePrefix p;
    offset = pos;
    try {
      ReadData(reinterpret_cast<uint8_t*>(&p), 8);
    } catch (SerializationException&) { break; }
    if (p.continuation_token != kContinuationToken || p.metadata_size <= 0) break;
    if (!SpanInside(pos + 8, p.metadata_size, file_size)) break;
    meta.resize(static_cast<size_t>(p.metadata_size));
    ReadData(meta.data(), static_cast<idx_t>(p.metadata_size));
    MessageHeader h = DecodeMessageHeader(meta.data(), p.metadata_size);
    BatchIndexEntry e{pos, p.metadata_size, static_cast<int32_t>(h.type), 0, h.body_length, 0};
    int64_t body = (offset + 7) & ~static_cast<int64_t>(7);
    e.body_offset = body;
    if (!SpanInside(body, h.body_length, file_size)) break;
    if (h.type == MessageType::RECORD_BATCH || h.type == MessageType::DICTIONARY_BATCH)
      e.n_rows = DecodeRecordBatch(meta.data(), p.metadata_size).length;
    index.push_back(e);
    pos = body + h.body_length;
  }
  offset = saved;
  finished = saved_finished;
  index_built = true;
  return index;
}

// ------------------------------------------------------------------------------------------------ buffer reader
IPCBufferStreamReader::IPCBufferStreamReader(std::vector<ArrowIPCBuffer> buffers_p) : buffers(std::move(buffers_p)) {}

// The caller's buffers are read in place (ipc_buffer_stream_reader.cpp:36-41: a pointer bump): `view` is the window over the
// buffer being consumed, `view_index` its place in the list; a buffer is opened when the first byte of it is asked for.
bool IPCBufferStreamReader::SeekUnreadByte() {
  while (!view.opened || view.pos >= view.size) {
    const idx_t next = view.opened ? view_index + 1 : view_index;
    if (next >= buffers.size()) return false;
    view_index = next;
    view.opened = true;
    view.ptr = reinterpret_cast<const uint8_t*>(static_cast<uintptr_t>(buffers[next].ptr));
    view.size = static_cast<int64_t>(buffers[next].size);
    view.pos = 0;
    if (!view.ptr && view.size > 0) throw IOException("Arrow IPC buffer " + std::to_string(next) + " is a NULL pointer");
  }
  return true;
}

const uint8_t* IPCBufferStreamReader::ReadData(idx_t size) {
  // the reference only asserts (ipc_buffer_stream_reader.cpp:37); a short buffer is reported instead of read past
  if (!SpanInside(view.pos, static_cast<int64_t>(size), view.size)) {
    throw IOException("Unexpected end of Arrow IPC buffer: need " + std::to_string(size) + " bytes at position " +
                      std::to_string(view.pos) + " of " + std::to_string(view.size));
  }
  const uint8_t* p = view.ptr + view.pos;
  view.pos += static_cast<int64_t>(size);
  return p;
}

MessageType IPCBufferStreamReader::ReadNextMessage() {
  while (!finished && SeekUnreadByte()) {
    prefix_at = ReadData(sizeof(message_prefix));
    std::memcpy(&message_prefix, prefix_at, sizeof(message_prefix));
    // An IPC *file* handed over as a buffer begins with the magic the file reader steps over
    // (ipc_file_stream_reader.cpp:116-119); the reference's buffer reader has no such case, accepting it is a superset
    const bool file_magic = view_index == 0 && view.pos == 8 && std::memcmp("ARROW1\0\0", prefix_at, 8) == 0;
    if (file_magic) continue;
    if (message_prefix.continuation_token != kContinuationToken)
      throw IOException("Expected continuation token (0xFFFFFFFF) but got " + std::to_string(message_prefix.continuation_token));
    return FinishMessage();
  }
  finished = true;   // every buffer is consumed (or the end-of-stream marker was seen before)
  return MessageType::UNINITIALIZED;
}

bool IPCBufferStreamReader::DecodeHeader(idx_t message_header_size) {
  // the decoder wants prefix + metadata as one span: the metadata follows the prefix in the caller's buffer, so the span
  // starts where the prefix was read
  (void)ReadData(static_cast<idx_t>(message_prefix.metadata_size));
  const bool end_of_stream = !ParseHeader(prefix_at, message_header_size);
  if (end_of_stream) finished = true;
  return end_of_stream;
}

void IPCBufferStreamReader::DecodeBody() {
  cur_owner.reset();
  if (message.body_length > 0) {
    // bodies are 8-byte aligned relative to the start of the stream
    int64_t aligned = (view.pos + 7) & ~static_cast<int64_t>(7);
    if (aligned != view.pos) ReadData(static_cast<idx_t>(aligned - view.pos));
    cur_body_offset = view.pos;
    cur_ptr = ReadData(static_cast<idx_t>(message.body_length));
    cur_size = message.body_length;
  } else {
    cur_body_offset = view.pos;
    cur_ptr = nullptr;
    cur_size = 0;
  }
}

double IPCBufferStreamReader::GetProgress() {
  if (buffers.empty()) return 100;
  double done = static_cast<double>(view_index);
  if (view.size > 0 && view_index < buffers.size()) done += static_cast<double>(view.pos) / static_cast<double>(view.size);
  return std::min(100.0, 100.0 * done / static_cast<double>(buffers.size()));
}

const std::vector<BatchIndexEntry>& IPCBufferStreamReader::BuildIndex() {
  if (index_built) return index;
  GetBaseSchema();
  // header walk over every buffer from the current position, without touching reader state
  int64_t global_base = 0;
  for (idx_t b = 0; b < buffers.size(); b++) {
    const uint8_t* base = reinterpret_cast<const uint8_t*>(static_cast<uintptr_t>(buffers[b].ptr));
    int64_t size = static_cast<int64_t>(buffers[b].size);
    int64_t pos = 0;
    if (view.opened && b < view_index) { global_base += size; continue; }
    if (b == view_index && view.opened) pos = view.pos;
    if (pos == 0 && b == 0 && size >= 8 && std::memcmp("ARROW1\0\0", base, 8) == 0) pos = 8;
    while (pos + 8 <= size) {
      ArrowIpcMessagePrefix p;
      std::memcpy(&p, base + pos, 8);
      if (p.continuation_token != kContinuationToken || p.metadata_size <= 0) break;
      if (!SpanInside(pos + 8, p.metadata_size, size)) break;
      MessageHeader h = DecodeMessageHeader(base + pos + 8, p.metadata_size);
      int64_t body = (pos + 8 + p.metadata_size + 7) & ~static_cast<int64_t>(7);
      if (!SpanInside(body, h.body_length, size)) break;
      BatchIndexEntry e{global_base + pos, p.metadata_size, static_cast<int32_t>(h.type), global_base + body, h.body_length, 0};
      if (h.type == MessageType::RECORD_BATCH || h.type == MessageType::DICTIONARY_BATCH)
        e.n_rows = DecodeRecordBatch(base + pos + 8, p.metadata_size).length;
      index.push_back(e);
      pos = body + h.body_length;
    }
    global_base += size;
  }
  index_built = true;
  return index;
}

}  // namespace miarrow
