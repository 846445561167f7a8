// scan_operator.cpp -- see scan_operator.hpp.
#include "scan_operator.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

namespace miarrow {

int WrapC(const std::function<void()>& f);  // c_api.cpp
void EnsureIoThreads(int n);                // ipc_stream_reader.cpp

namespace {
constexpr size_t kAlign = 256;
size_t RoundUp(size_t v, size_t a = kAlign) { return (v + a - 1) / a * a; }

std::map<std::string, std::string> ParseHive(const std::string& path) {
  // key=value path components (DuckDB's HivePartitioning::Parse behaviour for simple keys)
  std::map<std::string, std::string> out;
  size_t start = 0;
  while (start < path.size()) {
    size_t end = path.find_first_of("/\\", start);
    if (end == std::string::npos) break;  // the last component is the file name
    std::string part = path.substr(start, end - start);
    size_t eq = part.find('=');
    if (eq != std::string::npos && eq > 0 && eq + 1 < part.size()) out[part.substr(0, eq)] = part.substr(eq + 1);
    start = end + 1;
  }
  return out;
}

mi_string_t MakeHostString(const std::string& s) {
  mi_string_t r;
  std::memset(&r, 0, sizeof(r));
  r.value.inlined.length = static_cast<uint32_t>(s.size());
  if (s.size() <= 12) {
    std::memcpy(r.value.inlined.inlined, s.data(), s.size());
  } else {
    std::memcpy(r.value.pointer.prefix, s.data(), 4);
    r.value.pointer.ptr = reinterpret_cast<uint64_t>(s.data());
  }
  return r;
}

// ZSTD bodies in HBM pay only with many record batches side by side, and those need hardware queues of their own: the HIP
// runtime reads GPU_MAX_HW_QUEUES once, at its first call (default 4; the library asks for 24 when it is loaded, c_api.cpp).
bool ManyHardwareQueues() {
  const char* v = std::getenv("GPU_MAX_HW_QUEUES");
  return v != nullptr && std::atoi(v) >= 12;
}
bool DeferZstd(const mi_scan_options& o) { return o.host_decompress < 0 || (o.host_decompress == 0 && o.device_resident != 0 && ManyHardwareQueues()); }
int PipelineDepth(const mi_scan_options& o) { return std::max(2, std::min(kMaxDepth, o.pipeline_depth > 0 ? o.pipeline_depth : 3)); }
}  // namespace

ArrowScan::ArrowScan(Context* ctx_p, std::vector<std::string> paths, const mi_scan_options& o) : ctx(ctx_p), opts(o) {
  if (paths.empty()) throw InvalidInputException("read_arrow needs at least one file");
  for (auto& p : paths) {
    Source s;
    s.path = p;
    sources.push_back(std::move(s));
  }
  slots.resize(static_cast<size_t>(PipelineDepth(opts)));
  staging.resize(kMaxDepth + 2 * kMaxProducers + 2);   // the deepest pipeline's slots + queues + the bodies being read + a decompressed copy (buffers are allocated on first use)
}

ArrowScan::ArrowScan(Context* ctx_p, std::vector<ArrowIPCBuffer> buffers_p, const mi_scan_options& o)
    : ctx(ctx_p), opts(o), buffers(std::move(buffers_p)), is_buffers(true) {
  Source s;
  sources.push_back(std::move(s));
  slots.resize(static_cast<size_t>(PipelineDepth(opts)));
  staging.resize(kMaxDepth + 2 * kMaxProducers + 2);   // the deepest pipeline's slots + queues + the bodies being read + a decompressed copy (buffers are allocated on first use)
}

namespace {
inline int64_t TraceNow() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

ArrowScan::~ArrowScan() {
  StopProducer();
  if (trace)
    std::fprintf(stderr, "mi scan trace: %lld batches, %.2f in flight after a submit, %.2f ms from a submit to its batch being handed out; pipeline thread: enqueue %.3f s (K8: tables and copies %.3f s, launches %.3f s), waiting for input %.3f s, waiting for the GPU %.3f s, polling %.3f s; "
                         "producers (%d, summed): reading %.3f s, queue full %.3f s, no staging buffer %.3f s\n",
                 static_cast<long long>(stats.record_batches), stats.record_batches ? double(tr_inflight_sum) / stats.record_batches : 0.0,
                 stats.record_batches ? tr_latency_ns * 1e-6 / stats.record_batches : 0.0, tr_enqueue_ns * 1e-9, tr_k8_prep_ns * 1e-9, tr_k8_launch_ns * 1e-9, tr_fetch_wait_ns * 1e-9, tr_event_wait_ns * 1e-9, tr_poll_ns * 1e-9, n_producers,
                 tr_read_ns.load() * 1e-9, tr_push_wait_ns.load() * 1e-9, tr_lease_wait_ns.load() * 1e-9);
  try {
    ctx->Bind();
  } catch (...) {
  }
  (void)hipStreamSynchronize(ctx->h2d_stream);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipStreamSynchronize(ctx->d2h_stream);
  for (auto& s : slots) {
    s.batch = DecodedBatch();  // returns the staging lease
    s.plan.reset();
    s.gather_plan.reset();
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_out) (void)hipFree(s.d_out);
    if (s.h_out) (void)hipHostFree(s.h_out);
    if (s.h_status) (void)hipHostFree(s.h_status);
    if (s.h_counts) (void)hipHostFree(s.h_counts);
    if (s.h_aux) (void)hipHostFree(s.h_aux);
    if (s.d_aux) (void)hipFree(s.d_aux);
    if (s.lz4_stream) (void)hipStreamSynchronize(s.lz4_stream);
    if (s.d_comp) (void)hipFree(s.d_comp);
    if (s.d_lz4) (void)hipFree(s.d_lz4);
    if (s.h_lz4) (void)hipHostFree(s.h_lz4);
    if (s.h_mirror) (void)hipHostFree(s.h_mirror);
    if (s.lz4_stream && !s.lz4_stream_shared) (void)hipStreamDestroy(s.lz4_stream);
    if (s.lz4_done) (void)hipEventDestroy(s.lz4_done);
    if (s.h2d_done) (void)hipEventDestroy(s.h2d_done);
    if (s.compute_done) (void)hipEventDestroy(s.compute_done);
    if (s.d2h_done) (void)hipEventDestroy(s.d2h_done);
    if (s.filter_done) (void)hipEventDestroy(s.filter_done);
  }
  for (void* p : d_in_lists)
    if (p) (void)hipFree(p);
  for (void* p : retired_device) (void)hipFree(p);
  for (void* p : retired_host) (void)hipHostFree(p);
  dicts.clear();
  fetched.clear();
  extra_readers.clear();
  for (auto& st : staging)
    if (st.p) (void)hipHostFree(st.p);
}

ArrowScan::DictState::~DictState() {
  decode_plan.reset();
  if (d_data) (void)hipFree(d_data);
  if (d_validity) (void)hipFree(d_validity);
  if (h_data) (void)hipHostFree(h_data);
  if (h_words) (void)hipHostFree(h_words);   // h_validity is the same memory
  if (h_status) (void)hipHostFree(h_status);
  if (uploaded) (void)hipEventDestroy(uploaded);
}

void ArrowScan::OpenSource(size_t i) {
  Source& s = sources[i];
  if (s.opened) return;
  if (is_buffers) {
    s.reader = std::make_unique<IPCBufferStreamReader>(buffers);
  } else {
    s.reader = std::make_unique<IPCFileStreamReader>(s.path);
    if (opts.hive_partitioning) s.hive = ParseHive(s.path);
  }
  // LZ4_FRAME bodies stay compressed until they are in HBM (K8) when the consumer is on the device too.  A host consumer can
  // ask for it (host_decompress = -1: the string payloads come back beside the vectors, Slot::h_mirror), but by default its
  // bodies are decompressed by the reader's host threads: on this platform D2H copies run as copy kernels, which then
  // queue up with the K8 kernels instead of overlapping them (SF10: 0.85 s against 0.68 s, tools/lz4_bench.py)
  s.reader->SetDeferLz4(opts.host_decompress < 0 || (opts.host_decompress == 0 && opts.device_resident != 0));
  // ZSTD likewise, when the process has hardware queues for many record batches side by side (its entropy stage is one serial
  // chain per 128 KiB block: with the HIP runtime's default of 4 queues the reader's host threads are faster)
  s.reader->SetDeferZstd(DeferZstd(opts));
  s.reader->GetBaseSchema();
  s.opened = true;
}

const std::vector<ScanColumn>& ArrowScan::Bind() {
  if (bound) return all_columns;
  // schema of the first file (ArrowFileScan ctor, arrow_file_scan.cpp:9-23); union_by_name visits every file
  OpenSource(0);
  auto add_file_columns = [&](size_t si, bool first) {
    const ArrowSchemaModel& schema = sources[si].reader->GetBaseSchema();
    std::vector<std::string> names;
    for (auto& f : schema.fields) names.push_back(f.name);
    DeduplicateColumns(names);
    for (size_t c = 0; c < schema.fields.size(); c++) {
      auto it = std::find_if(all_columns.begin(), all_columns.end(), [&](const ScanColumn& sc) { return sc.name == names[c]; });
      if (it == all_columns.end()) {
        if (!first && !opts.union_by_name) continue;
        ScanColumn sc;
        sc.name = names[c];
        sc.field = schema.fields[c];
        all_columns.push_back(std::move(sc));
      }
    }
  };
  add_file_columns(0, true);
  if (opts.union_by_name) {
    for (size_t i = 1; i < sources.size(); i++) {
      OpenSource(i);
      add_file_columns(i, false);
    }
  }
  if (all_columns.empty()) {
    throw InvalidInputException("Provided table/dataframe must have at least one column");
  }
  if (opts.filename && !is_buffers) {
    ScanColumn sc;
    sc.name = "filename";
    sc.is_filename = true;
    sc.field.type = MI_AT_UTF8;
    sc.field.name = "filename";
    all_columns.push_back(sc);
  }
  if (opts.hive_partitioning && !is_buffers) {
    for (auto& kv : sources[0].hive) {
      ScanColumn sc;
      sc.name = kv.first;
      sc.is_hive = true;
      sc.hive_key = kv.first;
      sc.field.type = MI_AT_UTF8;
      sc.field.name = kv.first;
      all_columns.push_back(sc);
    }
  }
  bound = true;
  return all_columns;
}

void ArrowScan::SetFilter(FilterCnf cnf) {
  if (initialized) throw InvalidInputException("set the filter before mi_scan_init");
  if (has_filter) {  // a second filter is ANDed with the first
    filter.insert(filter.end(), cnf.begin(), cnf.end());
  } else {
    filter = std::move(cnf);
  }
  size_t leaves = 0;
  for (auto& c : filter) leaves += c.size();
  if (leaves > static_cast<size_t>(device::kMaxFilterLeaves))
    throw NotImplementedException("filter needs more than " + std::to_string(device::kMaxFilterLeaves) + " leaves");
  has_filter = true;
}

void ArrowScan::Init(const std::vector<std::string>& projected) {
  Bind();
  out_columns.clear();
  filter_only_columns.clear();
  if (projected.empty()) {
    out_columns = all_columns;
  } else {
    for (auto& name : projected) {
      auto it = std::find_if(all_columns.begin(), all_columns.end(), [&](const ScanColumn& sc) { return sc.name == name; });
      if (it == all_columns.end()) throw InternalException(std::string("Field '") + name + "' does not exist in IPC file schema");
      out_columns.push_back(*it);
    }
  }
  for (auto& c : out_columns) {
    if (c.is_filename || c.is_hive) continue;
    std::string why;
    if (!c.field.Supported(&why)) {
      throw NotImplementedException("Column '" + c.name + "': " + why + " is not decoded by the MI355X scan path yet");
    }
    if (c.field.has_dictionary && !opts.accept_dictionaries) {
      // the reference cannot read dictionary-encoded IPC at all (base_stream_reader.cpp:86-96)
      throw NotImplementedException("Column '" + c.name + "' is dictionary-encoded; enable accept_dictionaries");
    }
  }
  all_valid.assign(MI_VECTOR_SIZE / 64, ~0ull);
  ctx->Bind();
  compact = false;
  if (has_filter) {
    // resolve the leaves: a filter column is either projected (its decoded vector is reused) or decoded for the filter alone
    filter_columns.clear();
    std::vector<std::string> filter_names;
    for (auto& clause : filter) {
      for (auto& leaf : clause) {
        auto known = std::find(filter_names.begin(), filter_names.end(), leaf.column);
        if (known == filter_names.end()) {
          int32_t where = -1;
          for (size_t i = 0; i < out_columns.size(); i++)
            if (out_columns[i].name == leaf.column) where = static_cast<int32_t>(i);
          if (where < 0) {
            auto it = std::find_if(all_columns.begin(), all_columns.end(), [&](const ScanColumn& sc) { return sc.name == leaf.column; });
            if (it == all_columns.end()) throw InvalidInputException("filter column '" + leaf.column + "' does not exist in IPC file schema");
            filter_only_columns.push_back(*it);
            where = ~static_cast<int32_t>(filter_only_columns.size() - 1);
          }
          filter_names.push_back(leaf.column);
          filter_columns.push_back(where);
          known = filter_names.end() - 1;
        }
        leaf.out_col = static_cast<int32_t>(known - filter_names.begin());
        const int32_t where = filter_columns[static_cast<size_t>(leaf.out_col)];
        const ScanColumn& sc = where >= 0 ? out_columns[static_cast<size_t>(where)] : filter_only_columns[static_cast<size_t>(~where)];
        if (sc.is_filename || sc.is_hive) throw NotImplementedException("filter on the constant column '" + sc.name + "' is not pushed into the scan");
        if (leaf.op == device::kLeafIsNull || leaf.op == device::kLeafIsNotNull) {
          std::string why;
          if (!sc.field.Supported(&why)) throw NotImplementedException("Column '" + sc.name + "': " + why + " is not decoded by the MI355X scan path yet");
          continue;
        }
        int32_t kind, w, nb;
        int64_t param;
        // IN () -- an empty range -- keeps nothing (its negation every valid row) whatever the column holds
        if (!leaf.is_string && leaf.op == device::kLeafRange && !leaf.lo_open && !leaf.hi_open && leaf.lo > leaf.hi) continue;
        if (leaf.is_string) {
          // byte-string constants: the column must decode to string_t rows that point into ONE data buffer
          const bool value_ok = sc.field.Plan(&kind, &param, &w, &nb, /*value_only*/ true) && (kind == MI_K_STR32 || kind == MI_K_STR64 || kind == MI_K_FIXED_BINARY);
          if (!value_ok)
            throw NotImplementedException("string filter pushdown on column '" + sc.name + "' (" + sc.field.DuckType() +
                                          ") needs a utf8 / large_utf8 / binary / fixed_size_binary column (dictionary-encoded or not)");
          continue;
        }
        const bool ok = sc.field.Plan(&kind, &param, &w, &nb) && !sc.field.has_dictionary &&
                        (kind == MI_K_COPY || kind == MI_K_DEC128 || kind == MI_K_DATE64 || kind == MI_K_MUL_I32 || kind == MI_K_MUL_I64 ||
                         kind == MI_K_DIV_I64 || kind == MI_K_NARROW || kind == MI_K_BOOL) &&
                        (w == 1 || w == 2 || w == 4 || w == 8) && sc.field.type != MI_AT_FLOAT;
        if (!ok)
          throw NotImplementedException("filter pushdown on column '" + sc.name + "' (" + sc.field.DuckType() +
                                        ") needs an integer / boolean / date / time / timestamp / decimal(<=18) column");
      }
    }
    // IN-lists live in HBM for the lifetime of the scan
    for (void* p : d_in_lists)
      if (p) (void)hipFree(p);
    d_in_lists.clear();
    for (auto& clause : filter)
      for (auto& leaf : clause) {
        void* p = nullptr;
        if (leaf.op == device::kLeafIn) {
          MI_HIP_CHECK(hipMalloc(&p, leaf.in_values.size() * 8));
          MI_HIP_CHECK(hipMemcpy(p, leaf.in_values.data(), leaf.in_values.size() * 8, hipMemcpyHostToDevice));
        } else if ((leaf.op == device::kLeafStrIn || leaf.op == device::kLeafStrRange) && !leaf.str_values.empty()) {
          // 3 words per constant (its string_t image + the device address of its bytes), the bytes behind the table
          const size_t nc = leaf.str_values.size();
          size_t bytes = 0;
          for (auto& v : leaf.str_values) bytes += RoundUp(v.size() + 1, 8);
          std::vector<uint8_t> img(nc * 24 + bytes, 0);
          MI_HIP_CHECK(hipMalloc(&p, img.size()));
          size_t at = nc * 24;
          for (size_t k = 0; k < nc; k++) {
            const std::string& v = leaf.str_values[k];
            if (v.size() > 0xFFFFFFFFull) throw InvalidInputException("string filter constant too long");
            uint32_t dw[3] = {0, 0, 0};
            std::memcpy(dw, v.data(), std::min<size_t>(v.size(), v.size() <= 12 ? 12 : 4));
            const uint64_t w0 = static_cast<uint64_t>(v.size()) | (static_cast<uint64_t>(dw[0]) << 32);
            const uint64_t w1 = v.size() <= 12 ? (static_cast<uint64_t>(dw[1]) | (static_cast<uint64_t>(dw[2]) << 32)) : 0;
            const uint64_t w2 = reinterpret_cast<uint64_t>(static_cast<uint8_t*>(p) + at);
            std::memcpy(&img[k * 24], &w0, 8);
            std::memcpy(&img[k * 24 + 8], &w1, 8);
            std::memcpy(&img[k * 24 + 16], &w2, 8);
            std::memcpy(&img[at], v.data(), v.size());
            at += RoundUp(v.size() + 1, 8);
          }
          MI_HIP_CHECK(hipMemcpy(p, img.data(), img.size(), hipMemcpyHostToDevice));
        }
        d_in_lists.push_back(p);
      }
    if (opts.filter_compact) {
      for (auto& c : out_columns) {
        if (c.is_filename || c.is_hive) continue;
        int32_t kind, w, nb;
        int64_t param;
        c.field.Plan(&kind, &param, &w, &nb);
        if (!device::KindCanGather(kind) || !c.field.children.empty())
          throw NotImplementedException("filter_compact needs flat projected columns: '" + c.name + "' (" + c.field.DuckType() +
                                        ") is decoded window by window, use the selection vector instead");
      }
      compact = true;
    }
  }
  for (auto& s : slots) InitSlot(s);
  initialized = true;
}

void ArrowScan::InitSlot(Slot& s) {
  if (s.h2d_done) return;
  ctx->Bind();
  MI_HIP_CHECK(hipEventCreateWithFlags(&s.h2d_done, hipEventDisableTiming));
  MI_HIP_CHECK(hipEventCreateWithFlags(&s.compute_done, hipEventDisableTiming));
  MI_HIP_CHECK(hipEventCreateWithFlags(&s.d2h_done, hipEventDisableTiming));
  MI_HIP_CHECK(hipEventCreateWithFlags(&s.filter_done, hipEventDisableTiming));
  s.plan = std::make_unique<Plan>(ctx);
  s.gather_plan = std::make_unique<Plan>(ctx);
  MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.h_status), 64, hipHostMallocDefault));
  s.h_status[0] = s.h_status[1] = s.h_status[2] = 0;
}

// More record batches in flight / held by the caller at once (the COPY pump hands whole batches to several sink threads).
// Takes effect only before the first batch has been requested: the read-ahead thread sizes its staging ring from the slot
// count (a pump that asks later works with the slots there are: it waits for a release when all of them are held).
void ArrowScan::EnsurePipelineDepth(int depth) {
  depth = std::min(depth, kMaxDepth);
  if (static_cast<int>(slots.size()) >= depth) return;
  if (producer_started || !inflight.empty()) return;   // a scan that has started keeps the depth it has
  std::vector<Slot> bigger(static_cast<size_t>(depth));
  for (size_t i = 0; i < slots.size(); i++) bigger[i] = std::move(slots[i]);
  slots = std::move(bigger);
  staging.resize(kMaxDepth + 2 * kMaxProducers + 2);   // the deepest pipeline's slots + queues + the bodies being read + a decompressed copy (buffers are allocated on first use)
  if (initialized)
    for (auto& s : slots) InitSlot(s);
}

void ArrowScan::EnsureSlotBuffers(Slot& s, size_t in_bytes, size_t out_bytes) {
  ctx->Bind();
  if (in_bytes > s.d_in_cap) {
    if (s.d_in) RetireDevice(s.d_in);
    s.d_in = nullptr;
    s.d_in_cap = RoundUp(GrowCap(in_bytes, s.d_in_cap), 1 << 16);
    MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s.d_in), s.d_in_cap));
  }
  if (out_bytes > s.d_out_cap) {
    if (s.d_out) RetireDevice(s.d_out);
    s.d_out = nullptr;
    s.d_out_cap = RoundUp(GrowCap(out_bytes, s.d_out_cap), 1 << 16);
    MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s.d_out), s.d_out_cap));
  }
}

void ArrowScan::EnsureHostOut(Slot& s, size_t bytes) {
  if (opts.device_resident || bytes <= s.h_out_cap) return;
  ctx->Bind();
  if (s.h_out) RetireHost(s.h_out);
  s.h_out = nullptr;
  Context::PreferNode near_the_gpu(ctx);   // (the caller's thread allocates: only its policy, for the length of this call)
  s.h_out_cap = RoundUp(GrowCap(bytes, s.h_out_cap), 1 << 16);
  MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.h_out), s.h_out_cap, hipHostMallocDefault));
}

ArrowScan::Slot* ArrowScan::FreeSlot() {
  for (auto& s : slots)
    if (!s.busy) return &s;
  return nullptr;
}

void ArrowScan::DecodeDictionary(Source& src, const DecodedBatch& b) {
  ctx->Bind();
  (void)src;
  if (b.column_node.empty() || b.nodes.empty()) throw InternalException("DictionaryBatch without a value node");
  const ArrowField& f = *b.nodes[static_cast<size_t>(b.column_node[0])].field;  // the field that carries the id (any depth)
  int32_t kind, w, nb;
  int64_t param;
  if (!f.Plan(&kind, &param, &w, &nb, /*value_only*/ true))
    throw NotImplementedException("Dictionary value type " + f.Format() + " is not decoded by the MI355X scan path");
  // the values are decoded as ONE flat task below: value types that need more than {validity, buffer 1, buffer 2}
  // (string views: a table of variadic buffers; lists / structs: child nodes) are refused instead of mis-wired
  switch (kind) {
    case MI_K_COPY: case MI_K_BOOL: case MI_K_DEC128: case MI_K_DATE64: case MI_K_MUL_I32: case MI_K_MUL_I64: case MI_K_DIV_I64:
    case MI_K_STR32: case MI_K_STR64: case MI_K_FIXED_BINARY: case MI_K_DURATION: case MI_K_INTERVAL_MONTHS: case MI_K_INTERVAL_MDN:
    case MI_K_NARROW: case MI_K_HALF_FLOAT:
      break;
    default:
      throw NotImplementedException("Dictionary of value type " + f.Format() + " (field '" + f.name +
                                    "') is not decoded by the MI355X scan path: only flat value types are");
  }
  // isDelta: the new values are appended to the existing dictionary (indices keep their meaning); otherwise the
  // dictionary is replaced.  Either way a NEW version is built; batches already in flight keep theirs.
  std::shared_ptr<DictState> old = dicts.count(b.dict_id) ? dicts[b.dict_id] : nullptr;
  const bool delta = b.is_delta && old;
  if (delta && old->kind != kind) throw IOException("Delta dictionary changes the value type");
  auto d = std::make_shared<DictState>();
  const int64_t n_new = b.column_length[0];
  const int64_t n_old = delta ? old->dict_len : 0;
  const int64_t n = n_old + n_new;
  d->dict_len = n;
  d->kind = kind;
  d->out_width = w;
  if (delta) {
    d->d_heaps = old->d_heaps;
    d->host_bodies = old->host_bodies;
  }
  if (b.owner) d->host_bodies.push_back(b.owner);
  const size_t data_bytes = RoundUp(static_cast<size_t>(n + 1) * static_cast<size_t>(w));
  const size_t valid_bytes = RoundUp(static_cast<size_t>((n + 1 + 63) / 64) * 8);
  // Nothing below waits for the device: the body goes up on the copy stream, the values are decoded on the compute stream
  // behind it (the record batches that use the dictionary follow on the same stream), the decode's status word comes back
  // with the first such batch (DictState::h_status, checked in AcquireBatch).  The validity words are built on the host
  // from the Arrow bitmap (the value types admitted above are flat: a value is NULL exactly when its bit says so).
  MI_HIP_CHECK(hipMalloc(&d->d_data, data_bytes));
  MI_HIP_CHECK(hipMalloc(&d->d_validity, valid_bytes));
  MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&d->h_words), valid_bytes, hipHostMallocDefault));
  MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&d->h_status), 64, hipHostMallocDefault));
  d->h_status[0] = 0;
  MI_HIP_CHECK(hipMemsetAsync(d->d_data, 0, data_bytes, ctx->stream));
  uint8_t* heap = nullptr;
  if (b.body_size > 0) {
    void* p = nullptr;
    MI_HIP_CHECK(hipMalloc(&p, RoundUp(static_cast<size_t>(b.body_size) + 16)));
    d->d_heaps.push_back(std::shared_ptr<void>(p, [](void* q) { (void)hipFree(q); }));
    heap = static_cast<uint8_t*>(p);
    // the body is pinned (the read-ahead's allocator for DICTIONARY_BATCH messages) and lives in host_bodies
    MI_HIP_CHECK(hipMemcpyAsync(heap, b.body, static_cast<size_t>(b.body_size), hipMemcpyHostToDevice, ctx->h2d_stream));
    if (!d->uploaded) MI_HIP_CHECK(hipEventCreateWithFlags(&d->uploaded, hipEventDisableTiming));
    MI_HIP_CHECK(hipEventRecord(d->uploaded, ctx->h2d_stream));
    MI_HIP_CHECK(hipStreamWaitEvent(ctx->stream, d->uploaded, 0));
  }
  if (n_old > 0)
    MI_HIP_CHECK(hipMemcpyAsync(d->d_data, old->d_data, static_cast<size_t>(n_old) * static_cast<size_t>(w), hipMemcpyDeviceToDevice, ctx->stream));
  uint64_t* words = d->h_words;
  for (size_t i = 0; i < valid_bytes / 8; i++) words[i] = ~0ull;
  auto set_bit = [&](int64_t i, bool v) {
    if (v) words[static_cast<size_t>(i >> 6)] |= 1ull << (i & 63);
    else words[static_cast<size_t>(i >> 6)] &= ~(1ull << (i & 63));
  };
  for (int64_t i = 0; i < n_old; i++) set_bit(i, old->host_valid[static_cast<size_t>(i)] != 0);
  {
    const mi_buffer_span* sp = &b.buffers[0];
    const bool has_bitmap = sp[0].length > 0 && b.null_count[0] != 0;
    if (has_bitmap && sp[0].length < (n_new + 7) / 8) throw InternalException("Arrow IPC validation failed: dictionary validity bitmap is too short");
    for (int64_t i = 0; i < n_new; i++) set_bit(n_old + i, !has_bitmap || ((b.body[sp[0].offset + (i >> 3)] >> (i & 7)) & 1));
  }
  if (n_new > 0) {
    // decode the new values into a tile-aligned scratch vector, then append (the scratch lives as long as the version:
    // freeing it here would wait for the device)
    void* scratch_data = nullptr;
    MI_HIP_CHECK(hipMalloc(&scratch_data, RoundUp(static_cast<size_t>(n_new) * static_cast<size_t>(w) + 16)));
    d->d_heaps.push_back(std::shared_ptr<void>(scratch_data, [](void* q) { (void)hipFree(q); }));
    mi_col_task t;
    std::memset(&t, 0, sizeof(t));
    const mi_buffer_span* sp = &b.buffers[0];
    t.validity = sp[0].length ? heap + sp[0].offset : nullptr;
    t.buf1 = heap + sp[1].offset;
    t.buf2 = nb > 2 ? heap + sp[2].offset : nullptr;
    t.buf2_len = nb > 2 ? sp[2].length : 0;
    t.out_data = scratch_data;
    t.out_validity = nullptr;   // built on the host, above
    const int64_t data_off = nb > 2 ? sp[2].offset : sp[1].offset;
    t.ptr_base = opts.device_resident ? reinterpret_cast<uint64_t>(heap + data_off) : reinterpret_cast<uint64_t>(b.body + data_off);
    t.nrows = n_new;
    t.null_count = b.null_count[0];
    t.kind = kind;
    t.param = param;
    d->decode_plan = std::make_unique<Plan>(ctx, &t, 1);
    d->decode_plan->Launch(ctx->stream);
    MI_HIP_CHECK(hipMemcpyAsync(static_cast<uint8_t*>(d->d_data) + static_cast<size_t>(n_old) * static_cast<size_t>(w), scratch_data,
                                static_cast<size_t>(n_new) * static_cast<size_t>(w), hipMemcpyDeviceToDevice, ctx->stream));
    MI_HIP_CHECK(hipMemcpyAsync(d->h_status, d->decode_plan->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  }
  // string-valued dictionaries keep their values on the host too: pushed-down string predicates are matched against the
  // dictionary once and against the rows by index (the offsets are validated here; the device validates them again)
  d->host_valid.resize(static_cast<size_t>(n));
  for (int64_t i = 0; i < n; i++) d->host_valid[static_cast<size_t>(i)] = (words[static_cast<size_t>(i >> 6)] >> (i & 63)) & 1;
  if (kind == MI_K_STR32 || kind == MI_K_STR64 || kind == MI_K_FIXED_BINARY) {
    if (delta) d->host_strings = old->host_strings;
    const mi_buffer_span* sp = &b.buffers[0];
    for (int64_t i = 0; i < n_new; i++) {
      const bool ok = d->host_valid[static_cast<size_t>(n_old + i)] != 0;
      std::string v;
      if (ok) {
        if (kind == MI_K_FIXED_BINARY) {
          v.assign(reinterpret_cast<const char*>(b.body + sp[1].offset + i * param), static_cast<size_t>(param));
        } else {
          int64_t o0, o1;
          if (kind == MI_K_STR32) {
            int32_t a, c;
            std::memcpy(&a, b.body + sp[1].offset + 4 * i, 4);
            std::memcpy(&c, b.body + sp[1].offset + 4 * (i + 1), 4);
            o0 = a; o1 = c;
          } else {
            std::memcpy(&o0, b.body + sp[1].offset + 8 * i, 8);
            std::memcpy(&o1, b.body + sp[1].offset + 8 * (i + 1), 8);
          }
          if (o0 < 0 || o1 < o0 || o1 > sp[2].length) throw InternalException("Arrow IPC validation failed: dictionary offsets");
          v.assign(reinterpret_cast<const char*>(b.body + sp[2].offset + o0), static_cast<size_t>(o1 - o0));
        }
      }
      d->host_strings.push_back(std::move(v));
    }
  }
  set_bit(n, false);  // the extra NULL entry at index dict_len (ColumnArrowToDuckDBDictionary)
  MI_HIP_CHECK(hipMemcpyAsync(d->d_validity, words, valid_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (!opts.device_resident) {
    // host consumers read the values from pinned memory: the copy rides the compute stream too, and every batch that uses
    // the dictionary is handed out only after its own results have come back behind it
    MI_HIP_CHECK(hipHostMalloc(&d->h_data, data_bytes, hipHostMallocDefault));
    MI_HIP_CHECK(hipMemcpyAsync(d->h_data, d->d_data, data_bytes, hipMemcpyDeviceToHost, ctx->stream));
    d->h_validity = words;   // the same pinned words
  }
  dicts[b.dict_id] = d;
}

// Tables the tasks read from HBM (list windows, string-view buffers) travel in a pinned aux buffer beside the body.
void ArrowScan::UploadAux(Slot& s, const std::vector<uint64_t>& aux) {
  const size_t aux_bytes = aux.size() * 8;
  if (!aux_bytes) return;
  if (aux_bytes > s.h_aux_cap) {
    if (s.h_aux) RetireHost(s.h_aux);
    if (s.d_aux) RetireDevice(s.d_aux);
    s.h_aux_cap = s.d_aux_cap = RoundUp(aux_bytes * 2, 4096);
    MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.h_aux), s.h_aux_cap, hipHostMallocDefault));
    MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s.d_aux), s.d_aux_cap));
  }
  std::memcpy(s.h_aux, aux.data(), aux_bytes);
  MI_HIP_CHECK(hipMemcpyAsync(s.d_aux, s.h_aux, aux_bytes, hipMemcpyHostToDevice, ctx->h2d_stream));
}

// Stage A of a record batch: H2D of the body, the full-width decode tasks (every projected column; with compaction only
// the filter columns), the filter, the fused aggregate and -- unless the batch is compacted, which needs the selected row
// count on the host first (stage B) -- the copy back.
void ArrowScan::EnqueueBatch(Slot& s) {
  ctx->Bind();
  const DecodedBatch& b = s.batch;
  Source& src = sources[static_cast<size_t>(s.source)];
  const int64_t n = b.length;
  s.nrows = n;
  s.compact = compact && n > 0;
  s.needs_stage_b = false;
  const int64_t n_windows = (n + MI_VECTOR_SIZE - 1) / MI_VECTOR_SIZE;
  // d_in must be final before tasks take addresses inside it
  EnsureSlotBuffers(s, static_cast<size_t>(b.body_size) + 64, 0);

  PlannerOptions po;
  po.array_align = kAlign;
  // DirectConversion: a plain fixed-width column without NULLs needs no kernel at all -- its vector IS the
  // Arrow buffer.  For a device-resident consumer that is the HBM copy of the body; for a HOST consumer it is the pinned host
  // body itself, so the column crosses PCIe in neither direction (lineitem: 7 of 16 columns, 46 of 175 B/row in, 46 of 158
  // B/row back).  On unless asked otherwise (-1) -- except while a consumer reads the vectors on the GPU (the fused COPY).
  const bool zero_copy = opts.zero_copy_direct >= 0 && (opts.device_resident || !keep_on_device);
  // a host consumer of a body that only exists decompressed in HBM: string_t rows point into a pinned mirror of the body
  const bool mirror = b.deferred && !opts.device_resident;
  po.zero_copy_direct = zero_copy && !agg.on && !s.compact && !mirror;
  if (mirror && static_cast<size_t>(b.body_size) + 64 > s.h_mirror_cap) {
    if (s.h_mirror) RetireHost(s.h_mirror);
    s.h_mirror = nullptr;
    s.h_mirror_cap = RoundUp(GrowCap(static_cast<size_t>(b.body_size) + 64, s.h_mirror_cap), 1 << 16);
    MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.h_mirror), s.h_mirror_cap, hipHostMallocDefault));
  }
  po.unset_all_valid = opts.unset_all_valid != 0;
  s.planner.opts = po;
  s.planner.Clear();
  s.col_root.assign(out_columns.size(), -1);
  s.absent.assign(out_columns.size(), {0, 0});
  s.filter_root.assign(filter_columns.size(), -1);
  s.node_dict.clear();

  BatchPlacement where;
  where.batch = &b;
  where.in_base = s.d_in;
  where.consumer_base = opts.device_resident ? reinterpret_cast<uint64_t>(s.d_in)
                                             : reinterpret_cast<uint64_t>(mirror ? s.h_mirror : b.body);
  where.dict_len = [&](int64_t id) -> int64_t {
    auto it = dicts.find(id);
    if (it == dicts.end()) throw IOException("RecordBatch uses dictionary id " + std::to_string(id) + " before its DictionaryBatch");
    return it->second->dict_len;
  };
  // filter columns read their decoded vectors from HBM: never aliased into a body that may not even be uploaded
  std::vector<char> no_alias(b.nodes.size(), 0);
  if (has_filter) {
    for (size_t k = 0; k < filter_columns.size(); k++) {
      const int32_t wc = filter_columns[k];
      const int32_t fc = wc >= 0 ? src.out_to_file_column[static_cast<size_t>(wc)] : src.filter_to_file_column[static_cast<size_t>(~wc)];
      if (fc >= 0) no_alias[static_cast<size_t>(b.column_node[static_cast<size_t>(fc)])] = 1;
    }
    where.no_alias = &no_alias;
  }
  std::vector<int32_t> widths(out_columns.size(), 0);
  auto width_of = [](const ScanColumn& c) {
    int32_t kind, w, nb;
    int64_t param;
    c.field.Plan(&kind, &param, &w, &nb);
    return w;
  };
  // ---- layout.  Full decode: [projected columns | sel | counts] travel back, then the filter-only columns.
  //      Compaction (stage A): only the filter columns + sel + counts; the projected columns are planned in stage B.
  if (!s.compact) {
    for (size_t c = 0; c < out_columns.size(); c++) {
      if (out_columns[c].is_filename || out_columns[c].is_hive) continue;
      widths[c] = width_of(out_columns[c]);
      const int32_t fc = src.out_to_file_column[c];
      if (fc < 0) {  // column absent in this file (union_by_name): an all-NULL vector
        s.absent[c] = s.planner.AddAbsentColumn(n, widths[c]);
        continue;
      }
      s.col_root[c] = s.planner.AddColumn(where, b.column_node[static_cast<size_t>(fc)]);
    }
  }
  if (has_filter) {
    s.sel_off = s.planner.Reserve(static_cast<size_t>(n) * 4 + 16);
    s.sel_count_off = s.planner.Reserve(static_cast<size_t>(n_windows) * 4 + 16);
  }
  s.d2h_bytes = s.compact ? 0 : s.planner.arena_bytes;
  if (has_filter) {
    for (size_t k = 0; k < filter_columns.size(); k++) {
      const int32_t wc = filter_columns[k];
      if (wc >= 0 && !s.compact) {
        s.filter_root[k] = s.col_root[static_cast<size_t>(wc)];
        continue;
      }
      const int32_t fc = wc >= 0 ? src.out_to_file_column[static_cast<size_t>(wc)] : src.filter_to_file_column[static_cast<size_t>(~wc)];
      if (fc >= 0) s.filter_root[k] = s.planner.AddColumn(where, b.column_node[static_cast<size_t>(fc)]);
    }
  }
  s.stage_a_bytes = s.planner.arena_bytes;
  s.node_dict.resize(s.planner.nodes.size());
  for (size_t i = 0; i < s.planner.nodes.size(); i++)
    if (s.planner.nodes[i].dict_id >= 0) s.node_dict[i] = dicts[s.planner.nodes[i].dict_id];
  // worst case for stage B: every row selected
  size_t stage_b_worst = 0;
  if (s.compact) {
    stage_b_worst = 4096;
    for (auto& c : out_columns)
      if (!c.is_filename && !c.is_hive)
        stage_b_worst += RoundUp(static_cast<size_t>(n) * static_cast<size_t>(std::max(width_of(c), 1)) + 16) + RoundUp(static_cast<size_t>((n + 63) / 64) * 8 + 8);
  }
  EnsureSlotBuffers(s, static_cast<size_t>(b.body_size) + 64, s.stage_a_bytes + stage_b_worst + 64);
  EnsureHostOut(s, s.d2h_bytes + 64);
  if (static_cast<size_t>(n_windows + 1) * 4 > s.h_counts_cap) {
    if (s.h_counts) RetireHost(s.h_counts);
    s.h_counts_cap = RoundUp(static_cast<size_t>(n_windows + 1) * 8, 4096);
    MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.h_counts), s.h_counts_cap, hipHostMallocDefault));
  }
  UploadAux(s, s.planner.aux);
  s.planner.Rebase(0, s.d_out, s.d_aux);

  stats.record_batches++;
  s.h_status[2] = 0;
  if (b.deferred) {
    EnqueueLz4(s);   // compressed body -> HBM -> K8 kernels -> d_in; ctx->stream waits for them
    if (mirror) {    // the payload of every string-like buffer goes back to the host as soon as it is decompressed
      for (const DecodedNode& nd : b.nodes) {
        if (!nd.field) continue;
        size_t first = 0, last = 0;   // spans [first, last) hold bytes that string_t rows point at
        switch (nd.field->type) {
          case MI_AT_UTF8: case MI_AT_BINARY: case MI_AT_LARGE_UTF8: case MI_AT_LARGE_BINARY: first = 2; last = 3; break;
          case MI_AT_FIXED_BINARY: first = 1; last = 2; break;
          case MI_AT_UTF8_VIEW: case MI_AT_BINARY_VIEW: first = 2; last = nd.spans.size(); break;
          default: break;
        }
        for (size_t k = first; k < last && k < nd.spans.size(); k++)
          if (nd.spans[k].length > 0)
          {
            MI_HIP_CHECK(hipMemcpyAsync(s.h_mirror + nd.spans[k].offset, s.d_in + nd.spans[k].offset, static_cast<size_t>(nd.spans[k].length),
                                        hipMemcpyDeviceToHost, ctx->d2h_stream));
            stats.d2h_bytes += nd.spans[k].length;
          }
      }
    }
  } else {
  // ---- H2D of the body on the copy stream: only what the kernels read (projected columns; with zero_copy_direct not even
  // all of those): merge the buffer ranges, gaps below 64 KiB are cheaper to copy than to split.  A full scan is one copy.
  std::vector<std::pair<int64_t, int64_t>> upload = s.planner.upload;
  if (opts.device_resident)  // aliased vectors of a GPU consumer point into the HBM copy of the body
    for (auto& nd : s.planner.nodes)
      if (nd.alias_body_off >= 0) upload.emplace_back(nd.alias_body_off, nd.nrows * nd.width);
  if (s.compact) {  // stage B reads every projected column
    for (size_t c = 0; c < out_columns.size(); c++) {
      const int32_t fc = (out_columns[c].is_filename || out_columns[c].is_hive) ? -1 : src.out_to_file_column[c];
      if (fc < 0) continue;
      for (const auto& sp : b.nodes[static_cast<size_t>(b.column_node[static_cast<size_t>(fc)])].spans)
        if (sp.length > 0) upload.emplace_back(sp.offset, sp.length);
    }
  }
  if (b.body_size > 0) {
    std::sort(upload.begin(), upload.end());
    int64_t lo = -1, hi = -1;
    auto flush = [&]() {
      if (lo < 0) return;
      hi = std::min<int64_t>((hi + 63) & ~int64_t(63), b.body_size);
      MI_HIP_CHECK(hipMemcpyAsync(s.d_in + lo, b.body + lo, static_cast<size_t>(hi - lo), hipMemcpyHostToDevice, ctx->h2d_stream));
      stats.h2d_bytes += hi - lo;
    };
    for (const auto& r : upload) {
      if (lo >= 0 && r.first <= hi + (64 << 10)) {
        hi = std::max(hi, r.first + r.second);
        continue;
      }
      flush();
      lo = r.first & ~int64_t(63);
      hi = r.first + r.second;
    }
    flush();
  }
  }
  MI_HIP_CHECK(hipEventRecord(s.h2d_done, ctx->h2d_stream));
  MI_HIP_CHECK(hipStreamWaitEvent(ctx->stream, s.h2d_done, 0));
  // absent columns: all-NULL vectors (data 0, validity 0)
  if (!s.compact) {
    for (size_t c = 0; c < out_columns.size(); c++) {
      if (out_columns[c].is_filename || out_columns[c].is_hive) continue;
      if (src.out_to_file_column[c] < 0 && n > 0) {
        MI_HIP_CHECK(hipMemsetAsync(s.d_out + s.absent[c].first, 0, static_cast<size_t>(n) * static_cast<size_t>(std::max(widths[c], 1)), ctx->stream));
        MI_HIP_CHECK(hipMemsetAsync(s.d_out + s.absent[c].second, 0, static_cast<size_t>((n + 63) / 64) * 8, ctx->stream));
      }
    }
  }
  s.plan->Set(s.planner.tasks.data(), static_cast<int32_t>(s.planner.tasks.size()), ctx->stream);
  MI_HIP_CHECK(hipMemsetAsync(s.plan->d_status, 0, sizeof(uint32_t), ctx->stream));
  s.plan->Launch(ctx->stream);
  if (has_filter && n > 0) {
    device::FilterProgram prog;
    std::memset(&prog, 0, sizeof(prog));
    size_t li = 0;
    for (auto& clause : filter) {
      for (size_t j = 0; j < clause.size(); j++, li++) {
        const FilterLeaf& leaf = clause[j];
        device::FilterLeafDev& L = prog.leaves[prog.n_leaves++];
        const int32_t wc = filter_columns[static_cast<size_t>(leaf.out_col)];
        const ScanColumn& sc = wc >= 0 ? out_columns[static_cast<size_t>(wc)] : filter_only_columns[static_cast<size_t>(~wc)];
        const int32_t root = s.filter_root[static_cast<size_t>(leaf.out_col)];
        L.op = leaf.op;
        L.flags = (j + 1 == clause.size() ? device::kLeafEndsClause : 0) | (leaf.negate ? device::kLeafNegate : 0);
        L.lo = leaf.lo;
        L.hi = leaf.hi;
        L.in_values = static_cast<const int64_t*>(d_in_lists[li]);
        L.n_in = static_cast<int32_t>(leaf.is_string ? leaf.str_values.size() : leaf.in_values.size());
        if (leaf.op == device::kLeafStrRange)
          L.n_in = (leaf.lo_open ? 0 : 1) | (leaf.lo_incl ? 2 : 0) | (leaf.hi_open ? 0 : 4) | (leaf.hi_incl ? 8 : 0);
        L.width = 1;
        if (root < 0) {
          // the column is absent from this file (union_by_name): every row is NULL -- IS NULL keeps every row, everything
          // else keeps none (an empty, non-negated range over any readable bytes: the selection buffer itself)
          L.validity = nullptr;
          L.flags &= ~device::kLeafNegate;
          if (leaf.op == device::kLeafIsNull) {
            L.op = device::kLeafIsNotNull;
          } else {
            L.op = device::kLeafRange;
            L.lo = 1;
            L.hi = 0;
            L.data = s.d_out + s.sel_off;
          }
          continue;
        }
        const PlannedNode& pn = s.planner.nodes[static_cast<size_t>(root)];
        L.data = pn.alias_body_off >= 0 ? static_cast<const void*>(s.d_in + pn.alias_body_off) : static_cast<const void*>(s.d_out + pn.data_off);
        L.validity = pn.valid_off >= 0 ? reinterpret_cast<const uint64_t*>(s.d_out + pn.valid_off) : nullptr;
        L.width = std::max(pn.width, 1);
        const bool null_test = leaf.op == device::kLeafIsNull || leaf.op == device::kLeafIsNotNull;
        if (pn.kind == MI_K_DICT && (leaf.is_string || null_test)) {
          // dictionary-encoded: match the dictionary version this batch uses once (host), the rows by index.  IS [NOT] NULL
          // goes the same way: a row is NULL when its index or its dictionary entry is
          const std::shared_ptr<DictState>& dict = s.node_dict[static_cast<size_t>(root)];
          if (!dict || (leaf.is_string && static_cast<int64_t>(dict->host_strings.size()) != dict->dict_len))
            throw NotImplementedException("string filter on the dictionary-encoded column '" + leaf.column + "': its dictionary values are not strings");
          auto it = dict->match_maps.find(li);
          if (it == dict->match_maps.end()) {
            std::vector<uint8_t> codes(static_cast<size_t>(dict->dict_len) + 1, 0);
            auto passes = [&](const std::string& v) {   // std::string compares byte-wise (unsigned), a proper prefix first
              if (leaf.op != device::kLeafStrRange) return std::binary_search(leaf.str_values.begin(), leaf.str_values.end(), v);
              auto cmp = [](const std::string& a, const std::string& b) {
                const int c = std::memcmp(a.data(), b.data(), std::min(a.size(), b.size()));
                return c != 0 ? c : (a.size() < b.size() ? -1 : a.size() > b.size() ? 1 : 0);
              };
              if (!leaf.lo_open) {
                const int c = cmp(v, leaf.str_values[0]);
                if (c < 0 || (c == 0 && !leaf.lo_incl)) return false;
              }
              if (!leaf.hi_open) {
                const int c = cmp(v, leaf.str_values[1]);
                if (c > 0 || (c == 0 && !leaf.hi_incl)) return false;
              }
              return true;
            };
            for (int64_t e = 0; e < dict->dict_len; e++)
              codes[static_cast<size_t>(e)] = !dict->host_valid[static_cast<size_t>(e)] ? 2
                                              : (leaf.is_string && passes(dict->host_strings[static_cast<size_t>(e)])) ? 1 : 0;
            codes[static_cast<size_t>(dict->dict_len)] = 2;   // the NULL entry rows without a value point at
            // device copy + its pinned source, uploaded on the compute stream in front of the filter kernel that reads it
            void* p = nullptr;
            void* hp = nullptr;
            MI_HIP_CHECK(hipMalloc(&p, RoundUp(codes.size() + 16)));
            MI_HIP_CHECK(hipHostMalloc(&hp, RoundUp(codes.size() + 16), hipHostMallocDefault));
            std::shared_ptr<void> keep(p, [hp](void* q) {
              (void)hipFree(q);
              (void)hipHostFree(hp);
            });
            std::memcpy(hp, codes.data(), codes.size());
            MI_HIP_CHECK(hipMemcpyAsync(p, hp, codes.size(), hipMemcpyHostToDevice, ctx->stream));
            it = dict->match_maps.emplace(li, std::move(keep)).first;
          }
          L.op = device::kLeafDictMap;
          L.in_values = static_cast<const int64_t*>(it->second.get());
          L.n_in = static_cast<int32_t>(std::min<int64_t>(dict->dict_len + 1, 0x7FFFFFFF));
          L.lo = leaf.op == device::kLeafIsNull ? 2 : leaf.op == device::kLeafIsNotNull ? 3 : leaf.negate ? 1 : 0;
          L.flags &= ~device::kLeafNegate;   // applied inside the kernel: a NULL entry fails = and <> alike
          L.width = 4;
          continue;
        }
        if (leaf.is_string) {
          // the rows' long-string pointers are consumer addresses (pn.ptr_base = byte 0 of the Arrow data buffer as the
          // consumer sees it); the kernel reads the bytes from the HBM copy of that buffer
          const DecodedNode& dn = b.nodes[static_cast<size_t>(pn.source_node)];
          const size_t data_span = pn.kind == MI_K_FIXED_BINARY ? 1 : 2;
          L.lo = static_cast<int64_t>(reinterpret_cast<uintptr_t>(s.d_in + (dn.spans.size() > data_span ? dn.spans[data_span].offset : 0)));
          L.hi = static_cast<int64_t>(pn.ptr_base);
          continue;
        }
        if (sc.field.type == MI_AT_INT && !sc.field.is_signed) {
          L.flags |= device::kLeafUnsigned;
          if (pn.width == 8 && leaf.op != device::kLeafIsNull && leaf.op != device::kLeafIsNotNull) {
            // uint64: compared through the order-preserving map x ^ 2^63 on both sides.  Constants arrive as int64, a
            // negative one is below every value of the column.
            L.flags |= device::kLeafBias;
            const int64_t bias = static_cast<int64_t>(0x8000000000000000ull);
            if (leaf.op == device::kLeafRange) {
              if (!leaf.hi_open && leaf.hi < 0) { L.lo = 1; L.hi = 0; }   // empty (its negation keeps every valid row, as it must)
              else {
                L.lo = (leaf.lo_open || leaf.lo < 0) ? bias : (leaf.lo ^ bias);            // bias = the image of 0
                L.hi = leaf.hi_open ? static_cast<int64_t>(0x7FFFFFFFFFFFFFFFull) : (leaf.hi ^ bias);   // image of UINT64_MAX
              }
            }
            // IN-lists of uint64 columns are uploaded unbiased: compare them unbiased too
            if (leaf.op == device::kLeafIn) L.flags &= ~device::kLeafBias;
          }
        }
      }
    }
    MI_HIP_CHECK(device::LaunchFilterProgram(prog, n, reinterpret_cast<mi_sel_t*>(s.d_out + s.sel_off),
                                             reinterpret_cast<uint32_t*>(s.d_out + s.sel_count_off), ctx->stream));
  }
  if (agg.on && n > 0) {
    // fused consumer: the decoded vectors are read once more by the aggregate kernel and never leave HBM
    device::AggSumProductArgs a;
    std::memset(&a, 0, sizeof(a));
    auto column = [&](int32_t c, const void** data, const uint64_t** valid, int32_t* width) {
      if (s.col_root[static_cast<size_t>(c)] < 0) throw InvalidInputException("aggregate column '" + out_columns[static_cast<size_t>(c)].name + "' is absent from a file of the scan");
      const PlannedNode& o = s.planner.nodes[static_cast<size_t>(s.col_root[static_cast<size_t>(c)])];
      *data = s.d_out + o.data_off;
      *valid = o.valid_off >= 0 ? reinterpret_cast<const uint64_t*>(s.d_out + o.valid_off) : nullptr;
      *width = o.width;
    };
    a.n_filters = static_cast<int32_t>(agg.filter_cols.size());
    for (int32_t k = 0; k < a.n_filters; k++) {
      column(agg.filter_cols[static_cast<size_t>(k)], &a.fcol[k], &a.fvalid[k], &a.fwidth[k]);
      a.lo[k] = agg.lo[static_cast<size_t>(k)];
      a.hi[k] = agg.hi[static_cast<size_t>(k)];
    }
    column(agg.col_a, &a.a, &a.avalid, &a.awidth);
    column(agg.col_b, &a.b, &a.bvalid, &a.bwidth);
    a.nrows = n;
    MI_HIP_CHECK(device::LaunchAggSumProduct(a, agg.d_acc, ctx->num_cus, ctx->stream));
    agg.rows_scanned += n;
  }
  MI_HIP_CHECK(hipEventRecord(s.compute_done, ctx->stream));
  MI_HIP_CHECK(hipStreamWaitEvent(ctx->d2h_stream, s.compute_done, 0));
  if (has_filter && n > 0)  // the per-window counts always come back (tiny): chunk sizes, Count(), the stage-B layout
    MI_HIP_CHECK(hipMemcpyAsync(s.h_counts, s.d_out + s.sel_count_off, static_cast<size_t>(n_windows) * 4, hipMemcpyDeviceToHost, ctx->d2h_stream));
  if (s.compact) {
    MI_HIP_CHECK(hipEventRecord(s.filter_done, ctx->d2h_stream));
    s.needs_stage_b = true;
    return;
  }
  s.host_vectors = !opts.device_resident && !agg.on && s.d2h_bytes > 0 && !keep_on_device;
  if (s.host_vectors) {
    MI_HIP_CHECK(hipMemcpyAsync(s.h_out, s.d_out, s.d2h_bytes, hipMemcpyDeviceToHost, ctx->d2h_stream));
    stats.d2h_bytes += static_cast<int64_t>(s.d2h_bytes);
  }
  for (const auto& nd : s.planner.nodes)
    if (nd.alias_body_off >= 0) stats.aliased_bytes += nd.nrows * nd.width;
  // the device status word travels with the results instead of costing a stream-wide synchronisation
  MI_HIP_CHECK(hipMemcpyAsync(&s.h_status[0], s.plan->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->d2h_stream));
  s.h_status[1] = 0;
  MI_HIP_CHECK(hipEventRecord(s.d2h_done, ctx->d2h_stream));
}

// K8: the record batch arrived with its LZ4_FRAME buffers still compressed (DecodedBatch::deferred).  The compressed bytes
// cross PCIe, the frames' blocks (tables built by the host reader from the block headers) are expanded into s.d_in at the
// decompressed layout every span of the batch already refers to.
void ArrowScan::EnqueueLz4(Slot& s) {
  const int64_t t_k8 = trace ? TraceNow() : 0;
  const DecodedBatch& b = s.batch;
  const DeferredLz4Body& d = *b.deferred;
  if (!s.lz4_stream) {
    // the slots share kLz4Streams streams (slot i uses stream i mod 3): the K8 kernels of neighbouring record batches overlap
    // -- the token walk is latency-bound and leaves the chip idle -- without every slot holding a hardware queue of its own
    // (one stream: 0.65 s for SF10, two 0.42, three 0.39, one per slot (8) 0.46)
    // ZSTD: the entropy stage is one serial chain per block (milliseconds, a few lanes busy): more batches side by side
    // With hardware queues to spare -- GPU_MAX_HW_QUEUES, which the HIP runtime reads at start-up (default 4), raised by
    // the deployment to at least slots + 3 -- every slot gets a stream of its own: 0.24 s instead of 0.29 s at 8 slots
    // and 20 queues (profiles/r02/lz4/streams_ab.txt); on the default 4 queues the same choice was the 0.46 s above.
    int kLz4Streams = d.codec == 1 ? 16 : 3;
    {
      const char* hwq = std::getenv("GPU_MAX_HW_QUEUES");
      const int n_slots = static_cast<int>(slots.size());
      if (hwq != nullptr && std::atoi(hwq) >= n_slots + 3) kLz4Streams = n_slots;
    }
    const int idx = static_cast<int>(&s - slots.data());
    if (idx >= kLz4Streams) {
      Slot& owner = slots[static_cast<size_t>(idx % kLz4Streams)];
      if (!owner.lz4_stream) MI_HIP_CHECK(hipStreamCreateWithFlags(&owner.lz4_stream, hipStreamNonBlocking));
      s.lz4_stream = owner.lz4_stream;
      s.lz4_stream_shared = true;
    } else {
      MI_HIP_CHECK(hipStreamCreateWithFlags(&s.lz4_stream, hipStreamNonBlocking));
    }
    MI_HIP_CHECK(hipEventCreateWithFlags(&s.lz4_done, hipEventDisableTiming));
  }
  const size_t out_size = static_cast<size_t>(b.body_size);
  const size_t nb = d.blocks.size(), nf = d.buffers.size();
  // scratch layout
  size_t at = 0;
  auto take = [&](size_t bytes) { const size_t o = at; at += RoundUp(bytes + 16, 256); return o; };
  const size_t o_blocks = take(nb * sizeof(device::Lz4BlockDev)), o_buffers = take(nf * sizeof(device::Lz4BufferDev));
  const bool is_zstd = d.codec == 1;
  const size_t o_zblocks = take(is_zstd ? nb * sizeof(zstd::BlockInfo) : 0);
  const size_t tables_bytes = at;
  uint64_t total_seq = 0, max_len = 0;
  uint32_t max_blocks = 0;
  for (auto& blk : d.blocks) total_seq += is_zstd ? blk.seq_cap : device::Lz4SeqCapacity(blk.comp_size);
  // ZSTD: the decoded literals of every block lie behind the compressed body, in the same allocation
  const size_t lit_base = RoundUp(static_cast<size_t>(d.comp_size) + 64, 256);
  const size_t comp_need = is_zstd ? lit_base + d.literal_scratch + 64 : static_cast<size_t>(d.comp_size) + 64;
  const size_t o_bsize = take(nb * 4), o_bnseq = take(nb * 4), o_bbase = take(nb * 8), o_chunk = take((nb + 1) * 4), o_bufok = take(nf * 4), o_round = take(40 * 4),
               o_status = take(4), o_mark = take(out_size + 16);
  const size_t counters_end = at;
  const size_t o_seq = take(static_cast<size_t>(total_seq) * 16), o_seqoff = take(static_cast<size_t>(total_seq) * 4);
  const size_t o_lane_out = take(nb * 256 * 4), o_lane_n = take(nb * 256 * 4), o_rep = take(is_zstd ? nb * 256 * 16 : 0);
  const size_t o_link = take(out_size * 4 + 16), o_skel = take(out_size * 4 + 16);
  if (at > s.d_lz4_cap) {
    if (s.d_lz4) RetireDevice(s.d_lz4);
    s.d_lz4 = nullptr;
    s.d_lz4_cap = RoundUp(GrowCap(at, s.d_lz4_cap), 1 << 20);
    MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s.d_lz4), s.d_lz4_cap));
  }
  if (comp_need > s.d_comp_cap) {
    if (s.d_comp) RetireDevice(s.d_comp);
    s.d_comp = nullptr;
    s.d_comp_cap = RoundUp(GrowCap(comp_need, s.d_comp_cap), 1 << 16);
    MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s.d_comp), s.d_comp_cap));
  }
  if (tables_bytes > s.h_lz4_cap) {
    if (s.h_lz4) RetireHost(s.h_lz4);
    s.h_lz4 = nullptr;
    s.h_lz4_cap = RoundUp(std::max(tables_bytes, s.h_lz4_cap * 2), 1 << 16);
    MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.h_lz4), s.h_lz4_cap, hipHostMallocDefault));
  }
  auto* hb = reinterpret_cast<device::Lz4BlockDev*>(s.h_lz4 + o_blocks);
  auto* hf = reinterpret_cast<device::Lz4BufferDev*>(s.h_lz4 + o_buffers);
  uint32_t seq_at = 0;
  for (size_t i = 0; i < nb; i++) {
    const auto& blk = d.blocks[i];
    hb[i].comp_off = blk.comp_off;
    hb[i].comp_size = blk.comp_size;
    hb[i].buffer = blk.buffer;
    hb[i].stored = blk.stored;
    hb[i].seq_base = seq_at;
    hb[i].seq_cap = is_zstd ? blk.seq_cap : device::Lz4SeqCapacity(blk.comp_size);
    seq_at += hb[i].seq_cap;
  }
  if (is_zstd) {
    auto* hz = reinterpret_cast<zstd::BlockInfo*>(s.h_lz4 + o_zblocks);
    for (size_t i = 0; i < nb; i++) {
      hz[i] = d.zblocks[i];
      const bool in_scratch = hz[i].type == 1 || (hz[i].type == 2 && hz[i].lit_type != 0);
      if (in_scratch) hz[i].lit_pos += static_cast<uint32_t>(lit_base);
    }
  }
  for (size_t i = 0; i < nf; i++) {
    const auto& f = d.buffers[i];
    hf[i].out_off = static_cast<uint64_t>(f.out_off);
    // a raw buffer (length prefix -1: Arrow C++ with min_space_savings, arrow-rs, Arrow Java) has no blocks: the layout
    // kernels check "sum of the block sizes == out_len", which for it is 0 == 0; its bytes are copied below
    hf[i].out_len = f.raw ? 0 : static_cast<uint64_t>(f.out_len);
    hf[i].first_block = f.first_block;
    hf[i].n_blocks = f.raw ? 0 : f.n_blocks;
    hf[i].block_max = f.block_max;
    hf[i]._pad = 0;
    if (!f.raw) {
      max_len = std::max<uint64_t>(max_len, static_cast<uint64_t>(f.out_len));
      max_blocks = std::max<uint32_t>(max_blocks, f.n_blocks);
    }
  }
  // H2D on the copy stream: the compressed bytes of the needed buffers (neighbours closer than 64 KiB travel as one copy)
  std::vector<std::pair<int64_t, int64_t>> ranges;
  for (auto& f : d.buffers) ranges.emplace_back(f.comp_off, f.comp_len);
  std::sort(ranges.begin(), ranges.end());
  int64_t lo = -1, hi = -1;
  auto flush = [&]() {
    if (lo < 0) return;
    MI_HIP_CHECK(hipMemcpyAsync(s.d_comp + lo, d.comp + lo, static_cast<size_t>(hi - lo), hipMemcpyHostToDevice, ctx->h2d_stream));
    stats.h2d_bytes += hi - lo;
  };
  if (is_zstd) stats.zstd_batches_on_device++;
  else stats.lz4_batches_on_device++;
  stats.decompressed_bytes += b.body_size;
  for (const auto& r : ranges) {
    if (lo >= 0 && r.first <= hi + (64 << 10)) {
      hi = std::max(hi, r.first + r.second);
      continue;
    }
    flush();
    lo = r.first;
    hi = r.first + r.second;
  }
  flush();
  MI_HIP_CHECK(hipMemcpyAsync(s.d_lz4, s.h_lz4, tables_bytes, hipMemcpyHostToDevice, ctx->h2d_stream));
  MI_HIP_CHECK(hipEventRecord(s.h2d_done, ctx->h2d_stream));
  hipStream_t q = s.lz4_stream;
  MI_HIP_CHECK(hipStreamWaitEvent(q, s.h2d_done, 0));
  // ONE memset per record batch: counters, status, marks.  The link words need none (every word a stage reads was written
  // by the stage before it), nor does the body: the bytes between its buffers are padding nobody reads.
  MI_HIP_CHECK(hipMemsetAsync(s.d_lz4 + tables_bytes, 0, counters_end - tables_bytes, q));
  for (auto& f : d.buffers)
    if (f.raw && f.out_len > 0)
      MI_HIP_CHECK(hipMemcpyAsync(s.d_in + f.out_off, s.d_comp + f.comp_off, static_cast<size_t>(f.out_len), hipMemcpyDeviceToDevice, q));
  device::Lz4Args a;
  std::memset(&a, 0, sizeof(a));
  a.comp = s.d_comp;
  a.out = s.d_in;
  a.out_size = out_size;
  a.max_buffer_len = max_len;
  a.max_buffer_blocks = max_blocks;
  a.blocks = reinterpret_cast<const device::Lz4BlockDev*>(s.d_lz4 + o_blocks);
  a.buffers = reinterpret_cast<const device::Lz4BufferDev*>(s.d_lz4 + o_buffers);
  a.n_blocks = static_cast<uint32_t>(nb);
  a.n_buffers = static_cast<uint32_t>(nf);
  a.min_block_comp = 0xFFFFFFFFu;
  for (auto& blk : d.blocks)
    if (!blk.stored) {   // LZ4: which token walk fits; ZSTD: how much LDS the staged block takes
      a.max_block_comp = std::max(a.max_block_comp, blk.comp_size);
      a.min_block_comp = std::min(a.min_block_comp, blk.comp_size);
    }
  a.seq = s.d_lz4 + o_seq;
  a.seq_off = reinterpret_cast<uint32_t*>(s.d_lz4 + o_seqoff);
  a.lane_out = reinterpret_cast<uint32_t*>(s.d_lz4 + o_lane_out);
  a.lane_nseq = reinterpret_cast<uint32_t*>(s.d_lz4 + o_lane_n);
  a.link = reinterpret_cast<uint32_t*>(s.d_lz4 + o_link);
  a.block_out_size = reinterpret_cast<uint32_t*>(s.d_lz4 + o_bsize);
  a.block_nseq = reinterpret_cast<uint32_t*>(s.d_lz4 + o_bnseq);
  a.block_out_base = reinterpret_cast<uint64_t*>(s.d_lz4 + o_bbase);
  a.chunk_base = reinterpret_cast<uint32_t*>(s.d_lz4 + o_chunk);
  a.buffer_ok = reinterpret_cast<uint32_t*>(s.d_lz4 + o_bufok);
  a.round_left = reinterpret_cast<uint32_t*>(s.d_lz4 + o_round);
  a.mark = s.d_lz4 + o_mark;
  a.skel = reinterpret_cast<uint32_t*>(s.d_lz4 + o_skel);
  a.status = reinterpret_cast<uint32_t*>(s.d_lz4 + o_status);
  a.zblocks = is_zstd ? s.d_lz4 + o_zblocks : nullptr;
  a.literals = s.d_comp;
  a.rep_state = is_zstd ? reinterpret_cast<uint32_t*>(s.d_lz4 + o_rep) : nullptr;
  const int64_t t_launch = trace ? TraceNow() : 0;
  MI_HIP_CHECK(device::LaunchLz4Decompress(a, ctx->num_cus, q));
  if (trace) {
    tr_k8_prep_ns += t_launch - t_k8;
    tr_k8_launch_ns += TraceNow() - t_launch;
  }
  MI_HIP_CHECK(hipMemcpyAsync(&s.h_status[2], a.status, sizeof(uint32_t), hipMemcpyDeviceToHost, q));
  MI_HIP_CHECK(hipMemcpyAsync(&s.h_status[4], a.round_left + 37, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, q));
  s.lz4_counted = false;
  MI_HIP_CHECK(hipEventRecord(s.lz4_done, q));
  MI_HIP_CHECK(hipStreamWaitEvent(ctx->stream, s.lz4_done, 0));
  MI_HIP_CHECK(hipStreamWaitEvent(ctx->d2h_stream, s.lz4_done, 0));
}

// Stage B of a compacted batch: the filter's counts are on the host, so the projected columns get a dense layout sized
// for the rows that survived; the gather kernel decodes exactly those and only they travel back.
void ArrowScan::EnqueueStageB(Slot& s) {
  ctx->Bind();
  MI_HIP_CHECK(hipEventSynchronize(s.filter_done));
  const DecodedBatch& b = s.batch;
  Source& src = sources[static_cast<size_t>(s.source)];
  const int64_t n = b.length;
  const int64_t n_windows = (n + MI_VECTOR_SIZE - 1) / MI_VECTOR_SIZE;
  int64_t total = 0;
  for (int64_t w = 0; w < n_windows; w++) total += s.h_counts[w];
  s.needs_stage_b = false;
  // a fresh planner pass for the projected columns: arena offsets relative to the compact region behind stage A's arrays
  BatchPlanner cp(s.planner.opts);
  cp.opts.zero_copy_direct = false;
  BatchPlacement where;
  where.batch = &b;
  where.in_base = s.d_in;
  where.consumer_base = opts.device_resident ? reinterpret_cast<uint64_t>(s.d_in)
                                             : reinterpret_cast<uint64_t>(b.deferred ? s.h_mirror : b.body);
  where.alloc_rows = total;
  where.dict_len = [&](int64_t id) -> int64_t {
    auto it = dicts.find(id);
    if (it == dicts.end()) throw IOException("RecordBatch uses dictionary id " + std::to_string(id) + " before its DictionaryBatch");
    return it->second->dict_len;
  };
  s.col_root.assign(out_columns.size(), -1);
  std::vector<int32_t> widths(out_columns.size(), 0);
  for (size_t c = 0; c < out_columns.size(); c++) {
    if (out_columns[c].is_filename || out_columns[c].is_hive) continue;
    int32_t kind, nb;
    int64_t param;
    out_columns[c].field.Plan(&kind, &param, &widths[c], &nb);
    const int32_t fc = src.out_to_file_column[c];
    if (fc < 0) {
      s.absent[c] = cp.AddAbsentColumn(total, widths[c]);
      continue;
    }
    s.col_root[c] = cp.AddColumn(where, b.column_node[static_cast<size_t>(fc)]);
  }
  s.d2h_bytes = cp.arena_bytes;
  const size_t region_off = RoundUp(s.stage_a_bytes, 4096);
  if (region_off + cp.arena_bytes + 64 > s.d_out_cap) throw InternalException("compact region exceeds the slot");
  uint8_t* region = s.d_out + region_off;
  EnsureHostOut(s, s.d2h_bytes + 64);
  cp.Rebase(0, region, nullptr);
  for (auto& t : cp.tasks) {
    t.sel = s.d_out + s.sel_off;
    t.sel_count = s.d_out + s.sel_count_off;
  }
  hipStream_t st = ctx->stream;
  // validity words start as all ones (the gather kernel clears the NULLs); absent columns are all NULL
  for (size_t c = 0; c < out_columns.size(); c++) {
    if (out_columns[c].is_filename || out_columns[c].is_hive || total == 0) continue;
    if (s.col_root[c] < 0) {
      MI_HIP_CHECK(hipMemsetAsync(region + s.absent[c].first, 0, static_cast<size_t>(total) * static_cast<size_t>(std::max(widths[c], 1)), st));
      MI_HIP_CHECK(hipMemsetAsync(region + s.absent[c].second, 0, static_cast<size_t>((total + 63) / 64) * 8, st));
      continue;
    }
    const PlannedNode& pn = cp.nodes[static_cast<size_t>(s.col_root[c])];
    if (pn.valid_off >= 0) MI_HIP_CHECK(hipMemsetAsync(region + pn.valid_off, 0xFF, static_cast<size_t>((total + 63) / 64) * 8 + 8, st));
  }
  s.gather_plan->Set(cp.tasks.data(), static_cast<int32_t>(total > 0 ? cp.tasks.size() : 0), st);
  MI_HIP_CHECK(hipMemsetAsync(s.gather_plan->d_status, 0, sizeof(uint32_t), st));
  if (total > 0) s.gather_plan->Launch(st);
  MI_HIP_CHECK(hipEventRecord(s.compute_done, st));
  MI_HIP_CHECK(hipStreamWaitEvent(ctx->d2h_stream, s.compute_done, 0));
  if (!opts.device_resident && s.d2h_bytes > 0 && total > 0) {
    MI_HIP_CHECK(hipMemcpyAsync(s.h_out, region, s.d2h_bytes, hipMemcpyDeviceToHost, ctx->d2h_stream));
    stats.d2h_bytes += static_cast<int64_t>(s.d2h_bytes);
  }
  MI_HIP_CHECK(hipMemcpyAsync(&s.h_status[0], s.plan->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->d2h_stream));
  MI_HIP_CHECK(hipMemcpyAsync(&s.h_status[1], s.gather_plan->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->d2h_stream));
  MI_HIP_CHECK(hipEventRecord(s.d2h_done, ctx->d2h_stream));
  // the chunk builder reads the compact layout from here on
  s.node_dict.assign(cp.nodes.size(), nullptr);
  for (size_t i = 0; i < cp.nodes.size(); i++)
    if (cp.nodes[i].dict_id >= 0) s.node_dict[i] = dicts[cp.nodes[i].dict_id];
  s.compact_region = region;
  s.planner.nodes = std::move(cp.nodes);
}

// The vector of node `node` for chunk window `window`, children included.  Full layout: rows win[window] .. win[window + 1]
// of the node; compacted batches: rows [2048 window, 2048 window + compact_rows) of the dense arrays.
void ArrowScan::BuildVector(const Slot& s, int32_t node, size_t window, int64_t compact_rows, uint8_t* base, ChunkStorage* st, mi_vector* v) {
  const PlannedNode& o = s.planner.nodes[static_cast<size_t>(node)];
  std::memset(v, 0, sizeof(*v));
  const int64_t r0 = compact_rows >= 0 ? static_cast<int64_t>(window) * MI_VECTOR_SIZE : o.win[window];
  const int64_t r1 = compact_rows >= 0 ? r0 + compact_rows : o.win[window + 1];
  if (o.alias_body_off >= 0) {
    const uint8_t* body = opts.device_resident ? s.d_in : s.batch.body;
    v->data = const_cast<uint8_t*>(body) + o.alias_body_off + static_cast<size_t>(r0) * static_cast<size_t>(o.width);
    v->validity = nullptr;  // all valid
  } else {
    v->data = base + o.data_off + static_cast<size_t>(r0) * static_cast<size_t>(o.width);
    if (o.valid_off >= 0) {
      v->validity = reinterpret_cast<mi_validity_t*>(base + o.valid_off) + r0 / 64;
      v->validity_shift = static_cast<int32_t>(r0 % 64);
    }
  }
  v->kind = o.kind;
  v->out_width = o.width;
  v->count = r1 - r0;
  if (o.kind == MI_K_STR32 || o.kind == MI_K_STR64 || o.kind == MI_K_FIXED_BINARY) {
    v->heap = reinterpret_cast<const void*>(o.ptr_base);
    v->heap_size = o.heap_size;
  }
  if (o.kind == MI_K_DICT && s.node_dict[static_cast<size_t>(node)]) {
    const DictState& d = *s.node_dict[static_cast<size_t>(node)];
    v->dictionary = opts.device_resident ? d.d_data : d.h_data;
    v->dictionary_validity = static_cast<const mi_validity_t*>(opts.device_resident ? d.d_validity : d.h_validity);
    v->dict_len = d.dict_len;
  }
  if (!o.children.empty()) {
    if (st->child_pool_used + o.children.size() > st->child_pool.size()) throw InternalException("nested vector pool exhausted");
    mi_vector* kids = st->child_pool.data() + st->child_pool_used;
    st->child_pool_used += o.children.size();
    for (size_t k = 0; k < o.children.size(); k++) BuildVector(s, o.children[k], window, -1, base, st, &kids[k]);
    v->children = kids;
    v->n_children = static_cast<int32_t>(o.children.size());
  }
}

// per-file column mapping by name (DuckDB's multi-file column mapping) + reader projection
void ArrowScan::PrepareSource(size_t si) {
  OpenSource(si);
  Source& src = sources[si];
  if (src.prepared) return;
  const ArrowSchemaModel& schema = src.reader->GetBaseSchema();
  std::vector<std::string> names;
  for (auto& f : schema.fields) names.push_back(f.name);
  DeduplicateColumns(names);
  std::vector<std::string> wanted;
  auto map_column = [&](const ScanColumn& col) -> int32_t {
    auto it = std::find(names.begin(), names.end(), col.name);
    if (it == names.end()) {
      if (!opts.union_by_name) {
        throw InvalidInputException("Failed to read file \"" + src.path + "\": schema mismatch: column \"" + col.name +
                                    "\" is missing. If you are trying to read files with different schemas, try setting union_by_name=True");
      }
      return -1;
    }
    const ArrowField& ff = schema.fields[static_cast<size_t>(it - names.begin())];
    if (ff.Format() != col.field.Format()) {
      throw NotImplementedException("Column \"" + col.name + "\" has type " + ff.DuckType() + " in file \"" + src.path +
                                    "\" but " + col.field.DuckType() +
                                    " in the first file; cross-file casts are done by DuckDB's MultiFileReader above this path");
    }
    auto dup = std::find(wanted.begin(), wanted.end(), *it);
    if (dup != wanted.end()) return static_cast<int32_t>(dup - wanted.begin());
    wanted.push_back(*it);
    return static_cast<int32_t>(wanted.size() - 1);
  };
  src.out_to_file_column.assign(out_columns.size(), -1);
  for (size_t c = 0; c < out_columns.size(); c++) {
    if (out_columns[c].is_filename || out_columns[c].is_hive) continue;
    src.out_to_file_column[c] = map_column(out_columns[c]);
  }
  src.filter_to_file_column.assign(filter_only_columns.size(), -1);
  for (size_t c = 0; c < filter_only_columns.size(); c++) src.filter_to_file_column[c] = map_column(filter_only_columns[c]);
  if (!wanted.empty()) src.reader->SetColumnProjection(wanted);
  {
    std::lock_guard<std::mutex> lk(q_mu);   // the other producers wait for this before they open their own reader of the file
    src.wanted = wanted;
    src.prepared = true;
  }
  q_cv.notify_all();
}

// A pinned staging buffer for one record-batch body; the returned handle gives it back when the batch is released.
std::shared_ptr<void> ArrowScan::LeaseStaging(size_t bytes, uint8_t** ptr) {
  Staging* st = nullptr;
  {
    std::unique_lock<std::mutex> lk(q_mu);
    const int64_t t0 = trace ? TraceNow() : 0;
    q_cv.wait(lk, [&] {
      if (producer_stop) return true;
      for (auto& x : staging)
        if (!x.leased) return true;
      return false;
    });
    if (trace) tr_lease_wait_ns += TraceNow() - t0;
    if (producer_stop) throw IOException("scan closed while reading");
    // prefer a free buffer that is already large enough
    for (auto& x : staging)
      if (!x.leased && x.cap >= bytes + 64) { st = &x; break; }
    if (!st)
      for (auto& x : staging)
        if (!x.leased) { st = &x; break; }
    st->leased = true;
  }
  if (bytes + 64 > st->cap) {
    ctx->Bind();
    if (st->p) RetireHost(st->p);
    st->p = nullptr;
    st->cap = RoundUp(GrowCap(bytes + 64, st->cap), 1 << 16);
    MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&st->p), st->cap, hipHostMallocDefault));
  }
  *ptr = st->p;
  return std::shared_ptr<void>(st->p, [this, st](void*) {
    {
      std::lock_guard<std::mutex> lk(q_mu);
      st->leased = false;
    }
    q_cv.notify_all();
  });
}

void ArrowScan::ProducerLoop(int p) {
  size_t cap = n_producers > 1 ? 2 : static_cast<size_t>(kReadAhead);
  if (const char* v = std::getenv("MI_SCAN_READAHEAD")) cap = static_cast<size_t>(std::max(1, std::min(4, std::atoi(v))));   // fetched batches a producer holds (A/B)
  auto push = [&](Fetched&& f) {
    std::unique_lock<std::mutex> lk(q_mu);
    const int64_t t0 = trace ? TraceNow() : 0;
    q_cv.wait(lk, [&] { return producer_stop || fetched[static_cast<size_t>(p)].size() < cap; });
    if (trace) tr_push_wait_ns += TraceNow() - t0;
    if (producer_stop) return false;
    fetched[static_cast<size_t>(p)].push_back(std::move(f));
    lk.unlock();
    q_cv.notify_all();
    return true;
  };
  try {
    ctx->Bind();
    ctx->BindThisThread();   // the GPU's NUMA node: this thread's preads (and the I/O pool's, for it) and its pinned staging buffers
    size_t si = 0;
    int64_t ordinal = 0, share = 0;   // record batches of the file list; of those, this scan's (rank / world)
    while (si < sources.size()) {
      {
        std::lock_guard<std::mutex> lk(q_mu);
        if (producer_stop) return;
      }
      IPCStreamReader* reader = nullptr;
      if (p == 0) {
        PrepareSource(si);
        reader = sources[si].reader.get();
      } else {
        // a reader of its own over the same file, with the projection producer 0 settled on
        auto& mine = extra_readers[static_cast<size_t>(p - 1)];
        if (mine.size() <= si) mine.resize(sources.size());
        if (!mine[si]) {
          std::vector<std::string> wanted;
          {
            std::unique_lock<std::mutex> lk(q_mu);
            q_cv.wait(lk, [&] { return producer_stop || producer_error || sources[si].prepared; });
            if (producer_stop) return;
            if (producer_error) std::rethrow_exception(producer_error);   // producer 0 could not prepare the file: same error here
            wanted = sources[si].wanted;
          }
          mine[si] = std::make_unique<IPCFileStreamReader>(sources[si].path);
          mine[si]->SetDeferLz4(opts.host_decompress < 0 || (opts.host_decompress == 0 && opts.device_resident != 0));
          mine[si]->SetDeferZstd(DeferZstd(opts));
          mine[si]->GetBaseSchema();
          if (!wanted.empty()) mine[si]->SetColumnProjection(wanted);
        }
        reader = mine[si].get();
      }
      reader->SetBodyAllocator([this](size_t bytes, MessageType type, uint8_t** ptr) -> std::shared_ptr<void> {
        if (type == MessageType::DICTIONARY_BATCH) {  // lives as long as the dictionary version that points into it
          ctx->Bind();
          void* q = nullptr;
          MI_HIP_CHECK(hipHostMalloc(&q, bytes + 64, hipHostMallocDefault));
          *ptr = static_cast<uint8_t*>(q);
          return std::shared_ptr<void>(q, [](void* x) { (void)hipHostFree(x); });
        }
        return LeaseStaging(bytes, ptr);
      });
      Fetched f;
      const bool in_share = opts.world <= 1 || (ordinal % opts.world) == opts.rank;
      const bool mine = in_share && (share % n_producers) == p;
      const int64_t t_read = trace ? TraceNow() : 0;
      const bool got = reader->GetNextBatch(&f.batch, opts.accept_dictionaries != 0, /*skip_body*/ !mine);
      if (trace) tr_read_ns += TraceNow() - t_read;
      reader->ReleaseCurrentBody();  // the lease belongs to the batch alone
      if (!got) {
        si++;
        continue;
      }
      f.source = static_cast<int32_t>(si);
      if (!f.batch.is_dictionary) {
        f.ordinal = ordinal++;
        if (in_share) share++;
        if (!mine) continue;
      }
      if (!push(std::move(f))) return;
    }
    Fetched end;
    end.end = true;
    push(std::move(end));
  } catch (...) {
    Fetched err;
    err.error = std::current_exception();
    {
      std::lock_guard<std::mutex> lk(q_mu);   // producers waiting for this one (a file it was to prepare) fail with it
      if (!producer_error) producer_error = err.error;
    }
    q_cv.notify_all();
    push(std::move(err));
  }
}

void ArrowScan::StartProducer() {
  if (producer_started) return;
  producer_started = true;
  trace = std::getenv("MI_SCAN_TRACE") != nullptr;
  // several producers only where record batches are independent of what came before them in the stream (no dictionary
  // batches, which every later batch of the file depends on) and where there is a pread to overlap (files, not caller buffers)
  // How many: ONE when the bodies only have to be read (plain bodies, and compressed ones that are expanded in HBM) -- its preads
  // already run on the whole I/O pool, and with a CPU quota of 16 more threads only throttle one another (SF10 host consumer:
  // 0.18 s with one producer, 0.20 with three) -- THREE when the reader's host threads decompress them (a producer then spends
  // most of its time waiting for its own body's decompression: LZ4 0.29 against 0.60 s, ZSTD 0.61 against 1.27 s).  Which it
  // is shows in the first record batch's header.
  n_producers = 1;
  if (!is_buffers && !opts.accept_dictionaries) {
    int wanted = 1;
    try {
      IPCFileStreamReader peek(sources[0].path);
      peek.GetBaseSchema();
      DecodedBatch first;
      if (peek.GetNextBatch(&first, /*accept_dictionaries*/ false, /*skip_body*/ true) && first.compression >= 0) {
        const bool in_hbm = first.compression == 1 ? DeferZstd(opts) : (opts.host_decompress < 0 || (opts.host_decompress == 0 && opts.device_resident != 0));
        if (!in_hbm) wanted = 3;
      }
    } catch (...) {   // whatever is wrong with the file, the scan itself will say
    }
    const char* v = std::getenv("MI_SCAN_PRODUCERS");
    n_producers = std::max(1, std::min(kMaxProducers, v ? std::atoi(v) : wanted));
  }
  fetched.assign(static_cast<size_t>(n_producers), {});
  extra_readers.resize(static_cast<size_t>(n_producers - 1));
  next_fetch = 0;
  for (int p = 0; p < n_producers; p++) producers.emplace_back([this, p] { ProducerLoop(p); });
}

void ArrowScan::StopProducer() {
  if (!producer_started) return;
  {
    std::lock_guard<std::mutex> lk(q_mu);
    producer_stop = true;
  }
  q_cv.notify_all();
  for (auto& t : producers)
    if (t.joinable()) t.join();
}

bool ArrowScan::SubmitNextBatch(bool may_block) {
  StartProducer();
  while (!exhausted) {
    Slot* slot = FreeSlot();
    if (!slot) return false;
    Fetched f;
    {
      // in order: batch j of this scan's share comes from producer j mod P (a dictionary batch -- single producer only --
      // does not count)
      std::unique_lock<std::mutex> lk(q_mu);
      auto& q = fetched[static_cast<size_t>(next_fetch % n_producers)];
      if (q.empty()) {
        if (!may_block) return false;
        const int64_t t0 = trace ? TraceNow() : 0;
        q_cv.wait(lk, [&] { return !q.empty(); });
        if (trace) tr_fetch_wait_ns += TraceNow() - t0;
      }
      f = std::move(q.front());
      q.pop_front();
    }
    q_cv.notify_all();
    if (f.error) {
      exhausted = true;
      std::rethrow_exception(f.error);
    }
    if (f.end) {
      exhausted = true;
      return false;
    }
    cur_source = static_cast<size_t>(f.source);
    Source& src = sources[cur_source];
    if (f.batch.is_dictionary) {
      DecodeDictionary(src, f.batch);
      continue;
    }
    next_fetch++;
    if (f.batch.deferred && opts.pipeline_depth == 0) {
      // compressed bodies that are expanded in HBM spend 0.5 (LZ4) to 5 ms (ZSTD) in latency-bound kernels that leave the chip
      // nearly idle: a caller who left the depth to the scan gets as many record batches side by side as that takes
      const size_t wanted = f.batch.deferred->codec == 1 ? 16 : 8;
      if (slots.size() < wanted) {
        const size_t had = slots.size();
        std::vector<Slot> bigger(wanted);
        for (size_t i = 0; i < had; i++) bigger[i] = std::move(slots[i]);   // (only this thread touches the slots; the rest of the scan names them by index)
        slots = std::move(bigger);
        if (initialized)
          for (size_t i = had; i < wanted; i++) InitSlot(slots[i]);
        slot = FreeSlot();
      }
    }
    Slot& s = *slot;
    s.batch = std::move(f.batch);
    s.source = f.source;
    s.batch_index = f.ordinal;
    s.busy = true;
    try {
      const int64_t t0 = trace ? TraceNow() : 0;
      EnqueueBatch(s);
      if (trace) {
        s.tr_enqueued_ns = TraceNow();
        tr_enqueue_ns += s.tr_enqueued_ns - t0;
        tr_inflight_sum += static_cast<int64_t>(inflight.size()) + 1;
      }
    } catch (...) {
      s.busy = false;
      s.batch = DecodedBatch();
      throw;
    }
    inflight.push_back(static_cast<int>(&s - slots.data()));
    return true;
  }
  return false;
}

bool ArrowScan::AcquireBatch(BatchRef* out) {
  if (!initialized) Init({});
  ctx->Bind();
  for (;;) {
    // keep the pipeline full (blocks for input only when nothing is in flight: the batch the caller needs next)
    while (FreeSlot() != nullptr && SubmitNextBatch(/*may_block*/ inflight.empty())) {
    }
    // compacted batches whose selected-row counts have arrived get their second stage, in batch order
    for (size_t k = 0; k < inflight.size(); k++) {
      Slot& s = slots[static_cast<size_t>(inflight[k])];
      if (!s.needs_stage_b) continue;
      if (k == 0 || hipEventQuery(s.filter_done) == hipSuccess) EnqueueStageB(s);
      else break;
    }
    if (inflight.empty()) return false;  // exhausted
    Slot& front = slots[static_cast<size_t>(inflight.front())];
    // Wait for the batch the caller needs.  While slots are free and the input is not exhausted, keep an eye on the producers
    // instead of sleeping in the event: a scan whose record batches spend milliseconds on the GPU (compressed bodies in HBM,
    // 16 slots) otherwise submits only what the producers had ready at the moment of this call -- their queues hold a few
    // batches -- and then waits a whole batch time with most slots idle.
    if (exhausted || FreeSlot() == nullptr) {
      const int64_t t0 = trace ? TraceNow() : 0;
      MI_HIP_CHECK(hipEventSynchronize(front.d2h_done));
      if (trace) tr_event_wait_ns += TraceNow() - t0;
      break;
    }
    const hipError_t q = hipEventQuery(front.d2h_done);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) MI_HIP_CHECK(q);
    const int64_t t0 = trace ? TraceNow() : 0;
    std::unique_lock<std::mutex> lk(q_mu);
    auto& ready = fetched[static_cast<size_t>(next_fetch % n_producers)];
    q_cv.wait_for(lk, std::chrono::microseconds(100), [&] { return !ready.empty(); });
    if (trace) tr_poll_ns += TraceNow() - t0;
  }
  const int si = inflight.front();
  Slot& s = slots[static_cast<size_t>(si)];
  inflight.pop_front();
  if (trace) tr_latency_ns += TraceNow() - s.tr_enqueued_ns;
  try {
    if (s.batch.deferred && !s.lz4_counted) {
      s.lz4_counted = true;
      stats.lz4_blocks += s.h_status[4];
      stats.lz4_parse_rounds += s.h_status[6];
      stats.lz4_parse_rounds_max = std::max<int64_t>(stats.lz4_parse_rounds_max, s.h_status[5]);
    }
    uint32_t dict_status = 0;   // the decode of the dictionary versions this batch uses ran in front of it on the same stream
    for (auto& nd : s.node_dict)
      if (nd && nd->h_status) dict_status |= nd->h_status[0];
    ThrowForStatus(s.h_status[0] | s.h_status[1] | s.h_status[2] | dict_status);
  } catch (...) {
    s.busy = false;
    s.batch.owner.reset();
    throw;
  }
  out->slot = si;
  out->batch_index = s.batch_index;
  out->nrows = s.nrows;
  out->source = s.source;
  out->selected = s.nrows;
  if (has_filter) {
    out->selected = 0;
    const int64_t n_windows = (s.nrows + MI_VECTOR_SIZE - 1) / MI_VECTOR_SIZE;
    for (int64_t w = 0; w < n_windows; w++) out->selected += s.h_counts[w];
  }
  out->chunk_rows = s.compact ? out->selected : s.nrows;
  out->n_windows = static_cast<int32_t>((out->chunk_rows + MI_VECTOR_SIZE - 1) / MI_VECTOR_SIZE);
  return true;
}

void ArrowScan::ReleaseBatch(const BatchRef& ref) {
  Slot& s = slots[static_cast<size_t>(ref.slot)];
  s.busy = false;
  s.batch.owner.reset();
}

void ArrowScan::EnsureHostVectors(const BatchRef& ref) {
  Slot& s = slots[static_cast<size_t>(ref.slot)];
  if (s.host_vectors || opts.device_resident || s.compact || s.d2h_bytes == 0) return;
  ctx->Bind();
  MI_HIP_CHECK(hipMemcpy(s.h_out, s.d_out, s.d2h_bytes, hipMemcpyDeviceToHost));
  stats.d2h_bytes += static_cast<int64_t>(s.d2h_bytes);
  s.host_vectors = true;
}

void ArrowScan::DeviceColumn(const BatchRef& ref, size_t c, DeviceColumnView* out) const {
  *out = DeviceColumnView();
  const Slot& s = slots[static_cast<size_t>(ref.slot)];
  if (s.compact || s.batch.deferred || c >= out_columns.size() || out_columns[c].is_filename || out_columns[c].is_hive || s.col_root[c] < 0) return;
  const PlannedNode& pn = s.planner.nodes[static_cast<size_t>(s.col_root[c])];
  if (!pn.children.empty() || pn.dict_id >= 0 || pn.source_node < 0) return;
  if (pn.alias_body_off >= 0 && !opts.device_resident) return;   // aliased into the HOST body: not in HBM at all
  const DecodedNode& dn = s.batch.nodes[static_cast<size_t>(pn.source_node)];
  out->kind = pn.kind;
  out->width = pn.width;
  out->null_count = pn.null_count;
  out->d_data = pn.alias_body_off >= 0 ? s.d_in + pn.alias_body_off : s.d_out + pn.data_off;
  out->d_validity = (pn.valid_off >= 0 && pn.null_count != 0) ? s.d_out + pn.valid_off : nullptr;
  if (pn.null_count != 0 && pn.valid_off < 0) return;
  if (pn.kind == MI_K_STR32 || pn.kind == MI_K_STR64) {
    if (dn.spans.size() < 3) return;
    out->offset_width = pn.kind == MI_K_STR64 ? 8 : 4;
    out->d_heap = s.d_in + dn.spans[2].offset;
    out->ptr_base = pn.ptr_base;
    out->h_offsets = s.batch.body + dn.spans[1].offset;
    out->h_validity = dn.spans[0].length > 0 ? s.batch.body + dn.spans[0].offset : nullptr;
  }
  out->flat = true;
}

void ArrowScan::BuildChunk(const BatchRef& ref, int32_t window, ChunkStorage* st, mi_data_chunk* out) {
  const Slot& s = slots[static_cast<size_t>(ref.slot)];
  std::memset(out, 0, sizeof(*out));
  const int64_t chunk_rows = s.compact ? ref.selected : s.nrows;
  const int64_t row0 = static_cast<int64_t>(window) * MI_VECTOR_SIZE;
  const int64_t n = std::min<int64_t>(MI_VECTOR_SIZE, chunk_rows - row0);
  if (window < 0 || n <= 0) throw InvalidInputException("chunk window outside the record batch");
  uint8_t* base = opts.device_resident ? (s.compact ? s.compact_region : s.d_out) : s.h_out;
  st->vectors.assign(out_columns.size(), mi_vector{});
  if (st->child_pool.size() < s.planner.nodes.size() + 1) st->child_pool.resize(s.planner.nodes.size() + 1);
  st->child_pool_used = 0;
  const Source& src = sources[static_cast<size_t>(s.source)];
  for (size_t c = 0; c < out_columns.size(); c++) {
    mi_vector& v = st->vectors[c];
    if (out_columns[c].is_filename || out_columns[c].is_hive) {
      // strings are kept alive in the source (path / hive map), one vector of 2048 copies per (file, column)
      const std::string& stable = out_columns[c].is_filename ? src.path : src.hive.at(out_columns[c].hive_key);
      const mi_string_t* cv;
      {
        std::lock_guard<std::mutex> lk(const_mu);
        auto& vec = const_vectors[{s.source, c}];
        if (vec.empty()) vec.assign(MI_VECTOR_SIZE, MakeHostString(stable));
        cv = vec.data();
      }
      v.data = const_cast<mi_string_t*>(cv);
      v.validity = all_valid.data();
      v.kind = MI_K_STR32;
      v.out_width = 16;
      v.count = n;
      continue;
    }
    if (s.col_root[c] >= 0) {
      BuildVector(s, s.col_root[c], static_cast<size_t>(window), s.compact ? n : -1, base, st, &v);
    } else {  // absent in this file: all NULL
      int32_t kind, w, nb;
      int64_t param;
      out_columns[c].field.Plan(&kind, &param, &w, &nb);
      v.data = base + s.absent[c].first + static_cast<size_t>(row0) * static_cast<size_t>(std::max(w, 1));
      v.validity = reinterpret_cast<mi_validity_t*>(base + s.absent[c].second) + row0 / 64;
      v.kind = kind;
      v.out_width = w;
      v.count = n;
    }
  }
  out->size = n;
  out->n_columns = static_cast<int32_t>(out_columns.size());
  out->file_index = s.source;
  out->batch_index = s.batch_index;
  out->chunk_offset = row0;
  out->columns = st->vectors.data();
  out->sel_count = n;
  out->source_rows = s.compact ? (window == 0 ? s.nrows : 0) : n;
  if (has_filter && !s.compact) {
    out->sel = reinterpret_cast<const mi_sel_t*>(base + s.sel_off) + row0;
    out->sel_count = s.h_counts[window];
  }
}

void ArrowScan::Next(mi_data_chunk* out) {
  if (!initialized) Init({});
  std::memset(out, 0, sizeof(*out));
  while (true) {
    if (have_cur && cur_window >= cur_ref.n_windows) {  // the previous chunk was the batch's last: recycle its slot
      ReleaseBatch(cur_ref);
      have_cur = false;
    }
    if (have_cur) break;
    if (!AcquireBatch(&cur_ref)) {
      out->size = 0;
      out->n_columns = static_cast<int32_t>(out_columns.size());
      return;  // exhausted
    }
    have_cur = true;
    cur_window = 0;  // an empty (or, when compacting, fully filtered) batch has no windows: the loop moves on
  }
  BuildChunk(cur_ref, cur_window, &next_storage, out);
  cur_window++;
}

void ArrowScan::Count(int64_t* rows, int64_t* selected, int64_t* chunks) {
  if (!initialized) Init({});
  int64_t r = 0, sel = 0, n = 0;
  if (have_cur) {  // mid-batch after mi_scan_next: the rest of the current batch counts chunk by chunk
    mi_data_chunk ch;
    while (cur_window < cur_ref.n_windows) {
      BuildChunk(cur_ref, cur_window++, &next_storage, &ch);
      r += ch.source_rows;
      sel += ch.sel ? ch.sel_count : ch.size;
      n++;
    }
    ReleaseBatch(cur_ref);
    have_cur = false;
  }
  BatchRef ref;
  while (AcquireBatch(&ref)) {   // whole batches: no chunk is materialised for a count
    r += ref.nrows;
    sel += ref.selected;
    n += ref.n_windows;
    ReleaseBatch(ref);
  }
  if (rows) *rows = r;
  if (selected) *selected = sel;
  if (chunks) *chunks = n;
}

void ArrowScan::SumProduct(const std::string& a, const std::string& b, const std::vector<std::string>& filter_columns_p,
                           const std::vector<int64_t>& lo, const std::vector<int64_t>& hi, mi_sum_product_result* out) {
  if (initialized) throw InvalidInputException("mi_scan_sum_product replaces mi_scan_init / mi_scan_next: call it right after bind");
  if (filter_columns_p.size() > 4) throw InvalidInputException("at most 4 range filters");
  if (has_filter) throw InvalidInputException("give the filters to mi_scan_sum_product instead of mi_scan_set_filter");
  ctx->Bind();
  // project exactly the columns the aggregate reads
  std::vector<std::string> proj;
  auto slot_of = [&](const std::string& name) {
    for (size_t i = 0; i < proj.size(); i++)
      if (proj[i] == name) return static_cast<int32_t>(i);
    proj.push_back(name);
    return static_cast<int32_t>(proj.size() - 1);
  };
  agg.col_a = slot_of(a);
  agg.col_b = slot_of(b);
  agg.filter_cols.clear();
  for (auto& f : filter_columns_p) agg.filter_cols.push_back(slot_of(f));
  agg.lo = lo;
  agg.hi = hi;
  Init(proj);
  for (auto& name : proj) {
    const ScanColumn& c = out_columns[static_cast<size_t>(slot_of(name))];
    int32_t kind, w, nb;
    int64_t param;
    const bool ok = !c.is_filename && !c.is_hive && c.field.Plan(&kind, &param, &w, &nb) &&
                    (kind == MI_K_COPY || kind == MI_K_DEC128 || kind == MI_K_DATE64 || kind == MI_K_MUL_I32 || kind == MI_K_MUL_I64 ||
                     kind == MI_K_DIV_I64 || kind == MI_K_NARROW) &&
                    (w == 1 || w == 2 || w == 4 || w == 8) && c.field.type != MI_AT_FLOAT;
    if (!ok) throw InvalidInputException("Column '" + name + "' (" + c.field.DuckType() + ") is not a fixed-width integer-like column: the fused aggregate takes integers, DATE, TIME/TIMESTAMP and DECIMAL(<=18)");
  }
  MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&agg.d_acc), 4 * sizeof(unsigned long long)));
  MI_HIP_CHECK(hipMemsetAsync(agg.d_acc, 0, 4 * sizeof(unsigned long long), ctx->stream));
  agg.on = true;
  agg.rows_scanned = 0;
  try {
    BatchRef ref;
    while (AcquireBatch(&ref)) ReleaseBatch(ref);   // the pull loop only recycles slots: nothing is copied back
    unsigned long long acc[4] = {0, 0, 0, 0};
    MI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    MI_HIP_CHECK(hipMemcpy(acc, agg.d_acc, sizeof(acc), hipMemcpyDeviceToHost));
    out->sum_lo = acc[0];
    out->sum_hi = static_cast<int64_t>(acc[1]);
    out->rows_selected = static_cast<int64_t>(acc[2]);
    out->rows_scanned = agg.rows_scanned;
  } catch (...) {
    (void)hipFree(agg.d_acc);
    agg.d_acc = nullptr;
    agg.on = false;
    throw;
  }
  MI_HIP_CHECK(hipFree(agg.d_acc));
  agg.d_acc = nullptr;
  agg.on = false;
}

double ArrowScan::Progress() {
  if (sources.empty()) return 100;
  double done = static_cast<double>(std::min(cur_source, sources.size()));
  if (cur_source < sources.size() && sources[cur_source].reader) done += sources[cur_source].reader->GetProgress() / 100.0;
  return std::min(100.0, 100.0 * done / static_cast<double>(sources.size()));
}

// ------------------------------------------------------------------------------------------------ multi-device
MultiDeviceScan::MultiDeviceScan(const std::vector<Context*>& ctxs, std::vector<std::string> paths, const mi_scan_options& o) {
  if (ctxs.empty()) throw InvalidInputException("mi_scan_open_files_multi needs at least one context");
  const int32_t n = static_cast<int32_t>(ctxs.size());
  const int32_t outer_world = o.world > 1 ? o.world : 1, outer_rank = o.world > 1 ? o.rank : 0;
  if (outer_rank < 0 || outer_rank >= outer_world) throw InvalidInputException("rank outside [0, world)");
  for (int32_t i = 0; i < n; i++) {
    if (!ctxs[static_cast<size_t>(i)]) throw InvalidInputException("mi_scan_open_files_multi: NULL context");
    mi_scan_options so = o;
    // batch k of the file list belongs to this scan when k mod world == rank; its j-th batch goes to context j mod n:
    // k = rank + world * j  =>  sub-scan i takes the batches with k mod (world * n) == rank + world * i
    so.world = outer_world * n;
    so.rank = outer_rank + outer_world * i;
    subs.push_back(std::make_unique<ArrowScan>(ctxs[static_cast<size_t>(i)], paths, so));
  }
  EnsureIoThreads(8 * n);  // every device reads its own record batches out of the page cache
  pending.resize(subs.size());
  have.assign(subs.size(), 0);
  done.assign(subs.size(), 0);
}

const std::vector<ScanColumn>& MultiDeviceScan::Bind() {
  for (size_t i = 1; i < subs.size(); i++) subs[i]->Bind();
  return subs[0]->Bind();
}

void MultiDeviceScan::Init(const std::vector<std::string>& projected) {
  for (auto& s : subs) s->Init(projected);
}

void MultiDeviceScan::SetFilter(FilterCnf cnf) {
  for (auto& s : subs) s->SetFilter(cnf);
}

void MultiDeviceScan::ForEachParallel(const std::function<void(size_t)>& fn) {
  std::vector<std::thread> threads;
  std::vector<std::exception_ptr> errors(subs.size());
  for (size_t i = 0; i < subs.size(); i++)
    threads.emplace_back([&, i] {
      try {
        fn(i);
      } catch (...) {
        errors[i] = std::current_exception();
      }
    });
  for (auto& t : threads) t.join();
  for (auto& e : errors)
    if (e) std::rethrow_exception(e);
}

// k-way merge on the record-batch ordinal: every sub-scan yields its own batches in ascending order, so the chunk to emit
// is the pending one with the smallest batch_index.  A pending chunk stays valid until its sub-scan is pulled again, which
// happens only after the consumer has come back for the next chunk.
void MultiDeviceScan::Next(mi_data_chunk* out) {
  if (last_emitted >= 0) {
    have[static_cast<size_t>(last_emitted)] = 0;
    last_emitted = -1;
  }
  int best = -1;
  for (size_t i = 0; i < subs.size(); i++) {
    if (!have[i] && !done[i]) {
      subs[i]->Next(&pending[i]);
      if (pending[i].size == 0) done[i] = 1;
      else have[i] = 1;
    }
    if (have[i] && (best < 0 || pending[i].batch_index < pending[static_cast<size_t>(best)].batch_index)) best = static_cast<int>(i);
  }
  if (best < 0) {
    std::memset(out, 0, sizeof(*out));
    out->n_columns = static_cast<int32_t>(subs[0]->NumOutputColumns());
    return;
  }
  *out = pending[static_cast<size_t>(best)];
  last_emitted = best;
}

void MultiDeviceScan::Count(int64_t* rows, int64_t* selected, int64_t* chunks) {
  std::vector<int64_t> r(subs.size(), 0), s(subs.size(), 0), c(subs.size(), 0);
  ForEachParallel([&](size_t i) { subs[i]->Count(&r[i], &s[i], &c[i]); });
  int64_t tr = 0, ts = 0, tc = 0;
  for (size_t i = 0; i < subs.size(); i++) {
    tr += r[i];
    ts += s[i];
    tc += c[i];
  }
  if (rows) *rows = tr;
  if (selected) *selected = ts;
  if (chunks) *chunks = tc;
}

void MultiDeviceScan::SumProduct(const std::string& a, const std::string& b, const std::vector<std::string>& filter_columns,
                                 const std::vector<int64_t>& lo, const std::vector<int64_t>& hi, mi_sum_product_result* out) {
  std::vector<mi_sum_product_result> parts(subs.size());
  for (auto& p : parts) std::memset(&p, 0, sizeof(p));
  ForEachParallel([&](size_t i) { subs[i]->SumProduct(a, b, filter_columns, lo, hi, &parts[i]); });
  unsigned __int128 sum = 0;
  std::memset(out, 0, sizeof(*out));
  for (auto& p : parts) {
    sum += (static_cast<unsigned __int128>(static_cast<uint64_t>(p.sum_hi)) << 64) | p.sum_lo;  // two's complement: wraps like the device
    out->rows_scanned += p.rows_scanned;
    out->rows_selected += p.rows_selected;
  }
  out->sum_lo = static_cast<uint64_t>(sum);
  out->sum_hi = static_cast<int64_t>(static_cast<uint64_t>(sum >> 64));
}

void ArrowScan::Stats(mi_scan_stats* out) {
  out->record_batches += stats.record_batches;
  out->lz4_batches_on_device += stats.lz4_batches_on_device;
  out->h2d_bytes += stats.h2d_bytes;
  out->decompressed_bytes += stats.decompressed_bytes;
  out->lz4_blocks += stats.lz4_blocks;
  out->lz4_parse_rounds += stats.lz4_parse_rounds;
  out->lz4_parse_rounds_max = std::max(out->lz4_parse_rounds_max, stats.lz4_parse_rounds_max);
  out->zstd_batches_on_device += stats.zstd_batches_on_device;
  out->d2h_bytes += stats.d2h_bytes;
  out->aliased_bytes += stats.aliased_bytes;
}

void MultiDeviceScan::Stats(mi_scan_stats* out) {
  for (auto& s : subs) s->Stats(out);
}

double MultiDeviceScan::Progress() {
  double p = 0;
  for (auto& s : subs) p += s->Progress();
  return p / static_cast<double>(subs.size());
}

}  // namespace miarrow

// ------------------------------------------------------------------------------------------------ C ABI
using namespace miarrow;

namespace miarrow {
Context* ContextOf(mi_ctx* c);
}

struct mi_scan {
  std::unique_ptr<ScanBase> scan;
  ArrowScan* single = nullptr;  // the scan when it is not a multi-device one (the COPY pump pulls whole batches from it)
};

namespace miarrow {
ArrowScan* SingleScanOf(mi_scan* s) { return s ? s->single : nullptr; }
}  // namespace miarrow

extern "C" {

int mi_scan_open_files(mi_ctx* ctx, const char* const* paths, int32_t n_paths, const mi_scan_options* opts, mi_scan** out) {
  return WrapC([&] {
    if (!ctx || !paths || n_paths <= 0 || !out) throw InvalidInputException("mi_scan_open_files: bad argument");
    mi_scan_options o;
    std::memset(&o, 0, sizeof(o));
    if (opts) o = *opts;
    std::vector<std::string> v;
    for (int32_t i = 0; i < n_paths; i++) v.emplace_back(paths[i]);
    auto s = std::make_unique<mi_scan>();
    auto scan = std::make_unique<ArrowScan>(ContextOf(ctx), std::move(v), o);
    s->single = scan.get();
    s->scan = std::move(scan);
    *out = s.release();
  });
}

int mi_scan_open_files_multi(mi_ctx* const* ctxs, int32_t n_ctxs, const char* const* paths, int32_t n_paths,
                             const mi_scan_options* opts, mi_scan** out) {
  return WrapC([&] {
    if (!ctxs || n_ctxs <= 0 || !paths || n_paths <= 0 || !out) throw InvalidInputException("mi_scan_open_files_multi: bad argument");
    mi_scan_options o;
    std::memset(&o, 0, sizeof(o));
    if (opts) o = *opts;
    std::vector<std::string> v;
    for (int32_t i = 0; i < n_paths; i++) v.emplace_back(paths[i]);
    std::vector<Context*> cs;
    for (int32_t i = 0; i < n_ctxs; i++) cs.push_back(ContextOf(ctxs[i]));
    auto s = std::make_unique<mi_scan>();
    s->scan = std::make_unique<MultiDeviceScan>(cs, std::move(v), o);
    *out = s.release();
  });
}

int mi_scan_open_buffers(mi_ctx* ctx, const mi_ipc_buffer* buffers, int32_t n_buffers, const mi_scan_options* opts, mi_scan** out) {
  return WrapC([&] {
    if (!ctx || (!buffers && n_buffers) || n_buffers < 0 || !out) throw InvalidInputException("mi_scan_open_buffers: bad argument");
    mi_scan_options o;
    std::memset(&o, 0, sizeof(o));
    if (opts) o = *opts;
    std::vector<ArrowIPCBuffer> v;
    for (int32_t i = 0; i < n_buffers; i++) v.emplace_back(buffers[i].ptr, buffers[i].size);
    auto s = std::make_unique<mi_scan>();
    auto scan = std::make_unique<ArrowScan>(ContextOf(ctx), std::move(v), o);
    s->single = scan.get();
    s->scan = std::move(scan);
    *out = s.release();
  });
}

void mi_scan_close(mi_scan* s) { delete s; }

int mi_scan_bind(mi_scan* s, mi_field* fields, int32_t cap, int32_t* n_fields) {
  return WrapC([&] {
    if (!s || !n_fields) throw InvalidInputException("mi_scan_bind: NULL argument");
    const auto& cols = s->scan->Bind();
    *n_fields = static_cast<int32_t>(cols.size());
    for (size_t i = 0; i < cols.size() && fields && static_cast<int32_t>(i) < cap; i++) {
      FillCField(cols[i].field, static_cast<int32_t>(i), &fields[i]);
      std::snprintf(fields[i].name, sizeof(fields[i].name), "%s", cols[i].name.c_str());
    }
  });
}

int mi_scan_init(mi_scan* s, const char* const* projected_names, int32_t n_projected) {
  return WrapC([&] {
    if (!s) throw InvalidInputException("mi_scan_init: NULL scan");
    std::vector<std::string> v;
    for (int32_t i = 0; i < n_projected; i++) v.emplace_back(projected_names[i]);
    s->scan->Init(v);
  });
}

int mi_scan_set_filter(mi_scan* s, const mi_filter_node* nodes, int32_t n_nodes, int32_t root) {
  return WrapC([&] {
    if (!s || !nodes) throw InvalidInputException("mi_scan_set_filter: NULL argument");
    s->scan->SetFilter(NormaliseFilter(nodes, n_nodes, root));
  });
}

int mi_scan_set_filter_range(mi_scan* s, const char* column, int64_t lo, int64_t hi) {
  return WrapC([&] {
    if (!s || !column) throw InvalidInputException("mi_scan_set_filter_range: NULL argument");
    mi_filter_node nodes[3];
    std::memset(nodes, 0, sizeof(nodes));
    nodes[0].op = MI_F_AND;
    nodes[0].first_child = 1;
    nodes[0].n_children = 2;
    nodes[1].op = MI_F_GE;
    nodes[1].column = column;
    nodes[1].value = lo;
    nodes[2].op = MI_F_LT;
    nodes[2].column = column;
    nodes[2].value = hi;
    s->scan->SetFilter(NormaliseFilter(nodes, 3, 0));
  });
}

int mi_scan_next(mi_scan* s, mi_data_chunk* out) {
  return WrapC([&] {
    if (!s || !out) throw InvalidInputException("mi_scan_next: NULL argument");
    s->scan->Next(out);
  });
}

int mi_scan_count(mi_scan* s, int64_t* rows, int64_t* selected, int64_t* chunks) {
  return WrapC([&] {
    if (!s) throw InvalidInputException("mi_scan_count: NULL argument");
    s->scan->Count(rows, selected, chunks);
  });
}

int mi_scan_sum_product(mi_scan* s, const char* column_a, const char* column_b, const mi_range_filter* filters,
                        int32_t n_filters, mi_sum_product_result* out) {
  return WrapC([&] {
    if (!s || !column_a || !column_b || !out || (n_filters > 0 && !filters)) throw InvalidInputException("mi_scan_sum_product: NULL argument");
    std::vector<std::string> cols;
    std::vector<int64_t> lo, hi;
    for (int32_t i = 0; i < n_filters; i++) {
      if (!filters[i].column) throw InvalidInputException("mi_scan_sum_product: filter without a column");
      cols.emplace_back(filters[i].column);
      lo.push_back(filters[i].lo);
      hi.push_back(filters[i].hi);
    }
    std::memset(out, 0, sizeof(*out));
    s->scan->SumProduct(column_a, column_b, cols, lo, hi, out);
  });
}

double mi_scan_progress(mi_scan* s) { return s ? s->scan->Progress() : 0; }

int mi_scan_get_stats(mi_scan* s, mi_scan_stats* out) {
  return WrapC([&] {
    if (!s || !out) throw InvalidInputException("mi_scan_get_stats: NULL argument");
    std::memset(out, 0, sizeof(*out));
    s->scan->Stats(out);
  });
}

}  // extern "C"
