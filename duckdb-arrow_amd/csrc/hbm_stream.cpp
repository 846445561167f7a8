// hbm_stream.cpp -- a whole Arrow IPC stream resident in HBM + ONE plan that decodes every record batch of it
// (mi_hbm_* in include/mi_arrow_ipc.h).  This is SURVEY.md 8d (i), the kernel-resident mode the roofline is measured in:
// the stream's bytes are uploaded once, the host reader slices every message, the shared BatchPlanner lays out one
// DuckDB vector array per (record batch, field node) and a launch is a handful of kernels however many batches there are.
// The reference's counterpart is the per-batch loop ArrowScanFunction -> ArrowToDuckDB
// (src/scanner/scan_arrow_ipc.cpp:56, src/file_scanner/arrow_file_scan.cpp:68-72).
#include <hip/hip_runtime.h>

#include <cstring>
#include <functional>
#include <map>
#include <memory>

#include "batch_planner.hpp"
#include "engine.hpp"

namespace miarrow {

int WrapC(const std::function<void()>& f);  // c_api.cpp
Context* ContextOf(mi_ctx* c);

class HbmStream {
 public:
  HbmStream(Context* ctx_p, const uint8_t* host, int64_t size, const mi_hbm_options& o) : ctx(ctx_p), planner(MakeOptions(o)) {
    if (!host || size <= 0) throw InvalidInputException("mi_hbm_open: empty stream");
    ctx->Bind();
    // ---- host parse: every message -> field nodes with their buffer spans
    std::vector<ArrowIPCBuffer> bufs;
    bufs.emplace_back(reinterpret_cast<uint64_t>(host), static_cast<uint64_t>(size));
    IPCBufferStreamReader rd(bufs);
    if (rd.GetBaseSchema().endianness != 0)
      throw NotImplementedException("mi_hbm_open takes little-endian streams: the resident copy IS the stream (big-endian bodies are swapped by the scan operator's reader)");
    if (o.columns && o.n_columns > 0) {
      std::vector<std::string> names;
      for (int32_t i = 0; i < o.n_columns; i++) names.emplace_back(o.columns[i] ? o.columns[i] : "");
      rd.SetColumnProjection(names);
    }
    std::vector<DecodedBatch> batches;
    std::map<int64_t, DecodedBatch> dict_batches;  // the last DictionaryBatch of every id
    while (true) {
      DecodedBatch b;
      if (!rd.GetNextBatch(&b, o.accept_dictionaries != 0)) break;
      if (b.compression != -1) throw NotImplementedException("mi_hbm_open takes uncompressed streams (compressed bodies go through the scan operator)");
      if (b.is_dictionary) {
        if (b.is_delta) throw NotImplementedException("delta dictionaries are handled by the scan operator, not by the HBM-resident mode");
        // one dictionary per id for the whole resident stream: a replacement that arrives after record batches (legal in the
        // stream format) would silently re-interpret the indices of the batches before it
        if (!batches.empty() && dict_batches.count(b.dict_id))
          throw NotImplementedException("dictionary " + std::to_string(b.dict_id) + " is replaced in mid-stream: dictionary versions are handled by "
                                        "the scan operator, not by the HBM-resident mode");
        dict_batches[b.dict_id] = std::move(b);
      } else {
        batches.push_back(std::move(b));
      }
    }
    // ---- HBM: the stream itself (slack so the last buffer's padding is addressable), caller-owned or ours
    if (o.device_stream) {
      d_in = static_cast<uint8_t*>(o.device_stream);
    } else {
      MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_in), static_cast<size_t>(size) + 320));
      own_in = true;
      MI_HIP_CHECK(hipMemcpy(d_in, host, static_cast<size_t>(size), hipMemcpyHostToDevice));
    }
    stream_size = size;
    switch (o.pointer_mode) {
      case MI_HBM_PTR_DEVICE: consumer_base = reinterpret_cast<uint64_t>(d_in); break;
      case MI_HBM_PTR_STREAM_OFFSET: consumer_base = 0; break;
      case MI_HBM_PTR_HOST: consumer_base = reinterpret_cast<uint64_t>(host); break;
      default: throw InvalidInputException("mi_hbm_open: unknown pointer_mode");
    }
    // ---- layout + tasks: dictionaries first (they sit at the start of the arena), then the record batches in order
    auto dict_len = [&](int64_t id) -> int64_t {
      auto it = dict_batches.find(id);
      if (it == dict_batches.end()) throw IOException("RecordBatch uses dictionary id " + std::to_string(id) + " before its DictionaryBatch");
      return it->second.nodes[static_cast<size_t>(it->second.column_node[0])].length;
    };
    auto add_batch = [&](const DecodedBatch& b, bool is_dict) {
      mi_hbm_batch hb;
      std::memset(&hb, 0, sizeof(hb));
      BatchPlacement where;
      where.batch = &b;
      where.in_base = d_in + b.body_file_offset;
      where.consumer_base = consumer_base + static_cast<uint64_t>(b.body_file_offset);
      where.dict_len = dict_len;
      hb.nrows = b.length;
      hb.body_off = b.body_file_offset;
      hb.body_len = b.body_size;
      hb.arena_begin = static_cast<int64_t>(planner.arena_bytes);
      hb.first_node = static_cast<int32_t>(planner.nodes.size());
      hb.is_dictionary = is_dict ? 1 : 0;
      hb.dict_id = is_dict ? b.dict_id : -1;
      hb.n_columns = static_cast<int32_t>(b.column_node.size());
      for (int32_t ni : b.column_node) planner.AddColumn(where, ni, is_dict ? 1 : 0);
      hb.n_nodes = static_cast<int32_t>(planner.nodes.size()) - hb.first_node;
      hb.arena_end = static_cast<int64_t>(planner.arena_bytes);
      // C view of the new nodes
      for (int32_t k = hb.first_node; k < hb.first_node + hb.n_nodes; k++) {
        const PlannedNode& p = planner.nodes[static_cast<size_t>(k)];
        const DecodedNode& dn = b.nodes[static_cast<size_t>(p.source_node)];
        mi_hbm_node c;
        std::memset(&c, 0, sizeof(c));
        std::snprintf(c.name, sizeof(c.name), "%s", dn.field->name.c_str());
        c.kind = p.kind;
        c.out_width = p.width;
        c.arrow_type = p.arrow_type;
        c.depth = p.depth;
        c.parent = p.parent;
        c.batch = static_cast<int32_t>(c_batches.size());
        c.param = p.param;
        c.nrows = p.nrows;
        c.null_count = p.null_count;
        c.dict_id = p.dict_id;
        c.data_off = p.alias_body_off >= 0 ? -1 : static_cast<int64_t>(p.data_off);
        c.valid_off = p.valid_off;
        c.alias_off = p.alias_body_off >= 0 ? b.body_file_offset + p.alias_body_off : -1;
        c.ptr_base = p.ptr_base;
        c.first_span = static_cast<int32_t>(c_spans.size());
        c.n_spans = static_cast<int32_t>(dn.spans.size());
        for (const auto& sp : dn.spans) c_spans.push_back(mi_buffer_span{b.body_file_offset + sp.offset, sp.length});
        c.first_window = static_cast<int32_t>(c_windows.size());
        c.n_windows = static_cast<int32_t>(p.win.size());
        c_windows.insert(c_windows.end(), p.win.begin(), p.win.end());
        c_nodes.push_back(c);
      }
      c_batches.push_back(hb);
      n_rows += is_dict ? 0 : b.length;
    };
    for (auto& kv : dict_batches) add_batch(kv.second, true);
    for (auto& b : batches) add_batch(b, false);
    // ---- arena + tables
    arena_bytes = std::max<size_t>(planner.arena_bytes, 256);
    if (o.device_arena) {
      if (o.device_arena_bytes < static_cast<int64_t>(arena_bytes))
        throw InvalidInputException("mi_hbm_open: device_arena holds " + std::to_string(o.device_arena_bytes) + " bytes, the layout needs " +
                                    std::to_string(arena_bytes) + " (call with device_arena = NULL to learn the size)");
      d_out = static_cast<uint8_t*>(o.device_arena);
    } else if (!o.defer_arena) {
      AllocateArena();
    }
    if (!planner.aux.empty()) {
      MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_aux), planner.aux.size() * 8));
      MI_HIP_CHECK(hipMemcpy(d_aux, planner.aux.data(), planner.aux.size() * 8, hipMemcpyHostToDevice));
    }
    if (d_out) Finish();
  }

  ~HbmStream() {
    try { ctx->Bind(); } catch (...) {}
    plan.reset();
    if (own_in && d_in) (void)hipFree(d_in);
    if (own_out && d_out) (void)hipFree(d_out);
    if (d_aux) (void)hipFree(d_aux);
  }

  void SetArena(void* p, int64_t bytes) {
    if (plan) throw InvalidInputException("mi_hbm_set_arena: the arena is already bound");
    if (!p || bytes < static_cast<int64_t>(arena_bytes)) throw InvalidInputException("mi_hbm_set_arena: arena too small");
    d_out = static_cast<uint8_t*>(p);
    Finish();
  }

  Plan& ThePlan() {
    if (!plan) throw InvalidInputException("mi_hbm: no arena bound yet (defer_arena: call mi_hbm_set_arena first)");
    return *plan;
  }

  void Layout(mi_hbm_layout* out) const {
    std::memset(out, 0, sizeof(*out));
    out->batches = c_batches.data();
    out->n_batches = static_cast<int32_t>(c_batches.size());
    out->nodes = c_nodes.data();
    out->n_nodes = static_cast<int32_t>(c_nodes.size());
    out->spans = c_spans.data();
    out->windows = c_windows.data();
    out->arena_bytes = static_cast<int64_t>(arena_bytes);
    out->device_stream = d_in;
    out->device_arena = d_out;
    out->stream_bytes = stream_size;
    out->n_rows = n_rows;
    out->n_tasks = static_cast<int32_t>(planner.tasks.size());
  }

  void Fetch(int64_t off, int64_t len, void* dst, bool from_stream) {
    ctx->Bind();
    const int64_t limit = from_stream ? stream_size : static_cast<int64_t>(arena_bytes);
    const uint8_t* base = from_stream ? d_in : d_out;
    if (!base || !SpanInside(off, len, limit) || (!dst && len)) throw InvalidInputException("mi_hbm_fetch: range outside the buffer");
    if (len) MI_HIP_CHECK(hipMemcpy(dst, base + off, static_cast<size_t>(len), hipMemcpyDeviceToHost));
  }

  Context* ctx;

 private:
  static PlannerOptions MakeOptions(const mi_hbm_options& o) {
    PlannerOptions p;
    p.array_align = o.array_align > 0 ? static_cast<size_t>(o.array_align) : (64u << 10);
    if (p.array_align & (p.array_align - 1)) throw InvalidInputException("mi_hbm_open: array_align must be a power of two");
    p.zero_copy_direct = o.zero_copy_direct != 0;
    p.unset_all_valid = o.unset_all_valid != 0;
    return p;
  }
  void AllocateArena() {
    MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_out), arena_bytes));
    own_out = true;
    // NULL slots of dictionaries and padding between arrays read as zero
    MI_HIP_CHECK(hipMemset(d_out, 0, arena_bytes));
    MI_HIP_CHECK(hipStreamSynchronize(nullptr));   // the null stream is not ordered with the (non-blocking) streams the plans run on
  }
  void Finish() {
    planner.Rebase(0, d_out, d_aux);
    plan = std::make_unique<Plan>(ctx, planner.tasks.data(), static_cast<int32_t>(planner.tasks.size()));
  }

  BatchPlanner planner;
  std::unique_ptr<Plan> plan;
  uint8_t* d_in = nullptr;
  uint8_t* d_out = nullptr;
  uint8_t* d_aux = nullptr;
  bool own_in = false, own_out = false;
  size_t arena_bytes = 0;
  int64_t stream_size = 0, n_rows = 0;
  uint64_t consumer_base = 0;
  std::vector<mi_hbm_batch> c_batches;
  std::vector<mi_hbm_node> c_nodes;
  std::vector<mi_buffer_span> c_spans;
  std::vector<int64_t> c_windows;
};

}  // namespace miarrow

using namespace miarrow;

struct mi_hbm {
  std::unique_ptr<HbmStream> hs;
};

extern "C" {

int mi_hbm_open(mi_ctx* ctx, const void* host_stream, int64_t size, const mi_hbm_options* opts, mi_hbm** out) {
  return WrapC([&] {
    if (!ctx || !out) throw InvalidInputException("mi_hbm_open: NULL argument");
    mi_hbm_options o;
    std::memset(&o, 0, sizeof(o));
    if (opts) o = *opts;
    auto h = std::make_unique<mi_hbm>();
    h->hs = std::make_unique<HbmStream>(ContextOf(ctx), static_cast<const uint8_t*>(host_stream), size, o);
    *out = h.release();
  });
}

void mi_hbm_close(mi_hbm* h) { delete h; }

int mi_hbm_set_arena(mi_hbm* h, void* device_arena, int64_t bytes) {
  return WrapC([&] {
    if (!h) throw InvalidInputException("mi_hbm_set_arena: NULL");
    h->hs->SetArena(device_arena, bytes);
  });
}

int mi_hbm_layout_get(mi_hbm* h, mi_hbm_layout* out) {
  return WrapC([&] {
    if (!h || !out) throw InvalidInputException("mi_hbm_layout_get: NULL argument");
    h->hs->Layout(out);
  });
}

int mi_hbm_launch(mi_hbm* h, void* stream) {
  return WrapC([&] {
    if (!h) throw InvalidInputException("mi_hbm_launch: NULL");
    h->hs->ThePlan().Launch(static_cast<hipStream_t>(stream));
  });
}

int mi_hbm_launch_timed(mi_hbm* h, void* stream, float* ms_per_class) {
  return WrapC([&] {
    if (!h || !ms_per_class) throw InvalidInputException("mi_hbm_launch_timed: NULL argument");
    h->hs->ThePlan().LaunchTimed(static_cast<hipStream_t>(stream), ms_per_class);
  });
}

int mi_hbm_status(mi_hbm* h, uint32_t* status_bits) {
  return WrapC([&] {
    if (!h || !status_bits) throw InvalidInputException("mi_hbm_status: NULL argument");
    *status_bits = h->hs->ThePlan().Status();
  });
}

int mi_hbm_stats(mi_hbm* h, int64_t* bytes_read, int64_t* bytes_written, int64_t* rows, int64_t* tiles) {
  return WrapC([&] {
    if (!h) throw InvalidInputException("mi_hbm_stats: NULL");
    Plan& p = h->hs->ThePlan();
    if (bytes_read) *bytes_read = p.bytes_read;
    if (bytes_written) *bytes_written = p.bytes_written;
    if (rows) *rows = p.rows;
    if (tiles) *tiles = p.total_tiles;
  });
}

const char* KernelNameOfClass(int32_t cls);  // c_api.cpp

int mi_hbm_class_stats(mi_hbm* h, int32_t kernel_class, int64_t* bytes_read, int64_t* bytes_written, int64_t* rows,
                       int64_t* tiles, const char** kernel_name) {
  return WrapC([&] {
    if (!h || kernel_class < 0 || kernel_class >= device::kNumClasses) throw InvalidInputException("mi_hbm_class_stats: bad argument");
    Plan& p = h->hs->ThePlan();
    if (bytes_read) *bytes_read = p.class_bytes_read[kernel_class];
    if (bytes_written) *bytes_written = p.class_bytes_written[kernel_class];
    if (rows) *rows = p.class_rows[kernel_class];
    if (tiles) *tiles = p.class_tiles[kernel_class];
    if (kernel_name) *kernel_name = KernelNameOfClass(kernel_class);
  });
}

int mi_hbm_fetch(mi_hbm* h, int32_t from_stream, int64_t offset, int64_t length, void* host_dst) {
  return WrapC([&] {
    if (!h) throw InvalidInputException("mi_hbm_fetch: NULL");
    h->hs->Fetch(offset, length, host_dst, from_stream != 0);
  });
}

}  // extern "C"
