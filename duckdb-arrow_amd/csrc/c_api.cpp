// c_api.cpp -- the extern "C" boundary (include/mi_arrow_ipc.h).  Nothing throws across it: exceptions are mapped
// to errno-style codes + a thread-local message, the way IpcArrayStream::Wrap does at the reference's C stream
// boundary (src/include/ipc/array_stream.hpp:29-48).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/mi_arrow_ipc.h"
#include "engine.hpp"
#include "ipc_format.hpp"
#include "ipc_stream_reader.hpp"
#include "scan_operator.hpp"
#include "writer.hpp"

using namespace miarrow;

namespace {
thread_local std::string g_last_error;

template <typename F>
int Wrap(F&& f) {
  try {
    f();
    return MI_OK;
  } catch (const Exception& e) {
    g_last_error = e.what();
    return e.code;
  } catch (const std::bad_alloc&) {
    g_last_error = "out of memory";
    return MI_ENOMEM;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return MI_EINVAL;
  }
}
}  // namespace

struct mi_reader {
  std::unique_ptr<IPCStreamReader> reader;
  DecodedBatch batch;
  std::vector<mi_batch_node> nodes;
  std::vector<mi_buffer_span> node_spans;
  std::vector<mi_batch_index_entry> index;
};
struct mi_ctx {
  std::unique_ptr<Context> ctx;
};
namespace miarrow {
// the reader moves into an exported Arrow C stream (c_stream.cpp); the handle stays valid for mi_reader_close only
std::unique_ptr<IPCStreamReader> TakeReader(mi_reader* r) { return std::move(r->reader); }
}  // namespace miarrow
static IPCStreamReader& Live(mi_reader* r) {
  if (!r || !r->reader) throw InvalidInputException("the reader was exported as an Arrow C stream (or is NULL): only mi_reader_close is valid");
  return *r->reader;
}
struct mi_plan {
  std::unique_ptr<Plan> plan;
};

// the dominant kernel of every class, as rocprofv3 prints it
extern "C" const char* KernelNameOfClass(int32_t cls) {
  static const char* names[device::kNumClasses] = {"transcode_copy", "transcode_dec128", "transcode_string", "transcode_misc_light",
                                                  "encode_fixed", "encode_string_1p", "transcode_gather"};
  return (cls >= 0 && cls < device::kNumClasses) ? names[cls] : "";
}

extern "C" {

const char* mi_last_error(void) { return g_last_error.c_str(); }
const char* mi_version(void) { return "mi_arrow_ipc 2 gfx950 0.7.0-SNAPSHOT"; }
// nanoarrow_version() (src/nanoarrow_extension.cpp:20-31) returns the linked nanoarrow's version; the metadata
// dialect implemented here is the one of apache/arrow-nanoarrow@4bf5a932 = "0.7.0-SNAPSHOT" (test/sql/nanoarrow.test:18).
const char* mi_nanoarrow_version(void) { return "0.7.0-SNAPSHOT"; }

// ------------------------------------------------------------------------------------------------ readers
int mi_reader_open_file(const char* path, mi_reader** out) {
  return Wrap([&] {
    if (!path || !out) throw InvalidInputException("mi_reader_open_file: NULL argument");
    auto r = std::make_unique<mi_reader>();
    r->reader = std::make_unique<IPCFileStreamReader>(path);
    *out = r.release();
  });
}

int mi_reader_open_buffers(const mi_ipc_buffer* buffers, int32_t n_buffers, mi_reader** out) {
  return Wrap([&] {
    if ((!buffers && n_buffers) || !out || n_buffers < 0) throw InvalidInputException("mi_reader_open_buffers: bad argument");
    std::vector<ArrowIPCBuffer> v;
    for (int32_t i = 0; i < n_buffers; i++) v.emplace_back(buffers[i].ptr, buffers[i].size);
    auto r = std::make_unique<mi_reader>();
    r->reader = std::make_unique<IPCBufferStreamReader>(std::move(v));
    *out = r.release();
  });
}

void mi_reader_close(mi_reader* r) { delete r; }

int mi_reader_schema(mi_reader* r, mi_field* fields, int32_t cap, int32_t* n_fields) {
  return Wrap([&] {
    if (!r || !n_fields) throw InvalidInputException("mi_reader_schema: NULL argument");
    const ArrowSchemaModel& s = Live(r).GetBaseSchema();
    *n_fields = static_cast<int32_t>(s.fields.size());
    int64_t flat = 0;
    for (size_t i = 0; i < s.fields.size(); i++) {
      if (fields && static_cast<int32_t>(i) < cap) FillCField(s.fields[i], static_cast<int32_t>(flat), &fields[i]);
      flat += s.fields[i].CountFields();
    }
  });
}

int mi_reader_schema_metadata(mi_reader* r, int32_t idx, const char** key, int32_t* key_len, const char** value,
                              int32_t* value_len, int32_t* count) {
  return Wrap([&] {
    if (!r) throw InvalidInputException("mi_reader_schema_metadata: NULL reader");
    const ArrowSchemaModel& s = Live(r).GetBaseSchema();
    if (count) *count = static_cast<int32_t>(s.metadata.size());
    if (idx < 0 || static_cast<size_t>(idx) >= s.metadata.size()) {
      if (key || value) throw InvalidInputException("schema metadata index out of range");
      return;
    }
    if (key) *key = s.metadata[static_cast<size_t>(idx)].first.data();
    if (key_len) *key_len = static_cast<int32_t>(s.metadata[static_cast<size_t>(idx)].first.size());
    if (value) *value = s.metadata[static_cast<size_t>(idx)].second.data();
    if (value_len) *value_len = static_cast<int32_t>(s.metadata[static_cast<size_t>(idx)].second.size());
  });
}

int mi_reader_set_projection(mi_reader* r, const char* const* names, int32_t n) {
  return Wrap([&] {
    if (!r) throw InvalidInputException("mi_reader_set_projection: NULL reader");
    std::vector<std::string> v;
    for (int32_t i = 0; i < n; i++) v.emplace_back(names[i]);
    Live(r).SetColumnProjection(v);
  });
}

int mi_reader_next_batch(mi_reader* r, int32_t accept_dictionaries, mi_batch* out) {
  bool got = false;
  int rc = Wrap([&] {
    if (!r || !out) throw InvalidInputException("mi_reader_next_batch: NULL argument");
    got = Live(r).GetNextBatch(&r->batch, accept_dictionaries != 0);
    if (!got) return;
    const DecodedBatch& b = r->batch;
    out->length = b.length;
    out->body = b.body;
    out->body_size = b.body_size;
    out->body_file_offset = b.body_file_offset;
    out->n_columns = static_cast<int32_t>(b.column_field.size());
    out->is_dictionary = b.is_dictionary;
    out->dict_id = b.dict_id;
    out->is_delta = b.is_delta;
    out->compression = b.compression;
    out->column_field = b.column_field.data();
    out->null_count = b.null_count.data();
    out->buffers = b.buffers.data();
    r->nodes.clear();
    r->node_spans.clear();
    for (auto& nd : b.nodes) {
      mi_batch_node c;
      std::memset(&c, 0, sizeof(c));
      std::snprintf(c.name, sizeof(c.name), "%s", nd.field->name.c_str());
      c.arrow_type = nd.field->type;
      int32_t kind = 0, w = 0, nb = 0;
      int64_t param = 0;
      if (nd.field->Plan(&kind, &param, &w, &nb, nd.value_only)) {
        c.kind = kind;
        c.out_width = w;
        c.param = param;
      }
      c.parent = nd.parent;
      c.depth = nd.depth;
      c.n_children = static_cast<int32_t>(nd.children.size());
      c.first_span = static_cast<int32_t>(r->node_spans.size());
      c.n_spans = static_cast<int32_t>(nd.spans.size());
      c.length = nd.length;
      c.null_count = nd.null_count;
      r->node_spans.insert(r->node_spans.end(), nd.spans.begin(), nd.spans.end());
      r->nodes.push_back(c);
    }
    out->n_nodes = static_cast<int32_t>(r->nodes.size());
    out->nodes = r->nodes.data();
    out->node_spans = r->node_spans.data();
    out->column_node = b.column_node.data();
  });
  if (rc != MI_OK) return rc;
  return got ? MI_OK : MI_ENODATA;
}

double mi_reader_progress(mi_reader* r) { return (r && r->reader) ? r->reader->GetProgress() : 0; }

int mi_reader_index(mi_reader* r, const mi_batch_index_entry** entries, int32_t* n) {
  return Wrap([&] {
    if (!r || !entries || !n) throw InvalidInputException("mi_reader_index: NULL argument");
    const auto& idx = Live(r).BuildIndex();
    r->index.clear();
    for (auto& e : idx) r->index.push_back(mi_batch_index_entry{e.prefix_offset, e.meta_len, e.type, e.body_offset, e.body_len, e.n_rows});
    *entries = r->index.data();
    *n = static_cast<int32_t>(r->index.size());
  });
}

// ------------------------------------------------------------------------------------------------ device
// The HIP runtime multiplexes every stream of a process onto GPU_MAX_HW_QUEUES hardware queues (4 unless the variable says
// otherwise) and reads the variable ONCE, when it initialises at the process's first HIP call.  The compressed-body scans (K8)
// are sets of latency-bound kernels that want the record batches of many pipeline slots side by side, each slot on a stream of
// its own: on 4 queues their kernels queue up behind one another (ZSTD, SF10: 0.63 s against 0.34 s).  So when the
// library is loaded and nobody has chosen a value, it asks for 24 -- 16 slots + the context's three streams, and a few for
// whatever else in the process makes streams (with torch beside it the scan lost a fifth on 20) -- effective when the library
// is loaded before the process touches HIP (a DuckDB process loading the extension; bench.py and the tools import the
// package first), a no-op otherwise; an explicit GPU_MAX_HW_QUEUES in the environment always wins.
namespace {
__attribute__((constructor)) void MiRuntimeDefaults() { (void)setenv("GPU_MAX_HW_QUEUES", "24", /*overwrite*/ 0); }
}  // namespace

int mi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mi_ctx_create(int32_t device_id, mi_ctx** out) {
  return Wrap([&] {
    if (!out) throw InvalidInputException("mi_ctx_create: NULL out");
    auto c = std::make_unique<mi_ctx>();
    c->ctx = std::make_unique<Context>(device_id);
    *out = c.release();
  });
}

void mi_ctx_destroy(mi_ctx* ctx) { delete ctx; }

int mi_ctx_numa(mi_ctx* ctx, int32_t* node, char* cpulist, int32_t cap) {
  return Wrap([&] {
    if (!ctx || !node) throw InvalidInputException("mi_ctx_numa: NULL argument");
    *node = ctx->ctx->numa_node;
    if (cpulist && cap > 0) {
      const std::string& l = ctx->ctx->local_cpulist;
      const size_t n = std::min(l.size(), static_cast<size_t>(cap - 1));
      std::memcpy(cpulist, l.data(), n);
      cpulist[n] = 0;
    }
  });
}

int mi_plan_create(mi_ctx* ctx, const mi_col_task* tasks, int32_t n_tasks, mi_plan** out) {
  return Wrap([&] {
    if (!ctx || !out || n_tasks < 0 || (!tasks && n_tasks)) throw InvalidInputException("mi_plan_create: bad argument");
    auto p = std::make_unique<mi_plan>();
    p->plan = std::make_unique<Plan>(ctx->ctx.get(), tasks, n_tasks);
    *out = p.release();
  });
}

void mi_plan_destroy(mi_plan* plan) { delete plan; }

int mi_plan_launch(mi_plan* plan, void* stream) {
  return Wrap([&] {
    if (!plan) throw InvalidInputException("mi_plan_launch: NULL plan");
    plan->plan->Launch(static_cast<hipStream_t>(stream));
  });
}

int mi_plan_status(mi_plan* plan, uint32_t* status_bits) {
  return Wrap([&] {
    if (!plan || !status_bits) throw InvalidInputException("mi_plan_status: NULL argument");
    *status_bits = plan->plan->Status();
  });
}

int mi_plan_stats(const mi_plan* plan, int64_t* bytes_read, int64_t* bytes_written, int64_t* rows, int64_t* tiles) {
  return Wrap([&] {
    if (!plan) throw InvalidInputException("mi_plan_stats: NULL plan");
    if (bytes_read) *bytes_read = plan->plan->bytes_read;
    if (bytes_written) *bytes_written = plan->plan->bytes_written;
    if (rows) *rows = plan->plan->rows;
    if (tiles) *tiles = plan->plan->total_tiles;
  });
}

int mi_plan_class_stats(const mi_plan* plan, int32_t cls, int64_t* bytes_read, int64_t* bytes_written, int64_t* rows,
                        int64_t* tiles, const char** kernel_name) {
  return Wrap([&] {
    if (!plan || cls < 0 || cls >= device::kNumClasses) throw InvalidInputException("mi_plan_class_stats: bad argument");
    if (bytes_read) *bytes_read = plan->plan->class_bytes_read[cls];
    if (bytes_written) *bytes_written = plan->plan->class_bytes_written[cls];
    if (rows) *rows = plan->plan->class_rows[cls];
    if (tiles) *tiles = plan->plan->class_tiles[cls];
    if (kernel_name) *kernel_name = KernelNameOfClass(cls);
  });
}

int mi_plan_launch_timed(mi_plan* plan, void* stream, float* ms_per_class) {
  return Wrap([&] {
    if (!plan || !ms_per_class) throw InvalidInputException("mi_plan_launch_timed: NULL argument");
    plan->plan->LaunchTimed(static_cast<hipStream_t>(stream), ms_per_class);
  });
}

int mi_plan_null_counts(mi_plan* plan, int64_t* out, int32_t n_tasks) {
  return Wrap([&] {
    if (!plan || !out) throw InvalidInputException("mi_plan_null_counts: NULL argument");
    auto v = plan->plan->NullCounts(/*reset*/ true);
    if (static_cast<size_t>(n_tasks) != v.size()) throw InvalidInputException("mi_plan_null_counts: task count mismatch");
    std::memcpy(out, v.data(), v.size() * sizeof(int64_t));
  });
}

int mi_status_to_error(uint32_t status_bits) {
  return Wrap([&] { ThrowForStatus(status_bits); });
}

int mi_filter_range(mi_ctx* ctx, const void* values, int32_t width, const void* validity, int64_t nrows, int64_t lo,
                    int64_t hi, mi_sel_t* sel_out, uint32_t* count_out, void* stream) {
  return Wrap([&] {
    if (!ctx || !values || !sel_out || !count_out) throw InvalidInputException("mi_filter_range: NULL argument");
    if (width != 2 && width != 4 && width != 8) throw InvalidInputException("mi_filter_range: width must be 2, 4 or 8");
    ctx->ctx->Bind();
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->ctx->stream;
    MI_HIP_CHECK(device::LaunchFilterRange(values, width, validity, nrows, lo, hi, sel_out, count_out, s));
  });
}

}  // extern "C"

// scan operator + writer entry points live in scan_operator.cpp / writer.cpp (same extern "C" rules)
namespace miarrow {
int WrapC(const std::function<void()>& f) { return Wrap(f); }
Context* ContextOf(mi_ctx* c) { return c ? c->ctx.get() : nullptr; }
}  // namespace miarrow
