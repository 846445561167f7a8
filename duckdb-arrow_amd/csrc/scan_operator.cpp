// scan_operator.cpp -- see scan_operator.hpp.
#include "scan_operator.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <functional>

namespace miarrow {

int WrapC(const std::function<void()>& f);  // c_api.cpp
struct mi_ctx_fwd;

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
}  // namespace

ArrowScan::ArrowScan(Context* ctx_p, std::vector<std::string> paths, const mi_scan_options& o) : ctx(ctx_p), opts(o) {
  if (paths.empty()) throw InvalidInputException("read_arrow needs at least one file");
  for (auto& p : paths) {
    Source s;
    s.path = p;
    sources.push_back(std::move(s));
  }
}

ArrowScan::ArrowScan(Context* ctx_p, std::vector<ArrowIPCBuffer> buffers_p, const mi_scan_options& o)
    : ctx(ctx_p), opts(o), buffers(std::move(buffers_p)), is_buffers(true) {
  Source s;
  sources.push_back(std::move(s));
}

ArrowScan::~ArrowScan() {
  StopProducer();
  try {
    ctx->Bind();
  } catch (...) {
  }
  (void)hipStreamSynchronize(ctx->h2d_stream);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipStreamSynchronize(ctx->d2h_stream);
  for (auto& s : slots) {
    s.batch = DecodedBatch();  // returns the staging lease
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_out) (void)hipFree(s.d_out);
    if (s.h_out) (void)hipHostFree(s.h_out);
    if (s.h_status) (void)hipHostFree(s.h_status);
    if (s.h_aux) (void)hipHostFree(s.h_aux);
    if (s.d_aux) (void)hipFree(s.d_aux);
    if (s.h2d_done) (void)hipEventDestroy(s.h2d_done);
    if (s.compute_done) (void)hipEventDestroy(s.compute_done);
    if (s.d2h_done) (void)hipEventDestroy(s.d2h_done);
  }
  dicts.clear();
  fetched.clear();
  for (auto& st : staging)
    if (st.p) (void)hipHostFree(st.p);
}

ArrowScan::DictState::~DictState() {
  if (d_data) (void)hipFree(d_data);
  if (d_validity) (void)hipFree(d_validity);
  if (h_data) (void)hipHostFree(h_data);
  if (h_validity) (void)hipHostFree(h_validity);
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

void ArrowScan::Init(const std::vector<std::string>& projected) {
  Bind();
  out_columns.clear();
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
  const_vectors.assign(out_columns.size(), {});
  chunk_vectors.assign(out_columns.size(), mi_vector{});
  if (has_filter) {
    filter_out_col = -1;
    for (size_t i = 0; i < out_columns.size(); i++)
      if (out_columns[i].name == filter_column) filter_out_col = static_cast<int>(i);
    if (filter_out_col < 0) throw InvalidInputException("filter column '" + filter_column + "' is not in the projection");
    int32_t kind, w, nb;
    int64_t param;
    out_columns[static_cast<size_t>(filter_out_col)].field.Plan(&kind, &param, &w, &nb);
    if ((kind != MI_K_COPY && kind != MI_K_DEC128 && kind != MI_K_DATE64) || (w != 2 && w != 4 && w != 8) ||
        out_columns[static_cast<size_t>(filter_out_col)].field.type == MI_AT_FLOAT) {
      throw NotImplementedException("range filter pushdown needs an integer / date / decimal(<=18) column");
    }
  }
  ctx->Bind();
  for (auto& s : slots) {
    if (!s.h2d_done) {
      MI_HIP_CHECK(hipEventCreateWithFlags(&s.h2d_done, hipEventDisableTiming));
      MI_HIP_CHECK(hipEventCreateWithFlags(&s.compute_done, hipEventDisableTiming));
      MI_HIP_CHECK(hipEventCreateWithFlags(&s.d2h_done, hipEventDisableTiming));
      s.plan = std::make_unique<Plan>(ctx);
      MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.h_status), 64, hipHostMallocDefault));
      *s.h_status = 0;
    }
  }
  initialized = true;
}

void ArrowScan::SetFilterRange(const std::string& column, int64_t lo, int64_t hi) {
  if (initialized) throw InvalidInputException("set the filter before mi_scan_init");
  has_filter = true;
  filter_column = column;
  filter_lo = lo;
  filter_hi = hi;
}

void ArrowScan::EnsureSlotBuffers(Slot& s, size_t in_bytes, size_t out_bytes) {
  ctx->Bind();
  auto grow = [](size_t need, size_t cap) { return std::max(need, cap + cap / 2); };
  if (in_bytes > s.d_in_cap) {
    if (s.d_in) MI_HIP_CHECK(hipFree(s.d_in));
    s.d_in = nullptr;
    s.d_in_cap = RoundUp(grow(in_bytes, s.d_in_cap), 1 << 16);
    MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s.d_in), s.d_in_cap));
  }
  if (out_bytes > s.d_out_cap) {
    if (s.d_out) MI_HIP_CHECK(hipFree(s.d_out));
    s.d_out = nullptr;
    s.d_out_cap = RoundUp(grow(out_bytes, s.d_out_cap), 1 << 16);
    MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s.d_out), s.d_out_cap));
  }
  if (!opts.device_resident && out_bytes > s.h_out_cap) {
    if (s.h_out) MI_HIP_CHECK(hipHostFree(s.h_out));
    s.h_out = nullptr;
    s.h_out_cap = RoundUp(grow(out_bytes, s.h_out_cap), 1 << 16);
    MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.h_out), s.h_out_cap, hipHostMallocDefault));
  }
}

ArrowScan::Slot* ArrowScan::FreeSlot() {
  for (auto& s : slots)
    if (!s.busy) return &s;
  return nullptr;
}

void ArrowScan::DecodeDictionary(Source& src, const DecodedBatch& b) {
  ctx->Bind();
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
  MI_HIP_CHECK(hipMalloc(&d->d_data, data_bytes));
  MI_HIP_CHECK(hipMalloc(&d->d_validity, valid_bytes));
  MI_HIP_CHECK(hipMemset(d->d_data, 0, data_bytes));
  uint8_t* heap = nullptr;
  if (b.body_size > 0) {
    void* p = nullptr;
    MI_HIP_CHECK(hipMalloc(&p, RoundUp(static_cast<size_t>(b.body_size) + 16)));
    d->d_heaps.push_back(std::shared_ptr<void>(p, [](void* q) { (void)hipFree(q); }));
    heap = static_cast<uint8_t*>(p);
    MI_HIP_CHECK(hipMemcpy(heap, b.body, static_cast<size_t>(b.body_size), hipMemcpyHostToDevice));
  }
  if (n_old > 0)
    MI_HIP_CHECK(hipMemcpy(d->d_data, old->d_data, static_cast<size_t>(n_old) * static_cast<size_t>(w), hipMemcpyDeviceToDevice));
  std::vector<uint64_t> words(valid_bytes / 8, ~0ull);
  auto set_bit = [&](int64_t i, bool v) {
    if (v) words[static_cast<size_t>(i >> 6)] |= 1ull << (i & 63);
    else words[static_cast<size_t>(i >> 6)] &= ~(1ull << (i & 63));
  };
  if (n_old > 0) {
    std::vector<uint64_t> ow(static_cast<size_t>((n_old + 63) / 64));
    MI_HIP_CHECK(hipMemcpy(ow.data(), old->d_validity, ow.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n_old; i++) set_bit(i, (ow[static_cast<size_t>(i >> 6)] >> (i & 63)) & 1);
  }
  if (n_new > 0) {
    // decode the new values into a tile-aligned scratch vector, then append
    void* scratch_data = nullptr;
    void* scratch_valid = nullptr;
    MI_HIP_CHECK(hipMalloc(&scratch_data, RoundUp(static_cast<size_t>(n_new) * static_cast<size_t>(w) + 16)));
    MI_HIP_CHECK(hipMalloc(&scratch_valid, RoundUp(static_cast<size_t>((n_new + 63) / 64) * 8 + 8)));
    std::shared_ptr<void> g1(scratch_data, [](void* q) { (void)hipFree(q); }), g2(scratch_valid, [](void* q) { (void)hipFree(q); });
    mi_col_task t;
    std::memset(&t, 0, sizeof(t));
    const mi_buffer_span* sp = &b.buffers[0];
    t.validity = sp[0].length ? heap + sp[0].offset : nullptr;
    t.buf1 = heap + sp[1].offset;
    t.buf2 = nb > 2 ? heap + sp[2].offset : nullptr;
    t.buf2_len = nb > 2 ? sp[2].length : 0;
    t.out_data = scratch_data;
    t.out_validity = scratch_valid;
    const int64_t data_off = nb > 2 ? sp[2].offset : sp[1].offset;
    t.ptr_base = opts.device_resident ? reinterpret_cast<uint64_t>(heap + data_off) : reinterpret_cast<uint64_t>(b.body + data_off);
    t.nrows = n_new;
    t.null_count = b.null_count[0];
    t.kind = kind;
    t.param = param;
    Plan plan(ctx, &t, 1);
    plan.Launch(ctx->stream);
    ThrowForStatus(plan.Status());
    MI_HIP_CHECK(hipMemcpy(static_cast<uint8_t*>(d->d_data) + static_cast<size_t>(n_old) * static_cast<size_t>(w), scratch_data,
                           static_cast<size_t>(n_new) * static_cast<size_t>(w), hipMemcpyDeviceToDevice));
    std::vector<uint64_t> nw(static_cast<size_t>((n_new + 63) / 64));
    MI_HIP_CHECK(hipMemcpy(nw.data(), scratch_valid, nw.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n_new; i++) set_bit(n_old + i, (nw[static_cast<size_t>(i >> 6)] >> (i & 63)) & 1);
  }
  set_bit(n, false);  // the extra NULL entry at index dict_len (ColumnArrowToDuckDBDictionary)
  MI_HIP_CHECK(hipMemcpy(d->d_validity, words.data(), valid_bytes, hipMemcpyHostToDevice));
  if (!opts.device_resident) {
    MI_HIP_CHECK(hipHostMalloc(&d->h_data, data_bytes, hipHostMallocDefault));
    MI_HIP_CHECK(hipHostMalloc(&d->h_validity, valid_bytes, hipHostMallocDefault));
    MI_HIP_CHECK(hipMemcpy(d->h_data, d->d_data, data_bytes, hipMemcpyDeviceToHost));
    std::memcpy(d->h_validity, words.data(), valid_bytes);
  }
  dicts[b.dict_id] = d;
}

// One field node (and, recursively, its children) of a record batch: output slots + the transcode task.
// `win` = first row of every top-level 2048-row chunk window in this node's row space (+ the end): top-level columns and
// struct children of them have win[k] = 2048k (the tiles themselves); the child of a list starts its window k at
// offsets[win[k]], the child of a fixed_size_list at size * win[k].
int32_t ArrowScan::AddNode(Slot& s, const DecodedBatch& b, int32_t ni, std::vector<int64_t> win, bool win_is_tiles,
                           int64_t parent_valid_off, int32_t parent_div, size_t* off, std::vector<mi_col_task>* tasks,
                           std::vector<uint64_t>* aux, std::vector<std::pair<size_t, size_t>>* aux_fixups) {
  const DecodedNode& nd = b.nodes[static_cast<size_t>(ni)];
  int32_t kind, w, nb;
  int64_t param;
  if (!nd.field->Plan(&kind, &param, &w, &nb, nd.value_only))
    throw NotImplementedException("Arrow type " + nd.field->Format() + " of field '" + nd.field->name + "' is not decoded by the MI355X scan path");
  const int64_t n = nd.length;
  const int32_t idx = static_cast<int32_t>(s.node_out.size());
  s.node_out.emplace_back();
  {
    Slot::NodeOut& o = s.node_out.back();
    o.kind = kind;
    o.width = w;
    o.param = param;
    o.nrows = n;
    o.arrow_type = nd.field->type;
    o.win = win;
    // reference behaviour for plain fixed-width columns: the vector aliases the Arrow buffer (DirectConversion) and an
    // array without NULLs leaves the ValidityMask unset
    if (opts.zero_copy_direct && !agg.on && kind == MI_K_COPY && nd.null_count == 0 && parent_valid_off < 0 && nd.spans.size() > 1 &&
        !(has_filter && nd.depth == 0 && ni == filter_node)) {
      o.alias = (opts.device_resident ? s.d_in : b.body) + nd.spans[1].offset;
      if (opts.device_resident) s.upload.emplace_back(nd.spans[1].offset, nd.spans[1].length);
      return idx;
    }
    o.data_off = *off;
    *off += RoundUp(static_cast<size_t>(n) * static_cast<size_t>(std::max(w, 1)) + 16);
    o.valid_off = *off;
    *off += RoundUp(static_cast<size_t>((n + 63) / 64) * 8 + 8);
  }
  for (const auto& sp : nd.spans)
    if (sp.length > 0) s.upload.emplace_back(sp.offset, sp.length);
  const size_t data_off = s.node_out[static_cast<size_t>(idx)].data_off, valid_off = s.node_out[static_cast<size_t>(idx)].valid_off;
  auto span = [&](size_t k) { return k < nd.spans.size() ? nd.spans[k] : mi_buffer_span{0, 0}; };
  auto consumer_addr = [&](int64_t body_offset) {
    return opts.device_resident ? reinterpret_cast<uint64_t>(s.d_in) + static_cast<uint64_t>(body_offset)
                                : reinterpret_cast<uint64_t>(b.body) + static_cast<uint64_t>(body_offset);
  };
  mi_col_task t;
  std::memset(&t, 0, sizeof(t));
  // out_data / out_validity / out_aux hold OFFSETS until the slot's buffers are final (EnqueueBatch rebases them)
  t.out_data = reinterpret_cast<void*>(data_off);
  t.out_validity = reinterpret_cast<void*>(valid_off);
  t.out_aux = reinterpret_cast<void*>(parent_valid_off >= 0 ? static_cast<size_t>(parent_valid_off) + 1 : 0);  // +1: 0 means none
  t.flags = parent_div;
  t.depth = nd.depth;
  t.validity = span(0).length ? s.d_in + span(0).offset : nullptr;
  t.buf1 = nd.spans.size() > 1 ? s.d_in + span(1).offset : s.d_in;
  t.nrows = n;
  t.null_count = nd.null_count;
  t.kind = kind;
  t.param = param;
  std::vector<int64_t> child_win;
  switch (kind) {
    case MI_K_STR32: case MI_K_STR64:
      t.buf2 = s.d_in + span(2).offset;
      t.buf2_len = span(2).length;
      t.ptr_base = consumer_addr(span(2).offset);
      break;
    case MI_K_FIXED_BINARY:
      t.ptr_base = consumer_addr(span(1).offset);
      break;
    case MI_K_STRVIEW: {
      const size_t at = aux->size();
      for (size_t k = 2; k < nd.spans.size(); k++) {
        aux->push_back(consumer_addr(nd.spans[k].offset));
        aux->push_back(static_cast<uint64_t>(nd.spans[k].length));
      }
      if (nd.spans.size() <= 2) { aux->push_back(0); aux->push_back(0); }
      t.buf2_len = static_cast<int64_t>(nd.spans.size() > 2 ? nd.spans.size() - 2 : 0);
      if (n > 0) aux_fixups->emplace_back(tasks->size(), at);  // an empty node gets no task (below): nothing to patch
      break;
    }
    case MI_K_DICT: {
      auto it = dicts.find(nd.field->dict_id);
      if (it == dicts.end()) throw IOException("RecordBatch uses dictionary id " + std::to_string(nd.field->dict_id) + " before its DictionaryBatch");
      s.node_out[static_cast<size_t>(idx)].dict = it->second;
      t.param2 = it->second->dict_len;
      break;
    }
    case MI_K_LIST32: case MI_K_LIST64: {
      if (nd.children.size() != 1) throw InternalException("list field without exactly one child");
      t.param = b.nodes[static_cast<size_t>(nd.children[0])].length;
      if (!win_is_tiles) {
        const size_t at = aux->size();
        for (int64_t r : win) aux->push_back(static_cast<uint64_t>(r));
        t.buf2_len = static_cast<int64_t>(win.size());
        if (n > 0) aux_fixups->emplace_back(tasks->size(), at);
      }
      // the child's windows start at offsets[win[k]] (read from the host copy of the body)
      const uint8_t* offs = b.body + span(1).offset;
      child_win.reserve(win.size());
      int64_t prev = 0;
      for (int64_t r : win) {
        int64_t v = 0;
        if (n > 0) {
          if (r < 0 || r > n) throw InternalException("Arrow IPC validation failed: list window outside the column");
          if (kind == MI_K_LIST32) { int32_t x; std::memcpy(&x, offs + 4 * r, 4); v = x; }
          else std::memcpy(&v, offs + 8 * r, 8);
          // the offsets sampled here place the child vectors of every chunk: they are checked on the host (the device
          // checks all of them, but only flags) so that no window ever points outside the child column
          if (v < prev || v > t.param)
            throw InternalException("Arrow IPC validation failed: offsets buffer is not monotonically non-decreasing or exceeds the data buffer");
          prev = v;
        }
        child_win.push_back(v);
      }
      break;
    }
    default: break;
  }
  if (n > 0) tasks->push_back(t);
  if (kind == MI_K_LIST32 || kind == MI_K_LIST64) {
    const int32_t c = AddNode(s, b, nd.children[0], child_win, false, -1, 0, off, tasks, aux, aux_fixups);
    s.node_out[static_cast<size_t>(idx)].children.push_back(c);
  } else if (kind == MI_K_STRUCT && !nd.children.empty()) {
    const bool fixed = nd.field->type == MI_AT_FIXED_LIST;
    const int64_t size = fixed ? param : 1;
    std::vector<int64_t> cw;
    for (int64_t r : win) cw.push_back(r * size);
    for (int32_t cn : nd.children) {
      const int32_t c = AddNode(s, b, cn, cw, win_is_tiles && !fixed, static_cast<int64_t>(valid_off), static_cast<int32_t>(fixed ? size : 1),
                                off, tasks, aux, aux_fixups);
      s.node_out[static_cast<size_t>(idx)].children.push_back(c);
    }
  }
  return idx;
}

void ArrowScan::EnqueueBatch(Slot& s) {
  ctx->Bind();
  const DecodedBatch& b = s.batch;
  Source& src = sources[static_cast<size_t>(s.source)];
  const int64_t n = b.length;
  s.nrows = n;
  std::vector<int64_t> top_win;
  for (int64_t r = 0; r < n; r += MI_VECTOR_SIZE) top_win.push_back(r);
  top_win.push_back(n);
  if (n == 0) top_win.push_back(0);
  // output layout + tasks (offsets first, rebased once the slot buffers are sized)
  size_t off = 0;
  s.node_out.clear();
  s.col_root.assign(out_columns.size(), -1);
  s.col_data_off.assign(out_columns.size(), 0);
  s.col_valid_off.assign(out_columns.size(), 0);
  std::vector<int32_t> widths(out_columns.size(), 0);
  std::vector<mi_col_task> tasks;
  std::vector<uint64_t> aux;
  std::vector<std::pair<size_t, size_t>> aux_fixups;  // (task index, first aux word)
  s.upload.clear();
  filter_node = -1;
  if (has_filter && src.out_to_file_column[static_cast<size_t>(filter_out_col)] >= 0)
    filter_node = b.column_node[static_cast<size_t>(src.out_to_file_column[static_cast<size_t>(filter_out_col)])];
  // d_in must be final before tasks take addresses inside it
  EnsureSlotBuffers(s, static_cast<size_t>(b.body_size) + 64, 0);
  for (size_t c = 0; c < out_columns.size(); c++) {
    if (out_columns[c].is_filename || out_columns[c].is_hive) continue;
    const int32_t fc = src.out_to_file_column[c];
    int32_t kind, nb;
    int64_t param;
    out_columns[c].field.Plan(&kind, &param, &widths[c], &nb);
    if (fc < 0) {  // column absent in this file (union_by_name): an all-NULL vector
      s.col_data_off[c] = off;
      off += RoundUp(static_cast<size_t>(n) * static_cast<size_t>(std::max(widths[c], 1)) + 16);
      s.col_valid_off[c] = off;
      off += RoundUp(static_cast<size_t>((n + 63) / 64) * 8 + 8);
      continue;
    }
    s.col_root[c] = AddNode(s, b, b.column_node[static_cast<size_t>(fc)], top_win, true, -1, 0, &off, &tasks, &aux, &aux_fixups);
    s.col_data_off[c] = s.node_out[static_cast<size_t>(s.col_root[c])].data_off;
    s.col_valid_off[c] = s.node_out[static_cast<size_t>(s.col_root[c])].valid_off;
  }
  if (has_filter) {
    s.sel_off = off;
    off += RoundUp(static_cast<size_t>(n) * 4 + 16);
    s.sel_count_off = off;
    off += RoundUp(static_cast<size_t>((n + MI_VECTOR_SIZE - 1) / MI_VECTOR_SIZE) * 4 + 16);
  }
  EnsureSlotBuffers(s, static_cast<size_t>(b.body_size) + 64, off + 64);
  // small tables (list windows, string-view buffers) ride in a pinned aux buffer
  const size_t aux_bytes = aux.size() * 8;
  if (aux_bytes) {
    if (aux_bytes > s.h_aux_cap) {
      if (s.h_aux) MI_HIP_CHECK(hipHostFree(s.h_aux));
      if (s.d_aux) MI_HIP_CHECK(hipFree(s.d_aux));
      s.h_aux_cap = s.d_aux_cap = RoundUp(aux_bytes * 2, 4096);
      MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.h_aux), s.h_aux_cap, hipHostMallocDefault));
      MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s.d_aux), s.d_aux_cap));
    }
    std::memcpy(s.h_aux, aux.data(), aux_bytes);
  }
  for (auto& t : tasks) {
    t.out_data = s.d_out + reinterpret_cast<size_t>(t.out_data);
    t.out_validity = s.d_out + reinterpret_cast<size_t>(t.out_validity);
    const size_t pv = reinterpret_cast<size_t>(t.out_aux);
    t.out_aux = pv ? s.d_out + (pv - 1) : nullptr;
  }
  for (auto& fx : aux_fixups) tasks[fx.first].buf2 = s.d_aux + fx.second * 8;
  // H2D of the body (+ tables) on the copy stream
  if (b.body_size > 0) {
    {
      // only what the kernels read (projected columns; with zero_copy_direct not even all of those): merge the buffer
      // ranges, gaps below 64 KiB are cheaper to copy than to split.  A full scan is one copy of the whole body.
      std::sort(s.upload.begin(), s.upload.end());
      int64_t lo = -1, hi = -1;
      auto flush = [&]() {
        if (lo < 0) return;
        hi = std::min<int64_t>((hi + 63) & ~int64_t(63), b.body_size);
        MI_HIP_CHECK(hipMemcpyAsync(s.d_in + lo, b.body + lo, static_cast<size_t>(hi - lo), hipMemcpyHostToDevice, ctx->h2d_stream));
      };
      for (const auto& r : s.upload) {
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
  if (aux_bytes) MI_HIP_CHECK(hipMemcpyAsync(s.d_aux, s.h_aux, aux_bytes, hipMemcpyHostToDevice, ctx->h2d_stream));
  MI_HIP_CHECK(hipEventRecord(s.h2d_done, ctx->h2d_stream));
  MI_HIP_CHECK(hipStreamWaitEvent(ctx->stream, s.h2d_done, 0));
  // absent columns: all-NULL vectors (data 0, validity 0)
  for (size_t c = 0; c < out_columns.size(); c++) {
    if (out_columns[c].is_filename || out_columns[c].is_hive) continue;
    if (src.out_to_file_column[c] < 0 && n > 0) {
      MI_HIP_CHECK(hipMemsetAsync(s.d_out + s.col_data_off[c], 0, static_cast<size_t>(n) * static_cast<size_t>(std::max(widths[c], 1)), ctx->stream));
      MI_HIP_CHECK(hipMemsetAsync(s.d_out + s.col_valid_off[c], 0, static_cast<size_t>((n + 63) / 64) * 8, ctx->stream));
    }
  }
  s.plan->Set(tasks.data(), static_cast<int32_t>(tasks.size()), ctx->stream);
  MI_HIP_CHECK(hipMemsetAsync(s.plan->d_status, 0, sizeof(uint32_t), ctx->stream));
  s.plan->Launch(ctx->stream);
  if (has_filter && n > 0) {
    const size_t fc = static_cast<size_t>(filter_out_col);
    MI_HIP_CHECK(device::LaunchFilterRange(s.d_out + s.col_data_off[fc], widths[fc], s.d_out + s.col_valid_off[fc], n, filter_lo,
                                           filter_hi, reinterpret_cast<mi_sel_t*>(s.d_out + s.sel_off),
                                           reinterpret_cast<uint32_t*>(s.d_out + s.sel_count_off), ctx->stream));
  }
  if (agg.on && n > 0) {
    // fused consumer: the decoded vectors are read once more by the aggregate kernel and never leave HBM
    device::AggSumProductArgs a;
    std::memset(&a, 0, sizeof(a));
    auto column = [&](int32_t c, const void** data, const uint64_t** valid, int32_t* width) {
      if (s.col_root[static_cast<size_t>(c)] < 0) throw InvalidInputException("aggregate column '" + out_columns[static_cast<size_t>(c)].name + "' is absent from a file of the scan");
      const Slot::NodeOut& o = s.node_out[static_cast<size_t>(s.col_root[static_cast<size_t>(c)])];
      *data = s.d_out + o.data_off;
      *valid = reinterpret_cast<const uint64_t*>(s.d_out + o.valid_off);
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
  if (!opts.device_resident && !agg.on && off > 0)
    MI_HIP_CHECK(hipMemcpyAsync(s.h_out, s.d_out, off, hipMemcpyDeviceToHost, ctx->d2h_stream));
  // the device status word travels with the results instead of costing a stream-wide synchronisation
  MI_HIP_CHECK(hipMemcpyAsync(s.h_status, s.plan->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->d2h_stream));
  MI_HIP_CHECK(hipEventRecord(s.d2h_done, ctx->d2h_stream));
}

// The vector of node `node` for chunk window `window` (rows win[window] .. win[window + 1] of the node), children included.
void ArrowScan::BuildVector(Slot& s, int32_t node, size_t window, uint8_t* base, mi_vector* v) {
  const Slot::NodeOut& o = s.node_out[static_cast<size_t>(node)];
  std::memset(v, 0, sizeof(*v));
  const int64_t r0 = o.win[window], r1 = o.win[window + 1];
  if (o.alias) {
    v->data = const_cast<uint8_t*>(o.alias) + static_cast<size_t>(r0) * static_cast<size_t>(o.width);
    v->validity = nullptr;  // all valid
  } else {
    v->data = base + o.data_off + static_cast<size_t>(r0) * static_cast<size_t>(o.width);
    v->validity = reinterpret_cast<mi_validity_t*>(base + o.valid_off) + r0 / 64;
    v->validity_shift = static_cast<int32_t>(r0 % 64);
  }
  v->kind = o.kind;
  v->out_width = o.width;
  v->count = r1 - r0;
  if (o.kind == MI_K_DICT && o.dict) {
    v->dictionary = opts.device_resident ? o.dict->d_data : o.dict->h_data;
    v->dictionary_validity = static_cast<const mi_validity_t*>(opts.device_resident ? o.dict->d_validity : o.dict->h_validity);
    v->dict_len = o.dict->dict_len;
  }
  if (!o.children.empty()) {
    if (child_pool_used + o.children.size() > child_pool.size()) throw InternalException("nested vector pool exhausted");
    mi_vector* kids = child_pool.data() + child_pool_used;
    child_pool_used += o.children.size();
    for (size_t k = 0; k < o.children.size(); k++) BuildVector(s, o.children[k], window, base, &kids[k]);
    v->children = kids;
    v->n_children = static_cast<int32_t>(o.children.size());
  }
}

// per-file column mapping by name (DuckDB's multi-file column mapping) + reader projection
void ArrowScan::PrepareSource(size_t si) {
  OpenSource(si);
  Source& src = sources[si];
  if (!src.out_to_file_column.empty()) return;
  const ArrowSchemaModel& schema = src.reader->GetBaseSchema();
  std::vector<std::string> names;
  for (auto& f : schema.fields) names.push_back(f.name);
  DeduplicateColumns(names);
  std::vector<std::string> wanted;
  src.out_to_file_column.assign(out_columns.size(), -1);
  for (size_t c = 0; c < out_columns.size(); c++) {
    if (out_columns[c].is_filename || out_columns[c].is_hive) continue;
    auto it = std::find(names.begin(), names.end(), out_columns[c].name);
    if (it == names.end()) {
      if (!opts.union_by_name) {
        throw InvalidInputException("Failed to read file \"" + src.path + "\": schema mismatch: column \"" + out_columns[c].name +
                                    "\" is missing. If you are trying to read files with different schemas, try setting union_by_name=True");
      }
      continue;
    }
    const ArrowField& ff = schema.fields[static_cast<size_t>(it - names.begin())];
    if (ff.Format() != out_columns[c].field.Format()) {
      throw NotImplementedException("Column \"" + out_columns[c].name + "\" has type " + ff.DuckType() + " in file \"" + src.path +
                                    "\" but " + out_columns[c].field.DuckType() +
                                    " in the first file; cross-file casts are done by DuckDB's MultiFileReader above this path");
    }
    src.out_to_file_column[c] = static_cast<int32_t>(wanted.size());
    wanted.push_back(*it);
  }
  if (!wanted.empty()) src.reader->SetColumnProjection(wanted);
  else src.out_to_file_column.assign(out_columns.size(), -1);
}

// A pinned staging buffer for one record-batch body; the returned handle gives it back when the batch is released.
std::shared_ptr<void> ArrowScan::LeaseStaging(size_t bytes, uint8_t** ptr) {
  Staging* st = nullptr;
  {
    std::unique_lock<std::mutex> lk(q_mu);
    q_cv.wait(lk, [&] {
      if (producer_stop) return true;
      for (auto& x : staging)
        if (!x.leased) return true;
      return false;
    });
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
    if (st->p) MI_HIP_CHECK(hipHostFree(st->p));
    st->p = nullptr;
    st->cap = RoundUp(std::max(bytes + 64, st->cap + st->cap / 2), 1 << 16);
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

void ArrowScan::ProducerLoop() {
  auto push = [&](Fetched&& f) {
    std::unique_lock<std::mutex> lk(q_mu);
    q_cv.wait(lk, [&] { return producer_stop || fetched.size() < static_cast<size_t>(kReadAhead); });
    if (producer_stop) return false;
    fetched.push_back(std::move(f));
    lk.unlock();
    q_cv.notify_all();
    return true;
  };
  try {
    ctx->Bind();
    size_t si = 0;
    int64_t ordinal = 0;
    while (si < sources.size()) {
      {
        std::lock_guard<std::mutex> lk(q_mu);
        if (producer_stop) return;
      }
      PrepareSource(si);
      Source& src = sources[si];
      src.reader->SetBodyAllocator([this](size_t bytes, MessageType type, uint8_t** ptr) -> std::shared_ptr<void> {
        if (type == MessageType::DICTIONARY_BATCH) {  // lives as long as the dictionary version that points into it
          ctx->Bind();
          void* p = nullptr;
          MI_HIP_CHECK(hipHostMalloc(&p, bytes + 64, hipHostMallocDefault));
          *ptr = static_cast<uint8_t*>(p);
          return std::shared_ptr<void>(p, [](void* q) { (void)hipHostFree(q); });
        }
        return LeaseStaging(bytes, ptr);
      });
      Fetched f;
      const bool mine = opts.world <= 1 || (ordinal % opts.world) == opts.rank;
      const bool got = src.reader->GetNextBatch(&f.batch, opts.accept_dictionaries != 0, /*skip_body*/ !mine);
      src.reader->ReleaseCurrentBody();  // the lease belongs to the batch alone
      if (!got) {
        si++;
        continue;
      }
      f.source = static_cast<int32_t>(si);
      if (!f.batch.is_dictionary) {
        f.ordinal = ordinal++;
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
    push(std::move(err));
  }
}

void ArrowScan::StartProducer() {
  if (producer_started) return;
  producer_started = true;
  producer = std::thread([this] { ProducerLoop(); });
}

void ArrowScan::StopProducer() {
  if (!producer_started) return;
  {
    std::lock_guard<std::mutex> lk(q_mu);
    producer_stop = true;
  }
  q_cv.notify_all();
  if (producer.joinable()) producer.join();
}

bool ArrowScan::SubmitNextBatch(bool may_block) {
  StartProducer();
  while (!exhausted) {
    Slot* slot = FreeSlot();
    if (!slot) return false;
    Fetched f;
    {
      std::unique_lock<std::mutex> lk(q_mu);
      if (fetched.empty()) {
        if (!may_block) return false;
        q_cv.wait(lk, [&] { return !fetched.empty(); });
      }
      f = std::move(fetched.front());
      fetched.pop_front();
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
    Slot& s = *slot;
    s.batch = std::move(f.batch);
    s.source = f.source;
    s.batch_index = f.ordinal;
    s.busy = true;
    EnqueueBatch(s);
    inflight.push_back(static_cast<int>(&s - slots));
    return true;
  }
  return false;
}

void ArrowScan::Next(mi_data_chunk* out) {
  if (!initialized) Init({});
  ctx->Bind();
  std::memset(out, 0, sizeof(*out));
  while (true) {
    // release the slot the previous chunk came from once it is fully consumed
    if (cur_slot >= 0 && cur_row >= slots[cur_slot].nrows) {
      slots[cur_slot].busy = false;
      slots[cur_slot].batch.owner.reset();
      inflight.erase(inflight.begin());
      cur_slot = -1;
      cur_row = 0;
    }
    // keep the pipeline full
    while (inflight.size() < static_cast<size_t>(kSlots) && SubmitNextBatch(/*may_block*/ inflight.empty())) {
    }
    if (cur_slot < 0) {
      if (inflight.empty()) {
        out->size = 0;
        out->n_columns = static_cast<int32_t>(out_columns.size());
        return;  // exhausted
      }
      cur_slot = inflight.front();
      cur_row = 0;
      Slot& s = slots[cur_slot];
      MI_HIP_CHECK(hipEventSynchronize(s.d2h_done));
      ThrowForStatus(*s.h_status);
      if (s.nrows == 0) continue;  // empty record batch: nothing to emit
    }
    break;
  }
  Slot& s = slots[cur_slot];
  const int64_t n = std::min<int64_t>(MI_VECTOR_SIZE, s.nrows - cur_row);
  uint8_t* base = opts.device_resident ? s.d_out : s.h_out;
  if (child_pool.size() < s.node_out.size() + 1) child_pool.resize(s.node_out.size() + 1);
  child_pool_used = 0;
  Source& src = sources[static_cast<size_t>(s.source)];
  for (size_t c = 0; c < out_columns.size(); c++) {
    mi_vector& v = chunk_vectors[c];
    std::memset(&v, 0, sizeof(v));
    if (out_columns[c].is_filename || out_columns[c].is_hive) {
      auto& cv = const_vectors[c];
      // strings are kept alive in the source (path / hive map), the vector points at them
      const std::string& stable = out_columns[c].is_filename ? src.path : src.hive[out_columns[c].hive_key];
      if (cv.size() != MI_VECTOR_SIZE || cv[0].value.inlined.length != stable.size() ||
          (stable.size() > 12 && cv[0].value.pointer.ptr != reinterpret_cast<uint64_t>(stable.data())) ||
          (stable.size() <= 12 && std::memcmp(cv[0].value.inlined.inlined, stable.data(), stable.size()) != 0)) {
        cv.assign(MI_VECTOR_SIZE, MakeHostString(stable));
      }
      v.data = cv.data();
      v.validity = all_valid.data();
      v.kind = MI_K_STR32;
      v.out_width = 16;
      v.count = n;
      continue;
    }
    if (s.col_root[c] >= 0) {
      BuildVector(s, s.col_root[c], static_cast<size_t>(cur_row / MI_VECTOR_SIZE), base, &v);
    } else {  // absent in this file: all NULL
      int32_t kind, w, nb;
      int64_t param;
      out_columns[c].field.Plan(&kind, &param, &w, &nb);
      v.data = base + s.col_data_off[c] + static_cast<size_t>(cur_row) * static_cast<size_t>(std::max(w, 1));
      v.validity = reinterpret_cast<mi_validity_t*>(base + s.col_valid_off[c]) + cur_row / 64;
      v.kind = kind;
      v.out_width = w;
      v.count = n;
    }
  }
  out->size = n;
  out->n_columns = static_cast<int32_t>(out_columns.size());
  out->file_index = s.source;
  out->batch_index = s.batch_index;
  out->chunk_offset = cur_row;
  out->columns = chunk_vectors.data();
  out->sel_count = n;
  if (has_filter) {
    out->sel = reinterpret_cast<const mi_sel_t*>(base + s.sel_off) + cur_row;
    if (opts.device_resident) {
      uint32_t cnt = 0;
      MI_HIP_CHECK(hipMemcpy(&cnt, s.d_out + s.sel_count_off + static_cast<size_t>(cur_row / MI_VECTOR_SIZE) * 4, 4, hipMemcpyDeviceToHost));
      out->sel_count = cnt;
    } else {
      out->sel_count = reinterpret_cast<const uint32_t*>(base + s.sel_count_off)[cur_row / MI_VECTOR_SIZE];
    }
  }
  cur_row += n;
}

void ArrowScan::SumProduct(const std::string& a, const std::string& b, const std::vector<std::string>& filter_columns,
                           const std::vector<int64_t>& lo, const std::vector<int64_t>& hi, mi_sum_product_result* out) {
  if (initialized) throw InvalidInputException("mi_scan_sum_product replaces mi_scan_init / mi_scan_next: call it right after bind");
  if (filter_columns.size() > 4) throw InvalidInputException("at most 4 range filters");
  if (has_filter) throw InvalidInputException("give the filters to mi_scan_sum_product instead of mi_scan_set_filter_range");
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
  for (auto& f : filter_columns) agg.filter_cols.push_back(slot_of(f));
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
  mi_data_chunk ch;
  try {
    while (true) {   // the pull loop only recycles slots: nothing is copied back
      Next(&ch);
      if (ch.size == 0) break;
      cur_row = slots[cur_slot].nrows;  // the whole batch is consumed on the device
    }
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

}  // namespace miarrow

// ------------------------------------------------------------------------------------------------ C ABI
using namespace miarrow;

namespace miarrow {
Context* ContextOf(mi_ctx* c);
}

struct mi_scan {
  std::unique_ptr<ArrowScan> scan;
};

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
    s->scan = std::make_unique<ArrowScan>(ContextOf(ctx), std::move(v), o);
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
    s->scan = std::make_unique<ArrowScan>(ContextOf(ctx), std::move(v), o);
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

int mi_scan_set_filter_range(mi_scan* s, const char* column, int64_t lo, int64_t hi) {
  return WrapC([&] {
    if (!s || !column) throw InvalidInputException("mi_scan_set_filter_range: NULL argument");
    s->scan->SetFilterRange(column, lo, hi);
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
    int64_t r = 0, sel = 0, n = 0;
    mi_data_chunk ch;
    while (true) {
      s->scan->Next(&ch);
      if (ch.size == 0) break;
      r += ch.size;
      sel += ch.sel ? ch.sel_count : ch.size;
      n++;
    }
    if (rows) *rows = r;
    if (selected) *selected = sel;
    if (chunks) *chunks = n;
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

}  // extern "C"
