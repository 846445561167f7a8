// c_stream.cpp -- the reader as an Arrow C stream (ArrowArrayStream), the narrowest seam of the reference:
// IpcArrayStream::{GetSchema, GetNext, Wrap} (src/ipc/array_stream.cpp:11-26, src/include/ipc/array_stream.hpp:29-48)
// behind ArrowIPCStreamFactory::Produce (src/ipc/stream_factory.cpp:14-30).  Host only.  Arrays are zero-copy views of
// the message bodies the reader holds (the body is kept alive by every array that points into it, like nanoarrow's
// ArrowIpcSharedBuffer); compressed bodies are decompressed by the reader first.
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "ipc_stream_reader.hpp"

namespace miarrow {

int WrapC(const std::function<void()>& f);  // c_api.cpp
std::unique_ptr<IPCStreamReader> TakeReader(mi_reader* r);  // c_api.cpp

namespace {
constexpr int64_t kFlagDictionaryOrdered = 1, kFlagNullable = 2;

// ---- schema ------------------------------------------------------------------------------------------------------
struct SchemaPrivate {
  std::string format, name, metadata;
  std::vector<ArrowSchema> children;
  std::vector<ArrowSchema*> child_ptrs;
  std::unique_ptr<ArrowSchema> dictionary;
};

void ReleaseSchema(ArrowSchema* s) {
  if (!s || !s->release) return;
  auto* p = static_cast<SchemaPrivate*>(s->private_data);
  for (auto& c : p->children)
    if (c.release) c.release(&c);
  if (p->dictionary && p->dictionary->release) p->dictionary->release(p->dictionary.get());
  delete p;
  s->release = nullptr;
}

// Arrow C data interface metadata: int32 n, then n x (int32 key length, key, int32 value length, value)
std::string EncodeMetadata(const std::vector<std::pair<std::string, std::string>>& kv) {
  if (kv.empty()) return std::string();
  std::string out;
  auto put = [&](int32_t v) { out.append(reinterpret_cast<const char*>(&v), 4); };
  put(static_cast<int32_t>(kv.size()));
  for (auto& e : kv) {
    put(static_cast<int32_t>(e.first.size()));
    out += e.first;
    put(static_cast<int32_t>(e.second.size()));
    out += e.second;
  }
  return out;
}

void ExportField(const ArrowField& f, bool as_value, ArrowSchema* out) {
  auto* p = new SchemaPrivate();
  std::memset(out, 0, sizeof(*out));
  const bool dict = f.has_dictionary && !as_value;
  if (dict) {  // the field's format is the index type; the value type hangs off `dictionary`
    ArrowField idx;
    idx.type = MI_AT_INT;
    idx.bit_width = f.dict_index_bit_width;
    idx.is_signed = f.dict_index_signed;
    p->format = idx.Format();
    p->dictionary = std::make_unique<ArrowSchema>();
    ExportField(f, true, p->dictionary.get());
  } else {
    p->format = f.Format();
    p->children.resize(f.children.size());
    for (size_t i = 0; i < f.children.size(); i++) ExportField(f.children[i], false, &p->children[i]);
    for (auto& c : p->children) p->child_ptrs.push_back(&c);
  }
  p->name = as_value ? std::string() : f.name;
  p->metadata = as_value ? std::string() : EncodeMetadata(f.metadata);
  out->format = p->format.c_str();
  out->name = p->name.c_str();
  out->metadata = p->metadata.empty() ? nullptr : p->metadata.data();
  out->flags = (f.nullable ? kFlagNullable : 0) | (dict && f.dict_ordered ? kFlagDictionaryOrdered : 0);
  out->n_children = static_cast<int64_t>(p->children.size());
  out->children = p->child_ptrs.empty() ? nullptr : p->child_ptrs.data();
  out->dictionary = p->dictionary.get();
  out->release = ReleaseSchema;
  out->private_data = p;
}

void ExportSchema(const ArrowSchemaModel& schema, ArrowSchema* out) {
  auto* p = new SchemaPrivate();
  std::memset(out, 0, sizeof(*out));
  p->format = "+s";
  p->metadata = EncodeMetadata(schema.metadata);
  p->children.resize(schema.fields.size());
  for (size_t i = 0; i < schema.fields.size(); i++) ExportField(schema.fields[i], false, &p->children[i]);
  for (auto& c : p->children) p->child_ptrs.push_back(&c);
  out->format = p->format.c_str();
  out->name = p->name.c_str();
  out->metadata = p->metadata.empty() ? nullptr : p->metadata.data();
  out->flags = 0;
  out->n_children = static_cast<int64_t>(p->children.size());
  out->children = p->child_ptrs.empty() ? nullptr : p->child_ptrs.data();
  out->release = ReleaseSchema;
  out->private_data = p;
}

// ---- arrays ------------------------------------------------------------------------------------------------------
struct ArrayPrivate {
  std::shared_ptr<void> body;                  // keeps the message body alive
  std::shared_ptr<void> dictionary_body;
  std::vector<const void*> buffers;
  std::vector<ArrowArray> children;
  std::vector<ArrowArray*> child_ptrs;
  std::unique_ptr<ArrowArray> dictionary;
};

void ReleaseArray(ArrowArray* a) {
  if (!a || !a->release) return;
  auto* p = static_cast<ArrayPrivate*>(a->private_data);
  for (auto& c : p->children)
    if (c.release) c.release(&c);
  if (p->dictionary && p->dictionary->release) p->dictionary->release(p->dictionary.get());
  delete p;
  a->release = nullptr;
}

struct DictValues {  // the last DictionaryBatch of an id, kept for the record batches that follow it
  DecodedBatch batch;
};

// NANOARROW_VALIDATION_LEVEL_FULL's data-dependent part (base_stream_reader.cpp:112-140), which the GPU path does on the
// device: a consumer of the C stream trusts offsets, view fields and dictionary indices, so they are walked here once.
template <typename OFF>
void CheckOffsets(const DecodedBatch& b, const DecodedNode& nd, int64_t limit, const char* what) {
  if (nd.length == 0) return;
  const OFF* off = reinterpret_cast<const OFF*>(b.body + nd.spans[1].offset);
  int64_t prev = static_cast<int64_t>(off[0]);
  bool ok = prev >= 0;
  for (int64_t i = 1; i <= nd.length; i++) {
    const int64_t v = static_cast<int64_t>(off[i]);
    ok = ok && v >= prev;
    prev = v;
  }
  if (!ok || prev > limit)
    throw IOException(std::string("Arrow IPC validation failed: ") + what + " offsets of '" + nd.field->name +
                      "' are not monotonically non-decreasing or exceed the data buffer");
}

void ValidateNodeFull(const DecodedBatch& b, const DecodedNode& nd, const std::map<int64_t, std::shared_ptr<DictValues>>& dicts) {
  int32_t kind, w, nb;
  int64_t param;
  if (!nd.field->Plan(&kind, &param, &w, &nb, nd.value_only))
    throw NotImplementedException("Arrow type " + nd.field->Format() + " of field '" + nd.field->name + "' is not exported by this reader");
  switch (kind) {
    case MI_K_STR32: CheckOffsets<int32_t>(b, nd, nd.spans[2].length, "string"); break;
    case MI_K_STR64: CheckOffsets<int64_t>(b, nd, nd.spans[2].length, "string"); break;
    case MI_K_LIST32: CheckOffsets<int32_t>(b, nd, b.nodes[static_cast<size_t>(nd.children[0])].length, "list"); break;
    case MI_K_LIST64: CheckOffsets<int64_t>(b, nd, b.nodes[static_cast<size_t>(nd.children[0])].length, "list"); break;
    case MI_K_STRVIEW: {
      const uint8_t* v = b.body + nd.spans[1].offset;
      const uint64_t* valid = nd.spans[0].length ? reinterpret_cast<const uint64_t*>(b.body + nd.spans[0].offset) : nullptr;
      const uint8_t* vbytes = reinterpret_cast<const uint8_t*>(valid);
      for (int64_t i = 0; i < nd.length; i++) {
        if (vbytes && !((vbytes[i >> 3] >> (i & 7)) & 1)) continue;
        int32_t len, bi, bo;
        std::memcpy(&len, v + 16 * i, 4);
        if (len <= 12 && len >= 0) continue;
        std::memcpy(&bi, v + 16 * i + 8, 4);
        std::memcpy(&bo, v + 16 * i + 12, 4);
        const int64_t nvar = static_cast<int64_t>(nd.spans.size()) - 2;
        if (len < 0 || bi < 0 || bi >= nvar || bo < 0 || static_cast<int64_t>(bo) + len > nd.spans[static_cast<size_t>(2 + bi)].length)
          throw IOException("Arrow IPC validation failed: string view " + std::to_string(i) + " of '" + nd.field->name + "' points outside its data buffers");
      }
      break;
    }
    case MI_K_DICT: {
      auto it = dicts.find(nd.field->dict_id);
      if (it == dicts.end()) break;  // reported by the caller
      const int64_t dict_len = it->second->batch.nodes[static_cast<size_t>(it->second->batch.column_node[0])].length;
      const int iw = static_cast<int>(param & 0xFF);
      const bool sgn = ((param >> 8) & 1) != 0;
      const uint8_t* idx = b.body + nd.spans[1].offset;
      const uint8_t* vbytes = nd.spans[0].length ? b.body + nd.spans[0].offset : nullptr;
      for (int64_t i = 0; i < nd.length; i++) {
        if (vbytes && !((vbytes[i >> 3] >> (i & 7)) & 1)) continue;
        int64_t v = 0;
        switch (iw) {
          case 1: v = sgn ? static_cast<int64_t>(reinterpret_cast<const int8_t*>(idx)[i]) : idx[i]; break;
          case 2: { int16_t x; std::memcpy(&x, idx + 2 * i, 2); v = sgn ? x : static_cast<uint16_t>(x); break; }
          case 4: { int32_t x; std::memcpy(&x, idx + 4 * i, 4); v = sgn ? x : static_cast<uint32_t>(x); break; }
          default: std::memcpy(&v, idx + 8 * i, 8); break;
        }
        if (v < 0 || v >= dict_len)
          throw IOException("Arrow IPC validation failed: dictionary index out of range in '" + nd.field->name + "'");
      }
      break;
    }
    default: break;
  }
}

void ExportNode(const DecodedBatch& b, int32_t ni, const std::map<int64_t, std::shared_ptr<DictValues>>& dicts, ArrowArray* out) {
  const DecodedNode& nd = b.nodes[static_cast<size_t>(ni)];
  std::memset(out, 0, sizeof(*out));
  ValidateNodeFull(b, nd, dicts);
  auto* p = new ArrayPrivate();
  std::memset(out, 0, sizeof(*out));
  out->release = ReleaseArray;  // from here on a throw releases what was built
  out->private_data = p;
  p->body = b.owner;
  out->length = nd.length;
  out->null_count = nd.null_count;
  out->offset = 0;
  for (size_t k = 0; k < nd.spans.size(); k++) {
    const mi_buffer_span& s = nd.spans[k];
    // an absent validity bitmap is a NULL pointer; other empty buffers still get an address
    p->buffers.push_back((k == 0 && s.length == 0) ? nullptr : static_cast<const void*>(b.body + s.offset));
  }
  const int32_t t = nd.field->type;
  const bool dict = nd.field->has_dictionary && !nd.value_only;
  if (!dict && (t == MI_AT_UTF8_VIEW || t == MI_AT_BINARY_VIEW)) {
    // C data interface: views carry one extra trailing buffer with the sizes of the variadic data buffers (int64 each)
    auto sizes = std::make_shared<std::vector<int64_t>>();
    for (size_t k = 2; k < nd.spans.size(); k++) sizes->push_back(nd.spans[k].length);
    if (sizes->empty()) sizes->push_back(0);
    p->buffers.push_back(sizes->data());
    p->dictionary_body = std::shared_ptr<void>(sizes, sizes->data());  // keeps the sizes alive with the array
  }
  if (t == MI_AT_NULL && !dict) p->buffers.clear();
  out->n_buffers = static_cast<int64_t>(p->buffers.size());
  out->buffers = p->buffers.empty() ? nullptr : p->buffers.data();
  if (dict) {
    auto it = dicts.find(nd.field->dict_id);
    if (it == dicts.end()) {
      ReleaseArray(out);
      throw IOException("RecordBatch uses dictionary id " + std::to_string(nd.field->dict_id) + " before its DictionaryBatch");
    }
    p->dictionary = std::make_unique<ArrowArray>();
    ExportNode(it->second->batch, it->second->batch.column_node[0], dicts, p->dictionary.get());
    out->dictionary = p->dictionary.get();
  } else {
    p->children.resize(nd.children.size());
    for (size_t i = 0; i < nd.children.size(); i++) {
      try {
        ExportNode(b, nd.children[i], dicts, &p->children[i]);
      } catch (...) {
        ReleaseArray(out);
        throw;
      }
    }
    for (auto& c : p->children) p->child_ptrs.push_back(&c);
    out->n_children = static_cast<int64_t>(p->children.size());
    out->children = p->child_ptrs.empty() ? nullptr : p->child_ptrs.data();
  }
}

// ---- stream ------------------------------------------------------------------------------------------------------
struct StreamPrivate {
  std::unique_ptr<IPCStreamReader> reader;
  std::map<int64_t, std::shared_ptr<DictValues>> dicts;
  std::string last_error;
  bool accept_dictionaries = false;
};

int StreamGetSchema(ArrowArrayStream* s, ArrowSchema* out) {
  auto* p = static_cast<StreamPrivate*>(s->private_data);
  try {
    ExportSchema(p->reader->GetOutputSchema(), out);
    return 0;
  } catch (const std::exception& e) {
    p->last_error = e.what();
    return MI_EIO;
  }
}

int StreamGetNext(ArrowArrayStream* s, ArrowArray* out) {
  auto* p = static_cast<StreamPrivate*>(s->private_data);
  std::memset(out, 0, sizeof(*out));  // release == NULL: end of stream
  try {
    while (true) {
      DecodedBatch b;
      if (!p->reader->GetNextBatch(&b, p->accept_dictionaries)) return 0;
      p->reader->ReleaseCurrentBody();
      if (b.is_dictionary) {
        if (b.is_delta) throw NotImplementedException("delta dictionaries cannot be exported zero-copy through the Arrow C stream");
        auto d = std::make_shared<DictValues>();
        d->batch = std::move(b);
        p->dicts[d->batch.dict_id] = d;
        continue;
      }
      // the batch is a struct array whose children are the (projected) columns
      auto* root = new ArrayPrivate();
      out->release = ReleaseArray;
      out->private_data = root;
      out->length = b.length;
      out->null_count = 0;
      root->buffers.push_back(nullptr);
      out->n_buffers = 1;
      out->buffers = root->buffers.data();
      root->children.resize(b.column_node.size());
      for (size_t c = 0; c < b.column_node.size(); c++) {
        try {
          ExportNode(b, b.column_node[c], p->dicts, &root->children[c]);
        } catch (...) {
          ReleaseArray(out);
          std::memset(out, 0, sizeof(*out));
          throw;
        }
      }
      for (auto& c : root->children) root->child_ptrs.push_back(&c);
      out->n_children = static_cast<int64_t>(root->children.size());
      out->children = root->child_ptrs.empty() ? nullptr : root->child_ptrs.data();
      return 0;
    }
  } catch (const NotImplementedException& e) {
    p->last_error = e.what();
    return MI_ENOTSUP;
  } catch (const std::bad_alloc&) {
    p->last_error = "out of memory";
    return MI_ENOMEM;
  } catch (const std::exception& e) {  // IOException -> EIO, like IpcArrayStream::Wrap's get_next
    p->last_error = e.what();
    return MI_EIO;
  }
}

const char* StreamGetLastError(ArrowArrayStream* s) { return static_cast<StreamPrivate*>(s->private_data)->last_error.c_str(); }

void StreamRelease(ArrowArrayStream* s) {
  if (!s || !s->release) return;
  delete static_cast<StreamPrivate*>(s->private_data);
  s->release = nullptr;
}
}  // namespace
}  // namespace miarrow

using namespace miarrow;

extern "C" int mi_reader_export_stream(mi_reader* r, int32_t accept_dictionaries, struct ArrowArrayStream* out) {
  return WrapC([&] {
    if (!r || !out) throw InvalidInputException("mi_reader_export_stream: NULL argument");
    auto p = std::make_unique<StreamPrivate>();
    p->reader = TakeReader(r);
    if (!p->reader) throw InvalidInputException("mi_reader_export_stream: the reader was already exported or closed");
    p->accept_dictionaries = accept_dictionaries != 0;
    std::memset(out, 0, sizeof(*out));
    out->get_schema = StreamGetSchema;
    out->get_next = StreamGetNext;
    out->get_last_error = StreamGetLastError;
    out->release = StreamRelease;
    out->private_data = p.release();
  });
}
