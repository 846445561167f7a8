// ipc_format.cpp -- see ipc_format.hpp.
#include "ipc_format.hpp"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "flatbuf.hpp"

namespace miarrow {

const char* MessageTypeString(MessageType t) {
  switch (t) {
    case MessageType::SCHEMA: return "Schema";
    case MessageType::RECORD_BATCH: return "RecordBatch";
    case MessageType::DICTIONARY_BATCH: return "DictionaryBatch";
    case MessageType::TENSOR: return "Tensor";
    case MessageType::SPARSE_TENSOR: return "SparseTensor";
    case MessageType::UNINITIALIZED: return "Uninitialized";
    default: return "";
  }
}

// ------------------------------------------------------------------------------------------------ field model
std::string ArrowField::Format() const {
  char tmp[64];
  switch (type) {
    case MI_AT_NULL: return "n";
    case MI_AT_INT: {
      const char* s = "cCsSiIlL";
      int idx = bit_width == 8 ? 0 : bit_width == 16 ? 2 : bit_width == 32 ? 4 : 6;
      return std::string(1, s[idx + (is_signed ? 0 : 1)]);
    }
    case MI_AT_FLOAT: return precision == 0 ? "e" : precision == 1 ? "f" : "g";
    case MI_AT_BOOL: return "b";
    case MI_AT_UTF8: return "u";
    case MI_AT_LARGE_UTF8: return "U";
    case MI_AT_BINARY: return "z";
    case MI_AT_LARGE_BINARY: return "Z";
    case MI_AT_UTF8_VIEW: return "vu";
    case MI_AT_BINARY_VIEW: return "vz";
    case MI_AT_DECIMAL:
      if (bit_width == 128) std::snprintf(tmp, sizeof(tmp), "d:%d,%d", precision, scale);
      else std::snprintf(tmp, sizeof(tmp), "d:%d,%d,%d", precision, scale, bit_width);
      return tmp;
    case MI_AT_DATE: return unit == 0 ? "tdD" : "tdm";
    case MI_AT_TIME: return std::string("tt") + "smun"[unit & 3];
    case MI_AT_TIMESTAMP: return std::string("ts") + "smun"[unit & 3] + ":" + timezone;
    case MI_AT_DURATION: return std::string("tD") + "smun"[unit & 3];
    case MI_AT_INTERVAL: return unit == 0 ? "tiM" : unit == 1 ? "tiD" : "tin";
    case MI_AT_FIXED_BINARY: std::snprintf(tmp, sizeof(tmp), "w:%d", byte_width); return tmp;
    case MI_AT_LIST: return "+l";
    case MI_AT_LARGE_LIST: return "+L";
    case MI_AT_STRUCT: return "+s";
    case MI_AT_MAP: return "+m";
    case MI_AT_FIXED_LIST: std::snprintf(tmp, sizeof(tmp), "+w:%d", byte_width); return tmp;
    default: return "?";
  }
}

std::string ArrowField::DuckType() const {
  char tmp[64];
  switch (type) {
    case MI_AT_NULL: return "\"NULL\"";
    case MI_AT_INT: {
      const char* names[] = {"TINYINT", "UTINYINT", "SMALLINT", "USMALLINT", "INTEGER", "UINTEGER", "BIGINT", "UBIGINT"};
      int idx = bit_width == 8 ? 0 : bit_width == 16 ? 2 : bit_width == 32 ? 4 : 6;
      return names[idx + (is_signed ? 0 : 1)];
    }
    case MI_AT_FLOAT: return precision == 2 ? "DOUBLE" : "FLOAT";
    case MI_AT_BOOL: return "BOOLEAN";
    case MI_AT_UTF8: case MI_AT_LARGE_UTF8: case MI_AT_UTF8_VIEW: return "VARCHAR";
    case MI_AT_BINARY: case MI_AT_LARGE_BINARY: case MI_AT_BINARY_VIEW: case MI_AT_FIXED_BINARY: return "BLOB";
    case MI_AT_DECIMAL: std::snprintf(tmp, sizeof(tmp), "DECIMAL(%d,%d)", precision, scale); return tmp;
    case MI_AT_DATE: return "DATE";
    case MI_AT_TIME: return "TIME";
    case MI_AT_TIMESTAMP:
      if (!timezone.empty()) return "TIMESTAMP WITH TIME ZONE";
      return unit == 0 ? "TIMESTAMP_S" : unit == 1 ? "TIMESTAMP_MS" : unit == 2 ? "TIMESTAMP" : "TIMESTAMP_NS";
    case MI_AT_DURATION: case MI_AT_INTERVAL: return "INTERVAL";
    case MI_AT_LIST: case MI_AT_LARGE_LIST: return children.empty() ? "?[]" : children[0].DuckType() + "[]";
    case MI_AT_FIXED_LIST:
      std::snprintf(tmp, sizeof(tmp), "[%d]", byte_width);
      return (children.empty() ? std::string("?") : children[0].DuckType()) + tmp;
    case MI_AT_STRUCT: {
      std::string s = "STRUCT(";
      for (size_t i = 0; i < children.size(); i++) {
        if (i) s += ", ";
        s += children[i].name + " " + children[i].DuckType();
      }
      return s + ")";
    }
    case MI_AT_MAP: {
      if (children.size() == 1 && children[0].children.size() == 2)
        return "MAP(" + children[0].children[0].DuckType() + ", " + children[0].children[1].DuckType() + ")";
      return "MAP(?, ?)";
    }
    default: return "?";
  }
}

bool ArrowField::Plan(int32_t* kind, int64_t* param, int32_t* out_width, int32_t* n_buffers, bool value_only) const {
  *kind = 0;
  *param = 0;
  *out_width = 0;
  *n_buffers = 2;
  if (has_dictionary && !value_only) {
    *kind = MI_K_DICT;
    *param = (dict_index_bit_width / 8) | (static_cast<int64_t>(dict_index_signed ? 1 : 0) << 8);
    *out_width = 4;
    return true;
  }
  auto set = [&](int32_t k, int64_t p, int32_t w) { *kind = k; *param = p; *out_width = w; return true; };
  switch (type) {
    case MI_AT_INT: return set(MI_K_COPY, bit_width / 8, bit_width / 8);
    case MI_AT_NULL: *n_buffers = 0; return set(MI_K_NULL, 0, 1);
    case MI_AT_FLOAT:
      if (precision == 0) return set(MI_K_HALF_FLOAT, 0, 4);
      return set(MI_K_COPY, precision == 1 ? 4 : 8, precision == 1 ? 4 : 8);
    case MI_AT_BOOL: return set(MI_K_BOOL, 0, 1);
    case MI_AT_DECIMAL:
      if ((bit_width == 32 && precision <= 9) || (bit_width == 64 && precision <= 18)) {
        const int32_t sw = bit_width / 8, dw = precision <= 4 ? 2 : precision <= 9 ? 4 : 8;
        return sw == dw ? set(MI_K_COPY, sw, sw) : set(MI_K_NARROW, sw | (dw << 8), dw);
      }
      if (bit_width != 128 || precision > 38) return false;
      if (precision <= 4) return set(MI_K_DEC128, 2, 2);
      if (precision <= 9) return set(MI_K_DEC128, 4, 4);
      if (precision <= 18) return set(MI_K_DEC128, 8, 8);
      return set(MI_K_COPY, 16, 16);
    case MI_AT_DATE: return unit == 0 ? set(MI_K_COPY, 4, 4) : set(MI_K_DATE64, 0, 4);
    case MI_AT_TIME:
      switch (unit) {
        case 0: return set(MI_K_MUL_I32, 1000000, 8);
        case 1: return set(MI_K_MUL_I32, 1000, 8);
        case 2: return set(MI_K_COPY, 8, 8);
        default: return set(MI_K_DIV_I64, 1000, 8);
      }
    case MI_AT_TIMESTAMP:
      if (timezone.empty()) return set(MI_K_COPY, 8, 8);
      switch (unit) {
        case 0: return set(MI_K_MUL_I64, 1000000, 8);
        case 1: return set(MI_K_MUL_I64, 1000, 8);
        case 2: return set(MI_K_COPY, 8, 8);
        default: return set(MI_K_DIV_I64, 1000, 8);
      }
    case MI_AT_DURATION:
      return set(MI_K_DURATION, unit == 0 ? 1000000 : unit == 1 ? 1000 : unit == 2 ? 1 : -1000, 16);
    case MI_AT_INTERVAL:
      if (unit == 0) return set(MI_K_INTERVAL_MONTHS, 0, 16);
      if (unit == 2) return set(MI_K_INTERVAL_MDN, 0, 16);
      return false;  // day_time: upstream's conversion is not restated
    case MI_AT_UTF8: case MI_AT_BINARY: *n_buffers = 3; return set(MI_K_STR32, 0, 16);
    case MI_AT_LARGE_UTF8: case MI_AT_LARGE_BINARY: *n_buffers = 3; return set(MI_K_STR64, 0, 16);
    case MI_AT_FIXED_BINARY: return set(MI_K_FIXED_BINARY, byte_width, 16);
    case MI_AT_UTF8_VIEW: case MI_AT_BINARY_VIEW: return set(MI_K_STRVIEW, 0, 16);  // + variadic data buffers
    case MI_AT_LIST: case MI_AT_MAP: return set(MI_K_LIST32, 0, 16);
    case MI_AT_LARGE_LIST: return set(MI_K_LIST64, 0, 16);
    case MI_AT_STRUCT: *n_buffers = 1; return set(MI_K_STRUCT, 0, 0);
    case MI_AT_FIXED_LIST: *n_buffers = 1; return set(MI_K_STRUCT, byte_width, 0);  // DuckDB ARRAY: validity + one child
    default: return false;
  }
}

bool ArrowField::Supported(std::string* why) const {
  int32_t kind, w, nb;
  int64_t param;
  if (!Plan(&kind, &param, &w, &nb)) {
    if (why) *why = "Arrow type " + Format() + " of field '" + name + "'";
    return false;
  }
  if (has_dictionary) return true;
  for (auto& c : children)
    if (!c.Supported(why)) return false;
  return true;
}

int64_t ArrowField::CountFields() const {
  int64_t n = 1;
  if (has_dictionary) return n;  // children of a dictionary-encoded field live in the dictionary batch
  for (auto& c : children) n += c.CountFields();
  return n;
}

int64_t ArrowField::CountBuffers() const {
  int64_t own;
  if (has_dictionary) return 2;
  switch (type) {
    case MI_AT_NULL: own = 0; break;
    case MI_AT_STRUCT: case MI_AT_FIXED_LIST: own = 1; break;
    case MI_AT_UTF8: case MI_AT_BINARY: case MI_AT_LARGE_UTF8: case MI_AT_LARGE_BINARY: own = 3; break;
    case MI_AT_UNION: own = 2; break;  // dense (sparse has 1): unions are outside the path
    default: own = 2; break;           // validity + data / offsets (list, map) / views (+ variadic, outside the path)
  }
  for (auto& c : children) own += c.CountBuffers();
  return own;
}

void FillCField(const ArrowField& f, int32_t flat_index, mi_field* out) {
  std::memset(out, 0, sizeof(*out));
  std::snprintf(out->name, sizeof(out->name), "%s", f.name.c_str());
  std::snprintf(out->timezone, sizeof(out->timezone), "%s", f.timezone.c_str());
  const std::string duck = f.DuckType();
  if (duck.size() >= sizeof(out->duck_type))
    throw NotImplementedException("Column '" + f.name + "': the DuckDB type description is longer than " +
                                  std::to_string(sizeof(out->duck_type) - 1) + " characters");
  std::snprintf(out->duck_type, sizeof(out->duck_type), "%s", duck.c_str());
  std::snprintf(out->format, sizeof(out->format), "%s", f.Format().c_str());
  out->arrow_type = f.type;
  out->bit_width = f.bit_width;
  out->is_signed = f.is_signed;
  out->precision = f.precision;
  out->scale = f.scale;
  out->unit = f.unit;
  out->byte_width = f.byte_width;
  out->nullable = f.nullable;
  out->has_dictionary = f.has_dictionary;
  out->dict_index_bit_width = f.dict_index_bit_width;
  out->dict_index_signed = f.dict_index_signed;
  out->dict_id = f.dict_id;
  int32_t kind, w, nb;
  int64_t param;
  if (f.Plan(&kind, &param, &w, &nb)) {
    out->kind = kind;
    out->param = param;
    out->out_width = w;
  }
  out->n_buffers = static_cast<int32_t>(f.CountBuffers());
  out->flat_index = flat_index;
}

// ------------------------------------------------------------------------------------------------ decode
MessageHeader DecodeMessageHeader(const uint8_t* meta, int64_t meta_len) {
  // Message { version:short [0]; header_type:ubyte [1]; header [2]; bodyLength:long [3]; custom_metadata [4] }
  fb::Buf b{meta, meta_len};
  fb::Table m = fb::root(&b);
  if (!m) throw IOException("Message flatbuffer verification failed");
  MessageHeader h;
  h.version = m.scalar<int16_t>(0, 0);
  int type = m.scalar<uint8_t>(1, 0);
  h.body_length = m.scalar<int64_t>(3, 0);
  if (type < 1 || type > 5) throw IOException("Unexpected Message header type " + std::to_string(type));
  if (!m.table(2)) throw IOException("Message header is missing");
  if (h.body_length < 0) throw IOException("Expected body size >= 0 but got " + std::to_string(h.body_length));
  h.type = static_cast<MessageType>(type);
  return h;
}

static void DecodeKeyValues(const fb::Table& t, int id, std::vector<std::pair<std::string, std::string>>* out) {
  uint32_t n;
  int64_t v = t.vector(id, &n);
  for (uint32_t i = 0; i < n && v >= 0; i++) {
    fb::Table kv = t.vector_table(v, i);
    if (!kv) throw IOException("Invalid KeyValue in schema metadata");
    std::string k, val;
    kv.string(0, &k);
    kv.string(1, &val);
    out->emplace_back(std::move(k), std::move(val));
  }
}

static ArrowField DecodeField(const fb::Table& f, int depth) {
  // Field { name [0]; nullable [1]; type_type [2]; type [3]; dictionary [4]; children [5]; custom_metadata [6] }
  if (depth > 64) throw IOException("Schema nesting too deep");
  ArrowField o;
  f.string(0, &o.name);
  o.nullable = f.scalar<uint8_t>(1, 0) != 0;
  o.type = f.scalar<uint8_t>(2, 0);
  fb::Table t = f.table(3);
  switch (o.type) {
    case MI_AT_INT:
      o.bit_width = t.scalar<int32_t>(0, 0);
      o.is_signed = t.scalar<uint8_t>(1, 0) != 0;
      if (o.bit_width != 8 && o.bit_width != 16 && o.bit_width != 32 && o.bit_width != 64)
        throw IOException("Expected integer bit width of 8, 16, 32 or 64 but got " + std::to_string(o.bit_width));
      break;
    case MI_AT_FLOAT: o.precision = t.scalar<int16_t>(0, 0); break;
    case MI_AT_DECIMAL:
      o.precision = t.scalar<int32_t>(0, 0);
      o.scale = t.scalar<int32_t>(1, 0);
      o.bit_width = t.scalar<int32_t>(2, 128);
      if (o.bit_width != 32 && o.bit_width != 64 && o.bit_width != 128 && o.bit_width != 256)
        throw IOException("Expected decimal bit width of 32, 64, 128 or 256 but got " + std::to_string(o.bit_width));
      break;
    case MI_AT_DATE: o.unit = t.scalar<int16_t>(0, 1); break;
    case MI_AT_TIME:
      o.unit = t.scalar<int16_t>(0, 1);
      o.bit_width = t.scalar<int32_t>(1, 32);
      if (o.bit_width != 32 && o.bit_width != 64) throw IOException("Expected time bit width of 32 or 64 but got " + std::to_string(o.bit_width));
      break;
    case MI_AT_TIMESTAMP:
      o.unit = t.scalar<int16_t>(0, 0);
      t.string(1, &o.timezone);
      break;
    case MI_AT_DURATION: o.unit = t.scalar<int16_t>(0, 1); break;
    case MI_AT_INTERVAL: o.unit = t.scalar<int16_t>(0, 0); break;
    // widths taken from the file multiply row counts and addresses later: nanoarrow rejects these at schema decode too
    case MI_AT_FIXED_BINARY:
      o.byte_width = t.scalar<int32_t>(0, 0);
      if (o.byte_width <= 0) throw IOException("Expected FixedSizeBinary byteWidth > 0 but got " + std::to_string(o.byte_width));
      break;
    case MI_AT_FIXED_LIST:
      o.byte_width = t.scalar<int32_t>(0, 0);
      if (o.byte_width < 0) throw IOException("Expected FixedSizeList listSize >= 0 but got " + std::to_string(o.byte_width));
      break;
    case MI_AT_UNION: o.unit = t.scalar<int16_t>(0, 0); break;  // UnionMode: 0 sparse, 1 dense
    default: break;
  }
  fb::Table d = f.table(4);
  if (d) {
    // DictionaryEncoding { id:long [0]; indexType:Int [1]; isOrdered:bool [2]; dictionaryKind:short [3] }
    o.has_dictionary = true;
    o.dict_id = d.scalar<int64_t>(0, 0);
    fb::Table it = d.table(1);
    o.dict_index_bit_width = it ? it.scalar<int32_t>(0, 32) : 32;
    o.dict_index_signed = it ? it.scalar<uint8_t>(1, 1) != 0 : true;
    if (o.dict_index_bit_width != 8 && o.dict_index_bit_width != 16 && o.dict_index_bit_width != 32 && o.dict_index_bit_width != 64)
      throw IOException("Expected dictionary index bit width of 8, 16, 32 or 64 but got " + std::to_string(o.dict_index_bit_width));
    o.dict_ordered = d.scalar<uint8_t>(2, 0) != 0;
  }
  uint32_t nchild;
  int64_t cv = f.vector(5, &nchild);
  for (uint32_t i = 0; i < nchild && cv >= 0; i++) {
    fb::Table c = f.vector_table(cv, i);
    if (!c) throw IOException("Invalid child Field in schema");
    o.children.push_back(DecodeField(c, depth + 1));
  }
  DecodeKeyValues(f, 6, &o.metadata);
  return o;
}

ArrowSchemaModel DecodeSchema(const uint8_t* meta, int64_t meta_len) {
  fb::Buf b{meta, meta_len};
  fb::Table m = fb::root(&b);
  if (!m || m.scalar<uint8_t>(1, 0) != 1) throw IOException("Expected Schema message");
  // Schema { endianness:short [0]; fields:[Field] [1]; custom_metadata [2]; features:[long] [3] }
  fb::Table s = m.table(2);
  if (!s) throw IOException("Schema message header is missing");
  ArrowSchemaModel out;
  out.endianness = s.scalar<int16_t>(0, 0);
  uint32_t nf;
  int64_t fv = s.vector(1, &nf);
  for (uint32_t i = 0; i < nf && fv >= 0; i++) {
    fb::Table f = s.vector_table(fv, i);
    if (!f) throw IOException("Invalid Field in schema");
    out.fields.push_back(DecodeField(f, 0));
  }
  DecodeKeyValues(s, 2, &out.metadata);
  uint32_t nfeat;
  int64_t ftv = s.vector(3, &nfeat);
  for (uint32_t i = 0; i < nfeat && ftv >= 0 && b.in(ftv + 8 * static_cast<int64_t>(i), 8); i++) {
    int64_t feat = fb::load<int64_t>(b.base + ftv + 8 * static_cast<int64_t>(i));
    if (feat >= 0 && feat < 32) out.features |= 1u << feat;
  }
  return out;
}

RecordBatchMeta DecodeRecordBatch(const uint8_t* meta, int64_t meta_len) {
  fb::Buf b{meta, meta_len};
  fb::Table m = fb::root(&b);
  if (!m) throw IOException("Message flatbuffer verification failed");
  int type = m.scalar<uint8_t>(1, 0);
  fb::Table rb = m.table(2);
  RecordBatchMeta out;
  if (type == 2) {
    // DictionaryBatch { id:long [0]; data:RecordBatch [1]; isDelta:bool [2] }
    out.is_dictionary = true;
    out.dict_id = rb.scalar<int64_t>(0, 0);
    out.is_delta = rb.scalar<uint8_t>(2, 0) != 0;
    rb = rb.table(1);
  } else if (type != 3) {
    throw IOException("Expected RecordBatch or DictionaryBatch message");
  }
  if (!rb) throw IOException("RecordBatch header is missing");
  // RecordBatch { length [0]; nodes:[FieldNode] [1]; buffers:[Buffer] [2]; compression [3]; variadicBufferCounts [4] }
  out.length = rb.scalar<int64_t>(0, 0);
  if (out.length < 0) throw IOException("RecordBatch length is negative");
  uint32_t nn, nb, nv;
  int64_t np = rb.vector(1, &nn);
  int64_t bp = rb.vector(2, &nb);
  if (nn && !b.in(np, 16 * static_cast<int64_t>(nn))) throw IOException("RecordBatch nodes out of bounds");
  if (nb && !b.in(bp, 16 * static_cast<int64_t>(nb))) throw IOException("RecordBatch buffers out of bounds");
  out.nodes.reserve(nn);
  for (uint32_t i = 0; i < nn; i++)
    out.nodes.emplace_back(fb::load<int64_t>(b.base + np + 16 * static_cast<int64_t>(i)),
                           fb::load<int64_t>(b.base + np + 16 * static_cast<int64_t>(i) + 8));
  out.buffers.reserve(nb);
  for (uint32_t i = 0; i < nb; i++)
    out.buffers.push_back(mi_buffer_span{fb::load<int64_t>(b.base + bp + 16 * static_cast<int64_t>(i)),
                                         fb::load<int64_t>(b.base + bp + 16 * static_cast<int64_t>(i) + 8)});
  fb::Table c = rb.table(3);
  if (c) out.compression = static_cast<int8_t>(c.scalar<uint8_t>(0, 0));  // BodyCompression.codec
  int64_t vp = rb.vector(4, &nv);
  for (uint32_t i = 0; i < nv && vp >= 0 && b.in(vp + 8 * static_cast<int64_t>(i), 8); i++)
    out.variadic_counts.push_back(fb::load<int64_t>(b.base + vp + 8 * static_cast<int64_t>(i)));
  return out;
}

bool DecodeFooter(const uint8_t* tail, int64_t tail_len, int64_t file_size, std::vector<FooterBlock>* dict_blocks,
                  std::vector<FooterBlock>* batch_blocks) {
  // ... footer flatbuffer | int32 footer_len | "ARROW1"
  if (tail_len < 10 || std::memcmp(tail + tail_len - 6, "ARROW1", 6) != 0) return false;
  int32_t flen = fb::load<int32_t>(tail + tail_len - 10);
  if (flen <= 0 || static_cast<int64_t>(flen) + 10 > tail_len || static_cast<int64_t>(flen) + 18 > file_size) return false;
  fb::Buf b{tail + tail_len - 10 - flen, flen};
  fb::Table f = fb::root(&b);
  if (!f) return false;
  // Footer { version [0]; schema [1]; dictionaries:[Block] [2]; recordBatches:[Block] [3] }; Block = 24 bytes
  auto read_blocks = [&](int id, std::vector<FooterBlock>* out) {
    uint32_t n;
    int64_t p = f.vector(id, &n);
    if (n && !b.in(p, 24 * static_cast<int64_t>(n))) return false;
    for (uint32_t i = 0; i < n; i++) {
      const uint8_t* e = b.base + p + 24 * static_cast<int64_t>(i);
      out->push_back(FooterBlock{fb::load<int64_t>(e), fb::load<int32_t>(e + 8), fb::load<int64_t>(e + 16)});
    }
    return true;
  };
  return read_blocks(2, dict_blocks) && read_blocks(3, batch_blocks);
}

// ------------------------------------------------------------------------------------------------ encode
static fb::Builder::Offset BuildKeyValues(fb::Builder& fbb, const std::vector<std::pair<std::string, std::string>>& kv) {
  if (kv.empty()) return 0;
  std::vector<fb::Builder::Offset> items;
  for (auto& p : kv) {
    auto k = fbb.CreateString(p.first);
    auto v = fbb.CreateString(p.second);
    fbb.StartTable();
    fbb.AddOffset(0, k);
    fbb.AddOffset(1, v);
    items.push_back(fbb.EndTable());
  }
  return fbb.CreateOffsetVector(items);
}

static fb::Builder::Offset BuildIntType(fb::Builder& fbb, int32_t bits, bool is_signed) {
  fbb.StartTable();
  fbb.AddScalar<int32_t>(0, bits, 0);
  fbb.AddScalar<uint8_t>(1, is_signed ? 1 : 0, 0);
  return fbb.EndTable();
}

static fb::Builder::Offset BuildField(fb::Builder& fbb, const ArrowField& f) {
  std::vector<fb::Builder::Offset> children;
  for (auto& c : f.children) children.push_back(BuildField(fbb, c));
  auto children_vec = fbb.CreateOffsetVector(children);
  auto name = fbb.CreateString(f.name);
  auto kv = BuildKeyValues(fbb, f.metadata);
  fb::Builder::Offset tz = 0;
  if (f.type == MI_AT_TIMESTAMP && !f.timezone.empty()) tz = fbb.CreateString(f.timezone);
  fb::Builder::Offset dict = 0;
  if (f.has_dictionary) {
    auto it = BuildIntType(fbb, f.dict_index_bit_width, f.dict_index_signed);
    fbb.StartTable();
    fbb.AddScalar<int64_t>(0, f.dict_id, 0);
    fbb.AddOffset(1, it);
    fbb.AddScalar<uint8_t>(2, f.dict_ordered ? 1 : 0, 0);
    dict = fbb.EndTable();
  }
  // the type table (tables without fields still need an (empty) table object)
  fb::Builder::Offset type;
  fbb.StartTable();
  switch (f.type) {
    case MI_AT_INT:
      fbb.AddScalar<int32_t>(0, f.bit_width, 0);
      fbb.AddScalar<uint8_t>(1, f.is_signed ? 1 : 0, 0);
      break;
    case MI_AT_FLOAT: fbb.AddScalar<int16_t>(0, static_cast<int16_t>(f.precision), 0); break;
    case MI_AT_DECIMAL:
      fbb.AddScalar<int32_t>(0, f.precision, 0);
      fbb.AddScalar<int32_t>(1, f.scale, 0);
      fbb.AddScalar<int32_t>(2, f.bit_width, 128);
      break;
    case MI_AT_DATE: fbb.AddScalar<int16_t>(0, static_cast<int16_t>(f.unit), 1); break;
    case MI_AT_TIME:
      fbb.AddScalar<int16_t>(0, static_cast<int16_t>(f.unit), 1);
      fbb.AddScalar<int32_t>(1, f.bit_width, 32);
      break;
    case MI_AT_TIMESTAMP:
      fbb.AddScalar<int16_t>(0, static_cast<int16_t>(f.unit), 0);
      fbb.AddOffset(1, tz);
      break;
    case MI_AT_DURATION: fbb.AddScalar<int16_t>(0, static_cast<int16_t>(f.unit), 1); break;
    case MI_AT_INTERVAL: fbb.AddScalar<int16_t>(0, static_cast<int16_t>(f.unit), 0); break;
    case MI_AT_FIXED_BINARY: case MI_AT_FIXED_LIST: fbb.AddScalar<int32_t>(0, f.byte_width, 0); break;
    default: break;
  }
  type = fbb.EndTable();
  fbb.StartTable();
  fbb.AddOffset(0, name);
  fbb.AddScalar<uint8_t>(1, f.nullable ? 1 : 0, 0);
  fbb.AddScalar<uint8_t>(2, static_cast<uint8_t>(f.type), 0);
  fbb.AddOffset(3, type);
  fbb.AddOffset(4, dict);
  fbb.AddOffset(5, children_vec);
  fbb.AddOffset(6, kv);
  return fbb.EndTable();
}

// 8-byte prefix {0xFFFFFFFF, int32 padded_len} + flatbuffer + zero padding to a multiple of 8
static std::vector<uint8_t> Encapsulate(const std::vector<uint8_t>& flatbuffer) {
  size_t padded = (flatbuffer.size() + 7) & ~static_cast<size_t>(7);
  std::vector<uint8_t> out(8 + padded, 0);
  uint32_t token = 0xFFFFFFFFu;
  int32_t len = static_cast<int32_t>(padded);
  std::memcpy(out.data(), &token, 4);
  std::memcpy(out.data() + 4, &len, 4);
  std::memcpy(out.data() + 8, flatbuffer.data(), flatbuffer.size());
  return out;
}

std::vector<uint8_t> EncodeSchemaMessage(const ArrowSchemaModel& schema) {
  fb::Builder fbb(4096);
  std::vector<fb::Builder::Offset> fields;
  for (auto& f : schema.fields) fields.push_back(BuildField(fbb, f));
  auto fields_vec = fbb.CreateOffsetVector(fields);
  auto kv = BuildKeyValues(fbb, schema.metadata);
  fbb.StartTable();
  fbb.AddScalar<int16_t>(0, static_cast<int16_t>(schema.endianness), 0);
  fbb.AddOffset(1, fields_vec);
  fbb.AddOffset(2, kv);
  auto s = fbb.EndTable();
  fbb.StartTable();
  fbb.AddScalar<int16_t>(0, 4, 0);   // MetadataVersion V5
  fbb.AddScalar<uint8_t>(1, 1, 0);   // MessageHeader.Schema
  fbb.AddOffset(2, s);
  fbb.AddScalar<int64_t>(3, 0, 0);   // bodyLength
  auto m = fbb.EndTable();
  return Encapsulate(fbb.Finish(m));
}

std::vector<uint8_t> EncodeRecordBatchMessage(int64_t length, const std::vector<std::pair<int64_t, int64_t>>& nodes,
                                              const std::vector<mi_buffer_span>& buffers, int64_t body_length) {
  fb::Builder fbb(1024 + 16 * (nodes.size() + buffers.size()));
  static_assert(sizeof(mi_buffer_span) == 16, "Buffer struct layout");
  static_assert(sizeof(mi_string_t) == 16, "string_t layout");
  static_assert(sizeof(std::pair<int64_t, int64_t>) == 16, "FieldNode struct layout");
  auto bufs = fbb.CreateStructVector(buffers.data(), buffers.size(), 16, 8);
  auto nds = fbb.CreateStructVector(nodes.data(), nodes.size(), 16, 8);
  fbb.StartTable();
  fbb.AddScalar<int64_t>(0, length, 0);
  fbb.AddOffset(1, nds);
  fbb.AddOffset(2, bufs);
  auto rb = fbb.EndTable();
  fbb.StartTable();
  fbb.AddScalar<int16_t>(0, 4, 0);   // V5
  fbb.AddScalar<uint8_t>(1, 3, 0);   // MessageHeader.RecordBatch
  fbb.AddOffset(2, rb);
  fbb.AddScalar<int64_t>(3, body_length, 0);
  auto m = fbb.EndTable();
  return Encapsulate(fbb.Finish(m));
}

// ------------------------------------------------------------------------------------------------ DuckDB -> Arrow
static std::string Upper(std::string s) {
  for (auto& c : s) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
  return s;
}

// Split "a INTEGER, b STRUCT(x INT, y INT)" on top-level commas (outside parentheses and double quotes).
static std::vector<std::string> SplitTopLevel(const std::string& text) {
  std::vector<std::string> parts;
  std::string cur;
  int depth = 0;
  bool quoted = false;
  for (char ch : text) {
    if (ch == '"') quoted = !quoted;
    if (!quoted) {
      if (ch == '(') depth++;
      if (ch == ')') depth--;
      if (ch == ',' && depth == 0) {
        parts.push_back(cur);
        cur.clear();
        continue;
      }
    }
    cur += ch;
  }
  if (!cur.empty() || !parts.empty()) parts.push_back(cur);
  for (auto& p : parts) {
    size_t a = p.find_first_not_of(" \t"), b = p.find_last_not_of(" \t");
    p = a == std::string::npos ? std::string() : p.substr(a, b - a + 1);
  }
  return parts;
}

// Nested DuckDB types the way ArrowConverter::ToArrowSchema exports them: LIST -> list<l: T>, ARRAY -> fixed_size_list
// <l: T>[N], STRUCT -> struct with the field names, MAP -> map<entries: struct<key: K not null, value: V> not null>.
static bool NestedFieldFromDuckType(const std::string& name, const std::string& duck_type, ArrowField* out) {
  std::string t = duck_type;
  while (!t.empty() && std::isspace(static_cast<unsigned char>(t.back()))) t.pop_back();
  if (t.empty()) return false;
  ArrowField f;
  f.name = name;
  f.nullable = true;
  if (t.back() == ']') {
    const size_t open = t.rfind('[');
    if (open == std::string::npos) throw InvalidInputException("Malformed type '" + duck_type + "'");
    const std::string size = t.substr(open + 1, t.size() - open - 2);
    f.children.push_back(FieldFromDuckType("l", t.substr(0, open)));
    if (size.empty()) {
      f.type = MI_AT_LIST;
    } else {
      f.type = MI_AT_FIXED_LIST;
      f.byte_width = std::atoi(size.c_str());
      if (f.byte_width <= 0) throw InvalidInputException("Malformed array size in '" + duck_type + "'");
    }
    *out = f;
    return true;
  }
  const std::string up = Upper(t);
  if (up.rfind("STRUCT(", 0) == 0 && t.back() == ')') {
    f.type = MI_AT_STRUCT;
    for (auto& part : SplitTopLevel(t.substr(7, t.size() - 8))) {
      std::string child_name, rest;
      if (!part.empty() && part[0] == '"') {
        const size_t q = part.find('"', 1);
        if (q == std::string::npos) throw InvalidInputException("Malformed struct field in '" + duck_type + "'");
        child_name = part.substr(1, q - 1);
        rest = part.substr(q + 1);
      } else {
        const size_t sp = part.find(' ');
        if (sp == std::string::npos) throw InvalidInputException("Malformed struct field in '" + duck_type + "'");
        child_name = part.substr(0, sp);
        rest = part.substr(sp + 1);
      }
      f.children.push_back(FieldFromDuckType(child_name, rest));
    }
    if (f.children.empty()) throw InvalidInputException("STRUCT without fields: '" + duck_type + "'");
    *out = f;
    return true;
  }
  if (up.rfind("MAP(", 0) == 0 && t.back() == ')') {
    auto kv = SplitTopLevel(t.substr(4, t.size() - 5));
    if (kv.size() != 2) throw InvalidInputException("MAP needs a key and a value type: '" + duck_type + "'");
    f.type = MI_AT_MAP;
    ArrowField entries;
    entries.name = "entries";
    entries.type = MI_AT_STRUCT;
    entries.nullable = false;
    entries.children.push_back(FieldFromDuckType("key", kv[0]));
    entries.children[0].nullable = false;
    entries.children.push_back(FieldFromDuckType("value", kv[1]));
    f.children.push_back(entries);
    *out = f;
    return true;
  }
  return false;
}

ArrowField FieldFromDuckType(const std::string& name, const std::string& duck_type) {
  ArrowField f;
  f.name = name;
  f.nullable = true;  // ArrowConverter::ToArrowSchema sets ARROW_FLAG_NULLABLE on every column
  if (NestedFieldFromDuckType(name, duck_type, &f)) return f;
  std::string t = Upper(duck_type);
  while (!t.empty() && std::isspace(static_cast<unsigned char>(t.back()))) t.pop_back();
  while (!t.empty() && std::isspace(static_cast<unsigned char>(t.front()))) t.erase(t.begin());
  auto integer = [&](int bits, bool sign) { f.type = MI_AT_INT; f.bit_width = bits; f.is_signed = sign; return f; };
  if (t == "BOOLEAN" || t == "BOOL") { f.type = MI_AT_BOOL; return f; }
  if (t == "TINYINT") return integer(8, true);
  if (t == "UTINYINT") return integer(8, false);
  if (t == "SMALLINT") return integer(16, true);
  if (t == "USMALLINT") return integer(16, false);
  if (t == "INTEGER" || t == "INT") return integer(32, true);
  if (t == "UINTEGER") return integer(32, false);
  if (t == "BIGINT") return integer(64, true);
  if (t == "UBIGINT") return integer(64, false);
  if (t == "FLOAT") { f.type = MI_AT_FLOAT; f.precision = 1; return f; }
  if (t == "DOUBLE") { f.type = MI_AT_FLOAT; f.precision = 2; return f; }
  if (t == "VARCHAR") { f.type = MI_AT_UTF8; return f; }
  if (t == "BLOB") { f.type = MI_AT_BINARY; return f; }
  if (t == "DATE") { f.type = MI_AT_DATE; f.unit = 0; return f; }
  if (t == "TIME") { f.type = MI_AT_TIME; f.unit = 2; f.bit_width = 64; return f; }
  if (t == "TIMESTAMP") { f.type = MI_AT_TIMESTAMP; f.unit = 2; return f; }
  if (t == "TIMESTAMP_S") { f.type = MI_AT_TIMESTAMP; f.unit = 0; return f; }
  if (t == "TIMESTAMP_MS") { f.type = MI_AT_TIMESTAMP; f.unit = 1; return f; }
  if (t == "TIMESTAMP_NS") { f.type = MI_AT_TIMESTAMP; f.unit = 3; return f; }
  if (t == "TIMESTAMP WITH TIME ZONE" || t == "TIMESTAMPTZ") { f.type = MI_AT_TIMESTAMP; f.unit = 2; f.timezone = "UTC"; return f; }
  if (t == "HUGEINT") { f.type = MI_AT_DECIMAL; f.precision = 38; f.scale = 0; f.bit_width = 128; return f; }
  int p = 0, s = 0;
  if (std::sscanf(t.c_str(), "DECIMAL(%d,%d)", &p, &s) == 2 || std::sscanf(t.c_str(), "DECIMAL(%d, %d)", &p, &s) == 2) {
    if (p < 1 || p > 38 || s < 0 || s > p) throw InvalidInputException("Invalid DECIMAL width/scale in '" + duck_type + "'");
    f.type = MI_AT_DECIMAL;
    f.precision = p;
    f.scale = s;
    f.bit_width = 128;
    return f;
  }
  throw NotImplementedException("Unsupported DuckDB type for Arrow export on this path: " + duck_type);
}

}  // namespace miarrow
