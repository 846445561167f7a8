/*
 * oracle_ipc.c -- CPU restatement of the reference's IPC framing (TEST INFRASTRUCTURE, see oracle.h).
 *
 * Follows:
 *   IPCFileStreamReader::ReadNextMessage      src/ipc/stream_reader/ipc_file_stream_reader.cpp:96-132
 *   IPCFileStreamReader::EnsureInputStreamAligned                                        :134-141
 *   IPCStreamReader::DecodeMetadata / DecodeMessage   src/ipc/stream_reader/base_stream_reader.cpp:214-236
 *   IPCFileStreamReader::DecodeHeader (ENODATA => end of stream)   ipc_file_stream_reader.cpp:47-69
 *   IPCFileStreamReader::DecodeBody (align, then body_size_bytes)                        :71-89
 * The flatbuffer walk restates what nanoarrow's ArrowIpcDecoderDecodeHeader / DecodeSchema do with the
 * published Message.fbs / Schema.fbs / File.fbs layouts (nanoarrow is not vendored: CMakeLists.txt:7-13).
 */
#include "oracle.h"

#include <stdio.h>
#include <string.h>

/* ---------------------------------------------------------------- little-endian + flatbuffer access */
static uint16_t rd_u16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
static int16_t rd_i16(const uint8_t* p) { int16_t v; memcpy(&v, p, 2); return v; }
static uint32_t rd_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static int32_t rd_i32(const uint8_t* p) { int32_t v; memcpy(&v, p, 4); return v; }
static int64_t rd_i64(const uint8_t* p) { int64_t v; memcpy(&v, p, 8); return v; }

typedef struct {
  const uint8_t* base;
  int64_t size;
} fb_buf;

/* A table is identified by its absolute position inside the buffer; <0 = absent/invalid. */
static int64_t fb_root(const fb_buf* b) {
  if (b->size < 4) return -1;
  uint32_t off = rd_u32(b->base);
  if ((int64_t)off + 4 > b->size) return -1;
  return off;
}

/* Absolute position of field `id` inside table `t`, or -1 when the field is absent. */
static int64_t fb_field(const fb_buf* b, int64_t t, int id) {
  if (t < 0 || t + 4 > b->size) return -1;
  int64_t vt = t - rd_i32(b->base + t);
  if (vt < 0 || vt + 4 > b->size) return -1;
  uint16_t vt_size = rd_u16(b->base + vt);
  int64_t slot = 4 + 2 * (int64_t)id;
  if (slot + 2 > vt_size || vt + slot + 2 > b->size) return -1;
  uint16_t off = rd_u16(b->base + vt + slot);
  if (off == 0) return -1;
  if (t + off >= b->size) return -1;
  return t + off;
}

static int64_t fb_indirect(const fb_buf* b, int64_t pos) {
  if (pos < 0 || pos + 4 > b->size) return -1;
  int64_t tgt = pos + rd_u32(b->base + pos);
  if (tgt + 4 > b->size) return -1;
  return tgt;
}

static int64_t fb_table_field(const fb_buf* b, int64_t t, int id) { return fb_indirect(b, fb_field(b, t, id)); }

static int64_t fb_i64(const fb_buf* b, int64_t t, int id, int64_t dflt) {
  int64_t p = fb_field(b, t, id);
  return (p < 0 || p + 8 > b->size) ? dflt : rd_i64(b->base + p);
}
static int32_t fb_i32(const fb_buf* b, int64_t t, int id, int32_t dflt) {
  int64_t p = fb_field(b, t, id);
  return (p < 0 || p + 4 > b->size) ? dflt : rd_i32(b->base + p);
}
static int16_t fb_i16(const fb_buf* b, int64_t t, int id, int16_t dflt) {
  int64_t p = fb_field(b, t, id);
  return (p < 0 || p + 2 > b->size) ? dflt : rd_i16(b->base + p);
}
static uint8_t fb_u8(const fb_buf* b, int64_t t, int id, uint8_t dflt) {
  int64_t p = fb_field(b, t, id);
  return (p < 0) ? dflt : b->base[p];
}
/* vector field -> position of element 0 and element count */
static int64_t fb_vector(const fb_buf* b, int64_t t, int id, uint32_t* len) {
  int64_t v = fb_table_field(b, t, id);
  *len = 0;
  if (v < 0) return -1;
  *len = rd_u32(b->base + v);
  return v + 4;
}
static void fb_string(const fb_buf* b, int64_t t, int id, char* out, size_t cap) {
  out[0] = 0;
  uint32_t len;
  int64_t s = fb_vector(b, t, id, &len);
  if (s < 0 || s + len > b->size) return;
  if (len >= cap) len = (uint32_t)cap - 1;
  memcpy(out, b->base + s, len);
  out[len] = 0;
}

/* ---------------------------------------------------------------- framing */
int orc_walk_stream(const uint8_t* buf, int64_t size, orc_msg* out, int32_t max, int32_t* n_out, char* err,
                    int32_t err_cap) {
  int64_t pos = 0;
  int32_t n = 0;
  if (err && err_cap > 0) err[0] = 0;
  *n_out = 0;
  while (n < max) {
    /* EnsureInputStreamAligned: ipc_file_stream_reader.cpp:134-141 */
    pos = (pos + 7) & ~(int64_t)7;
    if (pos + 8 > size) break; /* SerializationException => finished, :126-129 */
    uint32_t token = rd_u32(buf + pos);
    int32_t meta_len = rd_i32(buf + pos + 4);
    /* file-format magic at the very start is skipped and the embedded stream is read, :116-119 */
    if (pos == 0 && memcmp(buf, "ARROW1\0\0", 8) == 0) {
      pos = 8;
      continue;
    }
    if (token != 0xFFFFFFFFu) {
      if (err) snprintf(err, (size_t)err_cap, "Expected continuation token (0xFFFFFFFF) but got %u", token);
      return ORC_EIO;
    }
    if (meta_len < 0) { /* base_stream_reader.cpp:222-225 */
      if (err) snprintf(err, (size_t)err_cap, "Expected metadata size >= 0 but got %d", meta_len);
      return ORC_EIO;
    }
    if (meta_len == 0) break; /* EOS: DecodeHeader returns ENODATA, ipc_file_stream_reader.cpp:63-66 */
    if (pos + 8 + meta_len > size) { /* the prefix was read, so DecodeMessage runs outside the try block (:131): error */
      if (err) snprintf(err, (size_t)err_cap, "not enough data in file to deserialize result");
      return ORC_EIO;
    }
    orc_msg m;
    memset(&m, 0, sizeof(m));
    m.prefix_off = pos;
    m.meta_off = pos + 8;
    m.meta_len = meta_len;
    int32_t version;
    int rc = orc_decode_message(buf + m.meta_off, meta_len, &m.type, &m.body_len, &version);
    if (rc != ORC_OK) {
      if (err) snprintf(err, (size_t)err_cap, "invalid Message flatbuffer at offset %lld", (long long)m.meta_off);
      return ORC_EIO;
    }
    pos = m.meta_off + meta_len;
    if (m.body_len > 0) {
      pos = (pos + 7) & ~(int64_t)7; /* DecodeBody aligns first, ipc_file_stream_reader.cpp:72-73 */
      if (pos + m.body_len > size) { /* BufferedFileReader::ReadData throws inside DecodeBody (:80) */
        if (err) snprintf(err, (size_t)err_cap, "not enough data in file to deserialize result");
        return ORC_EIO;
      }
    }
    m.body_off = pos;
    pos += m.body_len;
    out[n++] = m;
  }
  *n_out = n;
  return ORC_OK;
}

/* Message { version:short [0]; header_type:ubyte [1]; header:table [2]; bodyLength:long [3]; } */
int orc_decode_message(const uint8_t* meta, int32_t meta_len, int32_t* type, int64_t* body_len,
                       int32_t* version) {
  fb_buf b = {meta, meta_len};
  int64_t root = fb_root(&b);
  if (root < 0) return ORC_EINVAL;
  *version = fb_i16(&b, root, 0, 0);
  *type = fb_u8(&b, root, 1, 0);
  *body_len = fb_i64(&b, root, 3, 0);
  if (*type < ORC_MSG_SCHEMA || *type > 5) return ORC_EINVAL;
  if (fb_table_field(&b, root, 2) < 0) return ORC_EINVAL;
  if (*body_len < 0) return ORC_EINVAL;
  return ORC_OK;
}

/* Field { name [0]; nullable [1]; type_type [2]; type [3]; dictionary [4]; children [5]; custom_metadata [6]; } */
static int decode_field(const fb_buf* b, int64_t f, orc_field* out, int32_t max, int32_t* n) {
  if (*n >= max) return ORC_EINVAL;
  orc_field* o = &out[(*n)++];
  memset(o, 0, sizeof(*o));
  fb_string(b, f, 0, o->name, sizeof(o->name));
  o->nullable = fb_u8(b, f, 1, 0);
  o->type = fb_u8(b, f, 2, 0);
  int64_t t = fb_table_field(b, f, 3);
  switch (o->type) {
    case ORC_T_INT: /* Int { bitWidth:int [0]; is_signed:bool [1]; } */
      o->bit_width = fb_i32(b, t, 0, 0);
      o->is_signed = fb_u8(b, t, 1, 0);
      break;
    case ORC_T_FLOAT: /* FloatingPoint { precision:short [0] } */
      o->precision = fb_i16(b, t, 0, 0);
      break;
    case ORC_T_DECIMAL: /* Decimal { precision:int [0]; scale:int [1]; bitWidth:int=128 [2] } */
      o->precision = fb_i32(b, t, 0, 0);
      o->scale = fb_i32(b, t, 1, 0);
      o->bit_width = fb_i32(b, t, 2, 128);
      break;
    case ORC_T_DATE: /* Date { unit:short = MILLISECOND [0] } */
      o->unit = fb_i16(b, t, 0, 1);
      break;
    case ORC_T_TIME: /* Time { unit:short = MILLISECOND [0]; bitWidth:int = 32 [1] } */
      o->unit = fb_i16(b, t, 0, 1);
      o->bit_width = fb_i32(b, t, 1, 32);
      break;
    case ORC_T_TIMESTAMP: /* Timestamp { unit:short [0]; timezone:string [1] } */
      o->unit = fb_i16(b, t, 0, 0);
      fb_string(b, t, 1, o->tz, sizeof(o->tz));
      break;
    case ORC_T_DURATION: /* Duration { unit:short = MILLISECOND [0] } */
      o->unit = fb_i16(b, t, 0, 1);
      break;
    case ORC_T_INTERVAL: /* Interval { unit:short [0] } */
      o->unit = fb_i16(b, t, 0, 0);
      break;
    case ORC_T_FIXED_BINARY: /* FixedSizeBinary { byteWidth:int [0] } */
      o->byte_width = fb_i32(b, t, 0, 0);
      break;
    case ORC_T_FIXED_LIST: /* FixedSizeList { listSize:int [0] } */
      o->byte_width = fb_i32(b, t, 0, 0);
      break;
    default:
      break;
  }
  /* DictionaryEncoding { id:long [0]; indexType:Int [1]; isOrdered [2]; dictionaryKind [3] } */
  int64_t d = fb_table_field(b, f, 4);
  if (d >= 0) {
    o->has_dict = 1;
    o->dict_id = fb_i64(b, d, 0, 0);
    int64_t it = fb_table_field(b, d, 1);
    o->dict_index_bit_width = it >= 0 ? fb_i32(b, it, 0, 32) : 32;
    o->dict_index_signed = it >= 0 ? fb_u8(b, it, 1, 1) : 1;
  }
  uint32_t nchild;
  int64_t cv = fb_vector(b, f, 5, &nchild);
  o->n_children = (int32_t)nchild;
  for (uint32_t i = 0; i < nchild; i++) {
    int64_t c = fb_indirect(b, cv + 4 * (int64_t)i);
    if (c < 0) return ORC_EINVAL;
    int rc = decode_field(b, c, out, max, n);
    if (rc) return rc;
  }
  return ORC_OK;
}

/* Schema { endianness:short [0]; fields:[Field] [1]; custom_metadata [2]; features [3] } */
int orc_decode_schema(const uint8_t* meta, int32_t meta_len, orc_field* out, int32_t max, int32_t* n_out,
                      int32_t* n_top_level, int32_t* endianness) {
  fb_buf b = {meta, meta_len};
  int64_t root = fb_root(&b);
  if (root < 0 || fb_u8(&b, root, 1, 0) != ORC_MSG_SCHEMA) return ORC_EINVAL;
  int64_t s = fb_table_field(&b, root, 2);
  if (s < 0) return ORC_EINVAL;
  *endianness = fb_i16(&b, s, 0, 0);
  uint32_t nf;
  int64_t fv = fb_vector(&b, s, 1, &nf);
  *n_top_level = (int32_t)nf;
  *n_out = 0;
  for (uint32_t i = 0; i < nf; i++) {
    int64_t f = fb_indirect(&b, fv + 4 * (int64_t)i);
    if (f < 0) return ORC_EINVAL;
    int rc = decode_field(&b, f, out, max, n_out);
    if (rc) return rc;
  }
  return ORC_OK;
}

/* RecordBatch { length [0]; nodes:[FieldNode] [1]; buffers:[Buffer] [2]; compression [3]; variadicBufferCounts [4] }
 * DictionaryBatch { id [0]; data:RecordBatch [1]; isDelta [2] } */
int orc_decode_record_batch(const uint8_t* meta, int32_t meta_len, int64_t* length, orc_node* nodes,
                            int32_t max_nodes, int32_t* n_nodes, orc_buf* bufs, int32_t max_bufs,
                            int32_t* n_bufs, int32_t* compression, int64_t* dict_id, int32_t* is_delta) {
  fb_buf b = {meta, meta_len};
  int64_t root = fb_root(&b);
  if (root < 0) return ORC_EINVAL;
  int type = fb_u8(&b, root, 1, 0);
  int64_t rb = fb_table_field(&b, root, 2);
  if (rb < 0) return ORC_EINVAL;
  *dict_id = -1;
  *is_delta = 0;
  if (type == ORC_MSG_DICTIONARY_BATCH) {
    *dict_id = fb_i64(&b, rb, 0, 0);
    *is_delta = fb_u8(&b, rb, 2, 0);
    rb = fb_table_field(&b, rb, 1);
    if (rb < 0) return ORC_EINVAL;
  } else if (type != ORC_MSG_RECORD_BATCH) {
    return ORC_EINVAL;
  }
  *length = fb_i64(&b, rb, 0, 0);
  uint32_t nn, nb;
  int64_t nv = fb_vector(&b, rb, 1, &nn);
  int64_t bv = fb_vector(&b, rb, 2, &nb);
  if ((int32_t)nn > max_nodes || (int32_t)nb > max_bufs) return ORC_EINVAL;
  if (nn && (nv < 0 || nv + 16 * (int64_t)nn > b.size)) return ORC_EINVAL;
  if (nb && (bv < 0 || bv + 16 * (int64_t)nb > b.size)) return ORC_EINVAL;
  for (uint32_t i = 0; i < nn; i++) {
    nodes[i].length = rd_i64(b.base + nv + 16 * (int64_t)i);
    nodes[i].null_count = rd_i64(b.base + nv + 16 * (int64_t)i + 8);
  }
  for (uint32_t i = 0; i < nb; i++) {
    bufs[i].offset = rd_i64(b.base + bv + 16 * (int64_t)i);
    bufs[i].length = rd_i64(b.base + bv + 16 * (int64_t)i + 8);
  }
  *n_nodes = (int32_t)nn;
  *n_bufs = (int32_t)nb;
  /* BodyCompression { codec:byte = LZ4_FRAME [0]; method:byte [1] } */
  int64_t c = fb_table_field(&b, rb, 3);
  *compression = c < 0 ? -1 : (int32_t)(int8_t)fb_u8(&b, c, 0, 0);
  return ORC_OK;
}

/* RecordBatch.variadicBufferCounts [4]: one entry per utf8_view / binary_view node, depth-first */
int orc_decode_variadic_counts(const uint8_t* meta, int32_t meta_len, int64_t* out, int32_t max, int32_t* n_out) {
  fb_buf b = {meta, meta_len};
  int64_t root = fb_root(&b);
  *n_out = 0;
  if (root < 0) return ORC_EINVAL;
  int64_t rb = fb_table_field(&b, root, 2);
  if (fb_u8(&b, root, 1, 0) == ORC_MSG_DICTIONARY_BATCH) rb = fb_table_field(&b, rb, 1);
  if (rb < 0) return ORC_EINVAL;
  uint32_t n;
  int64_t v = fb_vector(&b, rb, 4, &n);
  if (n && (v < 0 || v + 8 * (int64_t)n > b.size || (int32_t)n > max)) return ORC_EINVAL;
  for (uint32_t i = 0; i < n; i++) out[i] = rd_i64(b.base + v + 8 * (int64_t)i);
  *n_out = (int32_t)n;
  return ORC_OK;
}

/* File layout: "ARROW1\0\0" stream... footer  int32 footer_len "ARROW1".
 * Footer { version [0]; schema [1]; dictionaries:[Block] [2]; recordBatches:[Block] [3] }
 * Block struct { offset:long; metaDataLength:int; pad 4; bodyLength:long } = 24 bytes */
int orc_decode_footer(const uint8_t* file, int64_t size, int64_t* blocks3, int32_t max_blocks,
                      int32_t* n_blocks, int32_t* n_dict_blocks) {
  if (size < 8 + 10 || memcmp(file, "ARROW1\0\0", 8) != 0 || memcmp(file + size - 6, "ARROW1", 6) != 0)
    return ORC_EINVAL;
  int32_t flen = rd_i32(file + size - 10);
  if (flen <= 0 || (int64_t)flen + 10 + 8 > size) return ORC_EINVAL;
  fb_buf b = {file + size - 10 - flen, flen};
  int64_t root = fb_root(&b);
  if (root < 0) return ORC_EINVAL;
  uint32_t nd, nr;
  (void)fb_vector(&b, root, 2, &nd);
  int64_t rv = fb_vector(&b, root, 3, &nr);
  if ((int32_t)nr > max_blocks) return ORC_EINVAL;
  if (nr && (rv < 0 || rv + 24 * (int64_t)nr > b.size)) return ORC_EINVAL;
  for (uint32_t i = 0; i < nr; i++) {
    const uint8_t* p = b.base + rv + 24 * (int64_t)i;
    blocks3[3 * i + 0] = rd_i64(p);
    blocks3[3 * i + 1] = rd_i32(p + 8);
    blocks3[3 * i + 2] = rd_i64(p + 16);
  }
  *n_blocks = (int32_t)nr;
  *n_dict_blocks = (int32_t)nd;
  return ORC_OK;
}

int orc_validate_offsets32(const int32_t* off, int64_t n, int64_t data_len) {
  if (n < 0) return ORC_EINVAL;
  if (off[0] < 0) return ORC_EINVAL;
  for (int64_t i = 0; i < n; i++)
    if (off[i + 1] < off[i]) return ORC_EINVAL;
  if ((int64_t)off[n] > data_len) return ORC_EINVAL;
  return ORC_OK;
}

int orc_validate_offsets64(const int64_t* off, int64_t n, int64_t data_len) {
  if (n < 0) return ORC_EINVAL;
  if (off[0] < 0) return ORC_EINVAL;
  for (int64_t i = 0; i < n; i++)
    if (off[i + 1] < off[i]) return ORC_EINVAL;
  if (off[n] > data_len) return ORC_EINVAL;
  return ORC_OK;
}
