/*
 * oracle_transcode.c -- CPU restatement of the per-value Arrow <-> DuckDB-vector transcode
 * (TEST INFRASTRUCTURE, see oracle.h).
 *
 * The arithmetic lives in DuckDB core, which the reference only calls:
 *   decode  ArrowTableFunction::ArrowScanFunction -> ArrowToDuckDB -> ColumnArrowToDuckDB
 *           (call sites src/scanner/scan_arrow_ipc.cpp:56, src/file_scanner/arrow_file_scan.cpp:68-72)
 *   encode  ArrowConverter::ToArrowArray / ArrowAppender
 *           (call sites src/writer/column_data_collection_serializer.cpp:85, src/writer/to_arrow_ipc.cpp:134-141)
 * duckdb/ is an empty submodule (.gitmodules:1-4; CI pins v1.2.1 8e52ec43 and main ~v1.3.0,
 * .github/workflows/MainDistributionPipeline.yml:19,26-29), so each function below restates the published
 * behaviour of that version as summarised in SURVEY.md section 2.3 (K1..K7) and Appendix C.
 *
 * Layouts reproduced: validity_t = uint64 words, bit = 1 valid, LSB first; string_t = 16 bytes
 * {u32 len; len<=12 ? 12 inline bytes zero padded : 4 byte prefix + 8 byte pointer}; hugeint_t =
 * {u64 lower; i64 upper}; interval_t = {i32 months; i32 days; i64 micros}; sel_t = u32; bool = 1 byte.
 */
#include "oracle.h"

#include <string.h>

static inline int bit_get(const uint8_t* bits, int64_t i) { return (bits[i >> 3] >> (i & 7)) & 1; }
static inline int word_valid(const uint64_t* v, int64_t i) { return v == NULL || ((v[i >> 6] >> (i & 63)) & 1); }

/* ------------------------------------------------------------------------------------------------ K1 */
/* GetValidityMask: copied only when null_count != 0 and a bitmap exists; byte-aligned offsets are a
 * memcpy, others copy ceil(n/8)+1 bytes and shift right by o%8 across bytes.  The mask starts all-valid
 * (EnsureWritable), so untouched bits are 1; bits >= n in the last copied byte are canonicalised to 1. */
void orc_validity(const uint8_t* bitmap, int64_t null_count, int64_t o, int64_t n, uint64_t* out) {
  int64_t nwords = (n + 63) / 64;
  if (nwords == 0) return;
  memset(out, 0xFF, (size_t)nwords * 8);
  if (null_count == 0 || bitmap == NULL) return;
  uint8_t* ob = (uint8_t*)out;
  int64_t nbytes = (n + 7) / 8;
  int shift = (int)(o & 7);
  const uint8_t* src = bitmap + (o >> 3);
  if (shift == 0) {
    memcpy(ob, src, (size_t)nbytes);
  } else {
    /* equivalent of ShiftRight(temp, nbytes+1, shift) followed by memcpy(nbytes); the extra byte is only
     * read when some requested bit lives in it */
    int64_t last_bit = o + n - 1;
    int64_t last_src_byte = last_bit >> 3;
    for (int64_t i = 0; i < nbytes; i++) {
      int64_t b = (o >> 3) + i;
      uint8_t lo = bitmap[b];
      uint8_t hi = (b + 1 <= last_src_byte) ? bitmap[b + 1] : 0xFF;
      ob[i] = (uint8_t)((lo >> shift) | (hi << (8 - shift)));
    }
  }
  /* canonical pad bits */
  if (n & 63) out[nwords - 1] |= ~(uint64_t)0 << (n & 63);
}

/* ------------------------------------------------------------------------------------------------ K2 */
void orc_bool(const uint8_t* bits, int64_t o, int64_t n, uint8_t* out) {
  for (int64_t i = 0; i < n; i++) out[i] = (uint8_t)bit_get(bits, o + i);
}

/* ------------------------------------------------------------------------------------------------ K3a */
const uint8_t* orc_direct(const uint8_t* data, int32_t width, int64_t o) { return data + (int64_t)width * o; }

/* ------------------------------------------------------------------------------------------------ K3b */
/* Hugeint::TryCast on valid rows; the value is in range by the declared precision so the result is the
 * low bytes.  Null rows are skipped upstream; canonical = 0. */
void orc_decimal128_narrow(const uint8_t* data, const uint64_t* valid, int64_t o, int64_t n, int32_t out_width,
                           void* out) {
  for (int64_t i = 0; i < n; i++) {
    uint64_t lower = 0;
    if (word_valid(valid, i)) memcpy(&lower, data + 16 * (o + i), 8);
    switch (out_width) {
      case 2: ((int16_t*)out)[i] = (int16_t)lower; break;
      case 4: ((int32_t*)out)[i] = (int32_t)lower; break;
      default: ((int64_t*)out)[i] = (int64_t)lower; break;
    }
  }
}

/* ------------------------------------------------------------------------------------------------ K3c */
void orc_date64_to_date32(const int64_t* src, int64_t o, int64_t n, int32_t* out) {
  for (int64_t i = 0; i < n; i++) out[i] = (int32_t)(src[o + i] / (int64_t)(1000 * 60 * 60 * 24));
}

int orc_mul_i32_to_i64(const int32_t* src, const uint64_t* valid, int64_t o, int64_t n, int64_t factor,
                       int64_t* out) {
  int rc = ORC_OK;
  for (int64_t i = 0; i < n; i++) {
    if (!word_valid(valid, i)) { out[i] = 0; continue; }
    int64_t r;
    if (__builtin_mul_overflow((int64_t)src[o + i], factor, &r)) { rc = ORC_EINVAL; r = 0; }
    out[i] = r;
  }
  return rc;
}

int orc_mul_i64(const int64_t* src, const uint64_t* valid, int64_t o, int64_t n, int64_t factor, int64_t* out) {
  int rc = ORC_OK;
  for (int64_t i = 0; i < n; i++) {
    if (!word_valid(valid, i)) { out[i] = 0; continue; }
    int64_t r;
    if (__builtin_mul_overflow(src[o + i], factor, &r)) { rc = ORC_EINVAL; r = 0; }
    out[i] = r;
  }
  return rc;
}

void orc_div_i64(const int64_t* src, int64_t o, int64_t n, int64_t divisor, int64_t* out) {
  for (int64_t i = 0; i < n; i++) out[i] = src[o + i] / divisor;
}

int orc_duration_to_interval(const int64_t* src, const uint64_t* valid, int64_t o, int64_t n, int64_t factor,
                             uint8_t* out16) {
  int rc = ORC_OK;
  for (int64_t i = 0; i < n; i++) {
    int64_t micros = 0;
    if (factor < 0) {
      micros = src[o + i] / (-factor);
    } else if (word_valid(valid, i)) {
      if (__builtin_mul_overflow(src[o + i], factor, &micros)) { rc = ORC_EINVAL; micros = 0; }
    }
    memset(out16 + 16 * i, 0, 8); /* months = 0, days = 0 */
    memcpy(out16 + 16 * i + 8, &micros, 8);
  }
  return rc;
}

void orc_interval_months(const int32_t* src, int64_t o, int64_t n, uint8_t* out16) {
  for (int64_t i = 0; i < n; i++) {
    memset(out16 + 16 * i, 0, 16);
    memcpy(out16 + 16 * i, &src[o + i], 4);
  }
}

void orc_interval_mdn(const uint8_t* src16, int64_t o, int64_t n, uint8_t* out16) {
  for (int64_t i = 0; i < n; i++) {
    int64_t nanos;
    memcpy(out16 + 16 * i, src16 + 16 * (o + i), 8); /* months, days */
    memcpy(&nanos, src16 + 16 * (o + i) + 8, 8);
    int64_t micros = nanos / 1000;
    memcpy(out16 + 16 * i + 8, &micros, 8);
  }
}

/* decimal32 / decimal64 inputs (DuckDB v1.3): TryCast to the physical type of the declared precision on valid rows */
void orc_narrow(const void* src, int32_t src_width, const uint64_t* valid, int64_t o, int64_t n, int32_t dst_width, void* out) {
  for (int64_t i = 0; i < n; i++) {
    int64_t v = 0;
    if (word_valid(valid, i)) v = src_width == 4 ? ((const int32_t*)src)[o + i] : ((const int64_t*)src)[o + i];
    switch (dst_width) {
      case 2: ((int16_t*)out)[i] = (int16_t)v; break;
      case 4: ((int32_t*)out)[i] = (int32_t)v; break;
      default: ((int64_t*)out)[i] = v; break;
    }
  }
}

/* IEEE 754 binary16 -> binary32 (exact): subnormals are normalised, inf/nan keep their payload */
void orc_half_to_float(const uint16_t* src, int64_t o, int64_t n, uint32_t* out_bits) {
  for (int64_t i = 0; i < n; i++) {
    uint32_t h = src[o + i];
    uint32_t sign = (h & 0x8000u) << 16, exp = (h >> 10) & 0x1F, man = h & 0x3FF, f;
    if (exp == 0) {
      if (man == 0) f = sign;
      else {
        int e = -1;
        do { e++; man <<= 1; } while ((man & 0x400) == 0);
        f = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FF) << 13);
      }
    } else if (exp == 31) {
      f = sign | 0x7F800000u | (man << 13);
    } else {
      f = sign | ((exp + 112) << 23) | (man << 13);
    }
    out_bits[i] = f;
  }
}

/* ------------------------------------------------------------------------------------------------ K4 */
static inline void make_string_t(uint8_t* dst, const uint8_t* payload, uint32_t len, uint64_t ptr) {
  memcpy(dst, &len, 4);
  if (len <= 12) {
    memset(dst + 4, 0, 12);
    if (len) memcpy(dst + 4, payload, len);
  } else {
    memcpy(dst + 4, payload, 4);
    memcpy(dst + 8, &ptr, 8);
  }
}

/* SetVectorString<int32_t>: valid rows only; null rows canonical = 16 zero bytes. */
int orc_string32(const int32_t* off, const uint8_t* data, const uint64_t* valid, int64_t o, int64_t n,
                 uint64_t ptr_base, uint8_t* out16) {
  for (int64_t i = 0; i < n; i++) {
    uint8_t* dst = out16 + 16 * i;
    if (!word_valid(valid, i)) { memset(dst, 0, 16); continue; }
    int32_t a = off[o + i], b = off[o + i + 1];
    uint32_t len = (uint32_t)(b - a);
    make_string_t(dst, data + a, len, ptr_base + (uint64_t)(int64_t)a);
  }
  return ORC_OK;
}

/* SetVectorString<int64_t>: "DuckDB does not support Strings over 4GB" when a length (or, for the whole
 * array, the last offset) exceeds UINT32_MAX. */
int orc_string64(const int64_t* off, const uint8_t* data, const uint64_t* valid, int64_t o, int64_t n,
                 uint64_t ptr_base, uint8_t* out16) {
  int rc = ORC_OK;
  if (off[o + n] > (int64_t)UINT32_MAX) rc = ORC_EINVAL;
  for (int64_t i = 0; i < n; i++) {
    uint8_t* dst = out16 + 16 * i;
    if (!word_valid(valid, i)) { memset(dst, 0, 16); continue; }
    int64_t a = off[o + i], b = off[o + i + 1];
    if (b - a > (int64_t)UINT32_MAX) { rc = ORC_EINVAL; memset(dst, 0, 16); continue; }
    make_string_t(dst, data + a, (uint32_t)(b - a), ptr_base + (uint64_t)a);
  }
  return rc;
}

void orc_fixed_binary(const uint8_t* data, int32_t width, const uint64_t* valid, int64_t o, int64_t n,
                      uint64_t ptr_base, uint8_t* out16) {
  for (int64_t i = 0; i < n; i++) {
    uint8_t* dst = out16 + 16 * i;
    if (!word_valid(valid, i)) { memset(dst, 0, 16); continue; }
    int64_t a = (o + i) * (int64_t)width;
    make_string_t(dst, data + a, (uint32_t)width, ptr_base + (uint64_t)a);
  }
}

int orc_list_entries(const void* off, int32_t off_width, int64_t o, int64_t n, int64_t win_row, int64_t child_len, uint8_t* out16) {
  int rc = ORC_OK;
  for (int64_t i = 0; i < n; i++) {
    int64_t a, b, base;
    if (off_width == 4) {
      a = ((const int32_t*)off)[o + i]; b = ((const int32_t*)off)[o + i + 1]; base = ((const int32_t*)off)[win_row];
    } else {
      a = ((const int64_t*)off)[o + i]; b = ((const int64_t*)off)[o + i + 1]; base = ((const int64_t*)off)[win_row];
    }
    if (a < 0 || b < a || b > child_len || a < base) rc = ORC_EINVAL;
    uint64_t le[2] = {(uint64_t)(a - base), (uint64_t)(b - a)};
    memcpy(out16 + 16 * i, le, 16);
  }
  return rc;
}

int orc_string_view(const uint8_t* views16, const uint64_t* valid, int64_t o, int64_t n, const uint64_t* table, int64_t n_buffers,
                    uint8_t* out16) {
  int rc = ORC_OK;
  for (int64_t i = 0; i < n; i++) {
    uint8_t* dst = out16 + 16 * i;
    memset(dst, 0, 16);
    if (!word_valid(valid, i)) continue;
    const uint8_t* v = views16 + 16 * (o + i);
    uint32_t len;
    memcpy(&len, v, 4);
    if (len <= 12) {
      memcpy(dst, &len, 4);
      memcpy(dst + 4, v + 4, len);
    } else {
      int32_t bi, bo;
      memcpy(&bi, v + 8, 4);
      memcpy(&bo, v + 12, 4);
      if (bi < 0 || bi >= n_buffers || bo < 0 || (uint64_t)bo + len > table[2 * bi + 1]) { rc = ORC_EINVAL; continue; }
      uint64_t p = table[2 * bi] + (uint64_t)bo;
      memcpy(dst, &len, 4);
      memcpy(dst + 4, v + 4, 4);
      memcpy(dst + 8, &p, 8);
    }
  }
  return rc;
}

/* ------------------------------------------------------------------------------------------------ K5 */
/* SetSelectionVector: sel[i] = valid ? idx[i] : dict_len (the extra NULL entry appended to the decoded
 * dictionary); "DuckDB only supports indices that fit on an uint32" for wider out-of-range values. */
int orc_dict_sel(const void* idx, int32_t idx_width, int32_t idx_signed, const uint64_t* valid, int64_t o,
                 int64_t n, uint32_t dict_len, uint32_t* sel) {
  int rc = ORC_OK;
  for (int64_t i = 0; i < n; i++) {
    if (!word_valid(valid, i)) { sel[i] = dict_len; continue; }
    int64_t r = o + i;
    uint64_t v;
    switch (idx_width) {
      case 1: v = idx_signed ? (uint64_t)(int64_t)((const int8_t*)idx)[r] : ((const uint8_t*)idx)[r]; break;
      case 2: v = idx_signed ? (uint64_t)(int64_t)((const int16_t*)idx)[r] : ((const uint16_t*)idx)[r]; break;
      case 4: v = idx_signed ? (uint64_t)(int64_t)((const int32_t*)idx)[r] : ((const uint32_t*)idx)[r]; break;
      default: v = ((const uint64_t*)idx)[r]; break;
    }
    /* an index that does not fit uint32 ("DuckDB only supports indices that fit on an uint32") or points past the
     * dictionary is an error; the slot then selects the dictionary's NULL entry */
    if (v > (uint64_t)UINT32_MAX || v >= (uint64_t)dict_len) { rc = ORC_EINVAL; v = dict_len; }
    sel[i] = (uint32_t)v;
  }
  return rc;
}

/* ------------------------------------------------------------------------------------------------ K6 */
int64_t orc_filter_range_i32(const int32_t* v, const uint64_t* valid, int64_t n, int32_t lo, int32_t hi,
                             uint32_t* sel) {
  int64_t c = 0;
  for (int64_t i = 0; i < n; i++)
    if (word_valid(valid, i) && v[i] >= lo && v[i] < hi) sel[c++] = (uint32_t)i;
  return c;
}

int64_t orc_filter_range_i64(const int64_t* v, const uint64_t* valid, int64_t n, int64_t lo, int64_t hi,
                             uint32_t* sel) {
  int64_t c = 0;
  for (int64_t i = 0; i < n; i++)
    if (word_valid(valid, i) && v[i] >= lo && v[i] < hi) sel[c++] = (uint32_t)i;
  return c;
}

/* Predicate trees in conjunctive normal form over decoded fixed-width vectors: the rows DuckDB's own filter above the scan
 * would keep (the reference sets filter_pushdown = false, src/scanner/read_arrow.cpp:47-48; SQL semantics: a comparison
 * with NULL is not true).  Leaves are evaluated with the SQL operator itself (no range normalisation as in the product),
 * one row at a time.  op: 1 = 2 <> 3 < 4 <= 5 > 6 >= 7 IS NULL 8 IS NOT NULL 9 IN; ends_clause marks the last leaf of an
 * OR group; the filter is the AND of the groups. */
static int64_t leaf_value(const orc_filter_leaf* l, int64_t i) {
  switch (l->width) {
    case 1: return l->is_unsigned ? (int64_t)((const uint8_t*)l->data)[i] : (int64_t)((const int8_t*)l->data)[i];
    case 2: return l->is_unsigned ? (int64_t)((const uint16_t*)l->data)[i] : (int64_t)((const int16_t*)l->data)[i];
    case 4: return l->is_unsigned ? (int64_t)((const uint32_t*)l->data)[i] : (int64_t)((const int32_t*)l->data)[i];
    default: return ((const int64_t*)l->data)[i];
  }
}

static int leaf_true(const orc_filter_leaf* l, int64_t i) {
  const int valid = word_valid(l->validity, i);
  if (l->op == 7) return !valid;
  if (l->op == 8) return valid;
  if (!valid) return 0;
  const int64_t v = leaf_value(l, i), c = l->value;
  if (l->is_unsigned && l->width == 8) { /* uint64 column: unsigned order, negative constants lie below every value */
    const uint64_t uv = (uint64_t)v;
    if (l->op == 9) {
      for (int32_t k = 0; k < l->n_values; k++)
        if (l->values[k] >= 0 && (uint64_t)l->values[k] == uv) return 1;
      return 0;
    }
    if (c < 0) return l->op == 2 || l->op == 5 || l->op == 6;
    const uint64_t uc = (uint64_t)c;
    switch (l->op) {
      case 1: return uv == uc;
      case 2: return uv != uc;
      case 3: return uv < uc;
      case 4: return uv <= uc;
      case 5: return uv > uc;
      default: return uv >= uc;
    }
  }
  switch (l->op) {
    case 1: return v == c;
    case 2: return v != c;
    case 3: return v < c;
    case 4: return v <= c;
    case 5: return v > c;
    case 6: return v >= c;
    case 9:
      for (int32_t k = 0; k < l->n_values; k++)
        if (l->values[k] == v) return 1;
      return 0;
    default: return 0;
  }
}

int64_t orc_filter_cnf(const orc_filter_leaf* leaves, int32_t n_leaves, int64_t n, uint32_t* sel) {
  int64_t c = 0;
  for (int64_t i = 0; i < n; i++) {
    int keep = 1, group = 0;
    for (int32_t k = 0; k < n_leaves; k++) {
      group |= leaf_true(&leaves[k], i);
      if (leaves[k].ends_clause) {
        keep &= group;
        group = 0;
      }
    }
    if (keep) sel[c++] = (uint32_t)i;
  }
  return c;
}

/* ------------------------------------------------------------------------------------------------ K7 */
/* ArrowAppendData::AppendValidity: the buffer was resized with 0xFF; clear the bit of every NULL. */
void orc_enc_validity(const uint64_t* valid, int64_t n, int64_t row0, uint8_t* bitmap, int64_t* null_count) {
  if (valid == NULL) return;
  for (int64_t i = 0; i < n; i++) {
    if (!word_valid(valid, i)) {
      int64_t r = row0 + i;
      bitmap[r >> 3] &= (uint8_t)~(1u << (r & 7));
      (*null_count)++;
    }
  }
}

/* ArrowScalarData<hugeint_t, intN>: sign extension to {u64 lower, i64 upper}. */
void orc_enc_decimal_widen(const void* src, int32_t in_width, int64_t n, uint8_t* out16) {
  for (int64_t i = 0; i < n; i++) {
    int64_t v;
    switch (in_width) {
      case 2: v = ((const int16_t*)src)[i]; break;
      case 4: v = ((const int32_t*)src)[i]; break;
      default: v = ((const int64_t*)src)[i]; break;
    }
    int64_t upper = v < 0 ? -1 : 0;
    memcpy(out16 + 16 * i, &v, 8);
    memcpy(out16 + 16 * i + 8, &upper, 8);
  }
}

/* ArrowBoolData::Append: data bits start as 1; a valid false clears its bit; NULL rows keep 1. */
void orc_enc_bool(const uint8_t* src, const uint64_t* valid, int64_t n, int64_t row0, uint8_t* bits) {
  for (int64_t i = 0; i < n; i++) {
    if (word_valid(valid, i) && !src[i]) {
      int64_t r = row0 + i;
      bits[r >> 3] &= (uint8_t)~(1u << (r & 7));
    }
  }
}

/* ArrowVarcharData<int32_t>::AppendTemplated<false>. */
int orc_enc_varchar32(const uint8_t* str16, const uint64_t* valid, int64_t n, int64_t row0, uint64_t ptr_base,
                      const uint8_t* heap, int32_t* off, uint8_t* data) {
  if (row0 == 0) off[0] = 0;
  int64_t last = off[row0];
  for (int64_t i = 0; i < n; i++) {
    const uint8_t* s = str16 + 16 * i;
    if (!word_valid(valid, i)) { off[row0 + i + 1] = (int32_t)last; continue; }
    uint32_t len;
    memcpy(&len, s, 4);
    int64_t cur = last + (int64_t)len;
    if (cur > (int64_t)INT32_MAX) return ORC_EINVAL; /* "SET arrow_large_buffer_size=true ..." */
    off[row0 + i + 1] = (int32_t)cur;
    if (len <= 12) {
      memcpy(data + last, s + 4, len);
    } else {
      uint64_t p;
      memcpy(&p, s + 8, 8);
      memcpy(data + last, heap + (p - ptr_base), len);
    }
    last = cur;
  }
  return ORC_OK;
}

/* ------------------------------------------------------------------------------- whole-column driver */
int32_t orc_out_width(int32_t kind, int64_t param) {
  switch (kind) {
    case ORC_K_COPY: return (int32_t)param;
    case ORC_K_BOOL: return 1;
    case ORC_K_DEC128: return (int32_t)param;
    case ORC_K_DATE64: return 4;
    case ORC_K_MUL_I32: case ORC_K_MUL_I64: case ORC_K_DIV_I64: return 8;
    case ORC_K_STR32: case ORC_K_STR64: case ORC_K_FIXED_BINARY: case ORC_K_DURATION: return 16;
    case ORC_K_INTERVAL_MONTHS: case ORC_K_INTERVAL_MDN: return 16;
    case ORC_K_NARROW: return (int32_t)((param >> 8) & 0xFF);
    case ORC_K_HALF_FLOAT: return 4;
    case ORC_K_NULL: return 1;
    case ORC_K_STRVIEW: case ORC_K_LIST32: case ORC_K_LIST64: return 16;
    case ORC_K_STRUCT: return 0;
    case ORC_K_DICT: return 4;
    default: return 0;
  }
}

/* One call of ArrowScanFunction converts min(2048, remaining) rows of every column
 * (arrow_file_scan.cpp:68-72 -> ArrowToDuckDB); this is that loop for one column of one batch. */
int orc_decode_column(const orc_col_task* t, int32_t copy_direct) {
  int rc = ORC_OK;
  int32_t w = orc_out_width(t->kind, t->param);
  for (int64_t o = 0; o < t->nrows; o += ORC_VECTOR_SIZE) {
    int64_t n = t->nrows - o < ORC_VECTOR_SIZE ? t->nrows - o : ORC_VECTOR_SIZE;
    uint64_t* valid = t->out_validity + o / 64;
    orc_validity(t->validity, t->null_count, o, n, valid);
    if (t->kind == ORC_K_NULL) memset(valid, 0, (size_t)((n + 63) / 64) * 8);
    uint8_t* out = t->out_data + o * (int64_t)w;
    int r = ORC_OK;
    switch (t->kind) {
      case ORC_K_COPY:
        if (copy_direct) memcpy(out, orc_direct(t->buf1, w, o), (size_t)(n * w));
        break;
      case ORC_K_BOOL: orc_bool(t->buf1, o, n, out); break;
      case ORC_K_DEC128: orc_decimal128_narrow(t->buf1, valid, o, n, w, out); break;
      case ORC_K_DATE64: orc_date64_to_date32((const int64_t*)t->buf1, o, n, (int32_t*)out); break;
      case ORC_K_MUL_I32: r = orc_mul_i32_to_i64((const int32_t*)t->buf1, valid, o, n, t->param, (int64_t*)out); break;
      case ORC_K_MUL_I64: r = orc_mul_i64((const int64_t*)t->buf1, valid, o, n, t->param, (int64_t*)out); break;
      case ORC_K_DIV_I64: orc_div_i64((const int64_t*)t->buf1, o, n, t->param, (int64_t*)out); break;
      case ORC_K_DURATION: r = orc_duration_to_interval((const int64_t*)t->buf1, valid, o, n, t->param, out); break;
      case ORC_K_STR32: r = orc_string32((const int32_t*)t->buf1, t->buf2, valid, o, n, t->ptr_base, out); break;
      case ORC_K_STR64: r = orc_string64((const int64_t*)t->buf1, t->buf2, valid, o, n, t->ptr_base, out); break;
      case ORC_K_FIXED_BINARY: orc_fixed_binary(t->buf1, (int32_t)t->param, valid, o, n, t->ptr_base, out); break;
      case ORC_K_INTERVAL_MONTHS: orc_interval_months((const int32_t*)t->buf1, o, n, out); break;
      case ORC_K_INTERVAL_MDN: orc_interval_mdn(t->buf1, o, n, out); break;
      case ORC_K_NARROW: orc_narrow(t->buf1, (int32_t)(t->param & 0xFF), valid, o, n, w, out); break;
      case ORC_K_HALF_FLOAT: orc_half_to_float((const uint16_t*)t->buf1, o, n, (uint32_t*)out); break;
      case ORC_K_NULL: memset(out, 0, (size_t)n); break;
      case ORC_K_DICT:
        r = orc_dict_sel(t->buf1, (int32_t)(t->param & 0xFF), (int32_t)((t->param >> 8) & 1), valid, o, n,
                         (uint32_t)t->param2, (uint32_t*)out);
        break;
      default: r = ORC_ENOTSUP; break;
    }
    if (r != ORC_OK) rc = r;
  }
  return rc;
}

int orc_convert_column(const orc_col_task* t, const uint64_t* valid) {
  const int64_t n = t->nrows;
  const int32_t w = orc_out_width(t->kind, t->param);
  uint8_t* out = t->out_data;
  switch (t->kind) {
    case ORC_K_COPY: memcpy(out, t->buf1, (size_t)(n * w)); return ORC_OK;
    case ORC_K_BOOL: orc_bool(t->buf1, 0, n, out); return ORC_OK;
    case ORC_K_DEC128: orc_decimal128_narrow(t->buf1, valid, 0, n, w, out); return ORC_OK;
    case ORC_K_DATE64: orc_date64_to_date32((const int64_t*)t->buf1, 0, n, (int32_t*)out); return ORC_OK;
    case ORC_K_MUL_I32: return orc_mul_i32_to_i64((const int32_t*)t->buf1, valid, 0, n, t->param, (int64_t*)out);
    case ORC_K_MUL_I64: return orc_mul_i64((const int64_t*)t->buf1, valid, 0, n, t->param, (int64_t*)out);
    case ORC_K_DIV_I64: orc_div_i64((const int64_t*)t->buf1, 0, n, t->param, (int64_t*)out); return ORC_OK;
    case ORC_K_DURATION: return orc_duration_to_interval((const int64_t*)t->buf1, valid, 0, n, t->param, out);
    case ORC_K_INTERVAL_MONTHS: orc_interval_months((const int32_t*)t->buf1, 0, n, out); return ORC_OK;
    case ORC_K_INTERVAL_MDN: orc_interval_mdn(t->buf1, 0, n, out); return ORC_OK;
    case ORC_K_NARROW: orc_narrow(t->buf1, (int32_t)(t->param & 0xFF), valid, 0, n, w, out); return ORC_OK;
    case ORC_K_HALF_FLOAT: orc_half_to_float((const uint16_t*)t->buf1, 0, n, (uint32_t*)out); return ORC_OK;
    case ORC_K_NULL: memset(out, 0, (size_t)n); return ORC_OK;
    case ORC_K_STR32: return orc_string32((const int32_t*)t->buf1, t->buf2, valid, 0, n, t->ptr_base, out);
    case ORC_K_STR64: return orc_string64((const int64_t*)t->buf1, t->buf2, valid, 0, n, t->ptr_base, out);
    case ORC_K_FIXED_BINARY: orc_fixed_binary(t->buf1, (int32_t)t->param, valid, 0, n, t->ptr_base, out); return ORC_OK;
    case ORC_K_STRVIEW: return orc_string_view(t->buf1, valid, 0, n, (const uint64_t*)t->buf2, t->buf2_len, out);
    case ORC_K_DICT:
      return orc_dict_sel(t->buf1, (int32_t)(t->param & 0xFF), (int32_t)((t->param >> 8) & 1), valid, 0, n, (uint32_t)t->param2,
                          (uint32_t*)out);
    case ORC_K_STRUCT: return ORC_OK;
    default: return ORC_ENOTSUP;
  }
}
