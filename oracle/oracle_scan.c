/*
 * oracle_scan.c -- CPU restatement of the whole scan path for one in-memory IPC stream
 * (TEST INFRASTRUCTURE, see oracle.h).  This is what bench.py times as `cpu_baseline` (kind "port").
 *
 * Per RecordBatch message it performs, in the reference's order:
 *   1. body read = one allocation + one full copy     IPCFileStreamReader::DecodeBody
 *                                                     src/ipc/stream_reader/ipc_file_stream_reader.cpp:71-89
 *   2. buffer slicing + NANOARROW_VALIDATION_LEVEL_FULL (every offsets buffer walked)
 *                                                     IPCStreamReader::GetNextBatch
 *                                                     src/ipc/stream_reader/base_stream_reader.cpp:86-144
 *   3. the 2048-row pull loop, every column converted per window
 *                                                     ArrowFileScan::Scan src/file_scanner/arrow_file_scan.cpp:68-72
 * Single threaded like the reference's single-file scan (src/file_scanner/arrow_multi_file_info.cpp:77-86).
 */
#define _POSIX_C_SOURCE 199309L
#include "oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MAX_FIELDS 512
#define MAX_BUFS 2048

/* Arrow field -> (kind, param, number of IPC buffers).  DuckDB's arrow type mapping
 * (ArrowTableFunction::PopulateArrowTableType, called at arrow_file_scan.cpp:17-18). */
int orc_plan_column(const orc_field* f, int32_t* kind, int64_t* param, int32_t* n_buffers) {
  *param = 0;
  *n_buffers = 2;
  if (f->has_dict) {
    *kind = ORC_K_DICT;
    *param = (f->dict_index_bit_width / 8) | ((int64_t)(f->dict_index_signed ? 1 : 0) << 8);
    return ORC_OK;
  }
  switch (f->type) {
    case ORC_T_INT:
      *kind = ORC_K_COPY;
      *param = f->bit_width / 8;
      return ORC_OK;
    case ORC_T_NULL: *kind = ORC_K_NULL; *n_buffers = 0; return ORC_OK;
    case ORC_T_FLOAT:
      if (f->precision == 0) { *kind = ORC_K_HALF_FLOAT; return ORC_OK; }
      *kind = ORC_K_COPY;
      *param = f->precision == 1 ? 4 : 8;
      return ORC_OK;
    case ORC_T_BOOL: *kind = ORC_K_BOOL; return ORC_OK;
    case ORC_T_DECIMAL:
      if ((f->bit_width == 32 && f->precision <= 9) || (f->bit_width == 64 && f->precision <= 18)) {
        int32_t sw = f->bit_width / 8, dw = f->precision <= 4 ? 2 : f->precision <= 9 ? 4 : 8;
        if (sw == dw) { *kind = ORC_K_COPY; *param = sw; } else { *kind = ORC_K_NARROW; *param = sw | (dw << 8); }
        return ORC_OK;
      }
      if (f->bit_width != 128 || f->precision > 38) return ORC_ENOTSUP;
      if (f->precision <= 4) { *kind = ORC_K_DEC128; *param = 2; }
      else if (f->precision <= 9) { *kind = ORC_K_DEC128; *param = 4; }
      else if (f->precision <= 18) { *kind = ORC_K_DEC128; *param = 8; }
      else { *kind = ORC_K_COPY; *param = 16; }
      return ORC_OK;
    case ORC_T_DATE:
      if (f->unit == 0) { *kind = ORC_K_COPY; *param = 4; } else { *kind = ORC_K_DATE64; }
      return ORC_OK;
    case ORC_T_TIME:
      switch (f->unit) {
        case 0: *kind = ORC_K_MUL_I32; *param = 1000000; break;
        case 1: *kind = ORC_K_MUL_I32; *param = 1000; break;
        case 2: *kind = ORC_K_COPY; *param = 8; break;
        default: *kind = ORC_K_DIV_I64; *param = 1000; break;
      }
      return ORC_OK;
    case ORC_T_TIMESTAMP:
      if (f->tz[0] == 0) { *kind = ORC_K_COPY; *param = 8; return ORC_OK; } /* TIMESTAMP_S/MS/US/NS direct */
      switch (f->unit) {
        case 0: *kind = ORC_K_MUL_I64; *param = 1000000; break;
        case 1: *kind = ORC_K_MUL_I64; *param = 1000; break;
        case 2: *kind = ORC_K_COPY; *param = 8; break;
        default: *kind = ORC_K_DIV_I64; *param = 1000; break;
      }
      return ORC_OK;
    case ORC_T_DURATION:
      *kind = ORC_K_DURATION;
      *param = f->unit == 0 ? 1000000 : f->unit == 1 ? 1000 : f->unit == 2 ? 1 : -1000;
      return ORC_OK;
    case ORC_T_INTERVAL:
      if (f->unit == 0) { *kind = ORC_K_INTERVAL_MONTHS; return ORC_OK; }
      if (f->unit == 2) { *kind = ORC_K_INTERVAL_MDN; return ORC_OK; }
      return ORC_ENOTSUP; /* day_time: upstream's handling is not restated */
    case ORC_T_UTF8: case ORC_T_BINARY: *kind = ORC_K_STR32; *n_buffers = 3; return ORC_OK;
    case ORC_T_LARGE_UTF8: case ORC_T_LARGE_BINARY: *kind = ORC_K_STR64; *n_buffers = 3; return ORC_OK;
    case ORC_T_FIXED_BINARY: *kind = ORC_K_FIXED_BINARY; *param = f->byte_width; return ORC_OK;
    case ORC_T_UTF8_VIEW: case ORC_T_BINARY_VIEW: *kind = ORC_K_STRVIEW; return ORC_OK; /* + variadic buffers */
    case ORC_T_LIST: case ORC_T_MAP: *kind = ORC_K_LIST32; return ORC_OK;
    case ORC_T_LARGE_LIST: *kind = ORC_K_LIST64; return ORC_OK;
    case ORC_T_STRUCT: *kind = ORC_K_STRUCT; *n_buffers = 1; return ORC_OK;
    case ORC_T_FIXED_LIST: *kind = ORC_K_STRUCT; *param = f->byte_width; *n_buffers = 1; return ORC_OK;
    default: return ORC_ENOTSUP;
  }
}


static uint64_t fold(const uint8_t* p, int64_t n) {
  uint64_t h = 0;
  int64_t i = 0;
  for (; i + 8 <= n; i += 8) { uint64_t v; memcpy(&v, p + i, 8); h ^= v + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1); }
  for (; i < n; i++) h ^= (uint64_t)p[i] << (8 * (i & 7));
  return h;
}

/* Scans up to max_batches RecordBatch messages of a flat-schema stream.  Output vectors are 2048-row chunk
 * buffers reused for every window, exactly like the DataChunk the executor hands to the scan function.
 * checksum != 0 additionally folds every produced vector (used by tests, off for timing). */
int orc_scan_stream(const uint8_t* buf, int64_t size, int32_t max_batches, int32_t want_checksum,
                    orc_scan_stats* st) {
  memset(st, 0, sizeof(*st));
  /* scratch tables are thread-local: bench.py's multi-file CPU baseline runs one scan per thread */
  static _Thread_local orc_msg msgs[1 << 16];
  int32_t nmsg = 0;
  char err[128];
  int rc = orc_walk_stream(buf, size, msgs, 1 << 16, &nmsg, err, sizeof(err));
  if (rc) return rc;
  if (nmsg == 0 || msgs[0].type != ORC_MSG_SCHEMA) return ORC_EIO;
  static _Thread_local orc_field fields[MAX_FIELDS];
  int32_t nf = 0, ntop = 0, endian = 0;
  rc = orc_decode_schema(buf + msgs[0].meta_off, msgs[0].meta_len, fields, MAX_FIELDS, &nf, &ntop, &endian);
  if (rc) return rc;
  if (nf != ntop) return ORC_ENOTSUP; /* flat schemas only */
  int32_t kind[MAX_FIELDS], nbuf[MAX_FIELDS];
  int64_t param[MAX_FIELDS];
  for (int32_t c = 0; c < nf; c++) {
    rc = orc_plan_column(&fields[c], &kind[c], &param[c], &nbuf[c]);
    if (rc) return rc;
    if (kind[c] == ORC_K_DICT) return ORC_ENOTSUP; /* the reference rejects dictionary IPC (base_stream_reader.cpp:87-90) */
  }
  /* the reused DataChunk */
  uint8_t* chunk_data[MAX_FIELDS];
  uint64_t chunk_valid[ORC_VECTOR_SIZE / 64];
  for (int32_t c = 0; c < nf; c++) chunk_data[c] = (uint8_t*)malloc((size_t)ORC_VECTOR_SIZE * 16);

  static _Thread_local orc_node nodes[MAX_FIELDS];
  static _Thread_local orc_buf bufs[MAX_BUFS];
  int32_t done = 0;
  for (int32_t m = 1; m < nmsg && done < max_batches; m++) {
    if (msgs[m].type != ORC_MSG_RECORD_BATCH) { rc = ORC_EIO; break; } /* "Expected RecordBatch Arrow IPC message but got ..." */
    int64_t length, dict_id;
    int32_t nn, nb, comp, delta;
    rc = orc_decode_record_batch(buf + msgs[m].meta_off, msgs[m].meta_len, &length, nodes, MAX_FIELDS, &nn, bufs,
                                 MAX_BUFS, &nb, &comp, &dict_id, &delta);
    if (rc) break;
    if (comp != -1) { rc = ORC_ENOTSUP; break; }
    /* 1. body read: allocate + copy (ipc_file_stream_reader.cpp:74-80) */
    uint8_t* body = (uint8_t*)malloc((size_t)(msgs[m].body_len ? msgs[m].body_len : 8));
    memcpy(body, buf + msgs[m].body_off, (size_t)msgs[m].body_len);
    /* 2. slice + FULL validation */
    int32_t bi = 0;
    const uint8_t* cb[MAX_FIELDS][3];
    int64_t cl[MAX_FIELDS][3];
    for (int32_t c = 0; c < nf && rc == ORC_OK; c++) {
      for (int32_t k = 0; k < nbuf[c]; k++, bi++) {
        if (bi >= nb || bufs[bi].offset + bufs[bi].length > msgs[m].body_len) { rc = ORC_EINVAL; break; }
        cb[c][k] = bufs[bi].length ? body + bufs[bi].offset : NULL;
        cl[c][k] = bufs[bi].length;
        st->bytes_in += bufs[bi].length;
      }
      if (rc) break;
      if (kind[c] == ORC_K_STR32 && nodes[c].length > 0)
        rc = orc_validate_offsets32((const int32_t*)cb[c][1], nodes[c].length, cl[c][2]);
      if (kind[c] == ORC_K_STR64 && nodes[c].length > 0)
        rc = orc_validate_offsets64((const int64_t*)cb[c][1], nodes[c].length, cl[c][2]);
    }
    if (rc) { free(body); break; }
    /* 3. pull loop: one DataChunk of <= 2048 rows per call, every column converted */
    for (int64_t o = 0; o < length; o += ORC_VECTOR_SIZE) {
      int64_t n = length - o < ORC_VECTOR_SIZE ? length - o : ORC_VECTOR_SIZE;
      for (int32_t c = 0; c < nf; c++) {
        orc_col_task t;
        memset(&t, 0, sizeof(t));
        t.kind = kind[c];
        t.param = param[c];
        t.validity = nbuf[c] ? cb[c][0] : NULL;
        t.buf1 = nbuf[c] ? cb[c][1] : NULL;
        t.buf2 = nbuf[c] > 2 ? cb[c][2] : NULL;
        t.buf2_len = nbuf[c] > 2 ? cl[c][2] : 0;
        t.null_count = nodes[c].null_count;
        t.ptr_base = (uint64_t)(uintptr_t)t.buf2;
        int32_t w = orc_out_width(t.kind, t.param);
        /* window [o, o+n): emulate by shifting the task so that the driver's single window is this one */
        orc_validity(t.validity, t.null_count, o, n, chunk_valid);
        const uint8_t* produced = chunk_data[c];
        switch (t.kind) {
          case ORC_K_COPY: produced = orc_direct(t.buf1, w, o); break; /* zero-copy alias, no work */
          case ORC_K_BOOL: orc_bool(t.buf1, o, n, chunk_data[c]); break;
          case ORC_K_DEC128: orc_decimal128_narrow(t.buf1, chunk_valid, o, n, w, chunk_data[c]); break;
          case ORC_K_DATE64: orc_date64_to_date32((const int64_t*)t.buf1, o, n, (int32_t*)chunk_data[c]); break;
          case ORC_K_MUL_I32: rc |= orc_mul_i32_to_i64((const int32_t*)t.buf1, chunk_valid, o, n, t.param, (int64_t*)chunk_data[c]); break;
          case ORC_K_MUL_I64: rc |= orc_mul_i64((const int64_t*)t.buf1, chunk_valid, o, n, t.param, (int64_t*)chunk_data[c]); break;
          case ORC_K_DIV_I64: orc_div_i64((const int64_t*)t.buf1, o, n, t.param, (int64_t*)chunk_data[c]); break;
          case ORC_K_DURATION: rc |= orc_duration_to_interval((const int64_t*)t.buf1, chunk_valid, o, n, t.param, chunk_data[c]); break;
          case ORC_K_STR32: rc |= orc_string32((const int32_t*)t.buf1, t.buf2, chunk_valid, o, n, t.ptr_base, chunk_data[c]); break;
          case ORC_K_STR64: rc |= orc_string64((const int64_t*)t.buf1, t.buf2, chunk_valid, o, n, t.ptr_base, chunk_data[c]); break;
          case ORC_K_FIXED_BINARY: orc_fixed_binary(t.buf1, (int32_t)t.param, chunk_valid, o, n, t.ptr_base, chunk_data[c]); break;
          case ORC_K_INTERVAL_MONTHS: orc_interval_months((const int32_t*)t.buf1, o, n, chunk_data[c]); break;
          case ORC_K_INTERVAL_MDN: orc_interval_mdn(t.buf1, o, n, chunk_data[c]); break;
          case ORC_K_NARROW: orc_narrow(t.buf1, (int32_t)(t.param & 0xFF), chunk_valid, o, n, w, chunk_data[c]); break;
          case ORC_K_HALF_FLOAT: orc_half_to_float((const uint16_t*)t.buf1, o, n, (uint32_t*)chunk_data[c]); break;
          case ORC_K_NULL: memset(chunk_valid, 0, sizeof(chunk_valid)); memset(chunk_data[c], 0, (size_t)n); break;
          default: rc = ORC_ENOTSUP; break;
        }
        st->bytes_out += n * w + ((n + 63) / 64) * 8;
        if (want_checksum) {
          /* pointer-free fold: string_t pointers are rebased to offsets so the value is reproducible */
          if (t.kind == ORC_K_STR32 || t.kind == ORC_K_STR64 || t.kind == ORC_K_FIXED_BINARY) {
            for (int64_t i = 0; i < n; i++) {
              uint8_t s[16];
              memcpy(s, chunk_data[c] + 16 * i, 16);
              uint32_t len; memcpy(&len, s, 4);
              if (len > 12) { uint64_t p; memcpy(&p, s + 8, 8); p -= t.ptr_base; memcpy(s + 8, &p, 8); }
              st->checksum ^= fold(s, 16) * (uint64_t)(2 * (o + i) + 1);
            }
          } else {
            st->checksum ^= fold(produced, n * w) * (uint64_t)(2 * o + 1);
          }
          st->checksum ^= fold((const uint8_t*)chunk_valid, ((n + 63) / 64) * 8);
        }
      }
    }
    free(body);
    st->rows += length;
    st->batches++;
    done++;
    if (rc) break;
  }
  for (int32_t c = 0; c < nf; c++) free(chunk_data[c]);
  return rc;
}


/* ------------------------------------------------------------------------------------------------ COPY TO baseline */
static double now_seconds(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int orc_encode_stream(const uint8_t* buf, int64_t size, int32_t max_batches, int32_t verify, orc_encode_stats* st) {
  memset(st, 0, sizeof(*st));
  static _Thread_local orc_msg msgs[1 << 16];
  int32_t nmsg = 0;
  char err[128];
  int rc = orc_walk_stream(buf, size, msgs, 1 << 16, &nmsg, err, sizeof(err));
  if (rc) return rc;
  if (nmsg == 0 || msgs[0].type != ORC_MSG_SCHEMA) return ORC_EIO;
  static _Thread_local orc_field fields[MAX_FIELDS];
  int32_t nf = 0, ntop = 0, endian = 0;
  rc = orc_decode_schema(buf + msgs[0].meta_off, msgs[0].meta_len, fields, MAX_FIELDS, &nf, &ntop, &endian);
  if (rc) return rc;
  if (nf != ntop) return ORC_ENOTSUP;
  int32_t kind[MAX_FIELDS], nbuf[MAX_FIELDS];
  int64_t param[MAX_FIELDS];
  for (int32_t c = 0; c < nf; c++) {
    rc = orc_plan_column(&fields[c], &kind[c], &param[c], &nbuf[c]);
    if (rc) return rc;
    if (kind[c] != ORC_K_COPY && kind[c] != ORC_K_DEC128 && kind[c] != ORC_K_STR32 && kind[c] != ORC_K_BOOL) return ORC_ENOTSUP;
  }
  static _Thread_local orc_node nodes[MAX_FIELDS];
  static _Thread_local orc_buf bufs[MAX_BUFS];
  int32_t done = 0;
  for (int32_t m = 1; m < nmsg && done < max_batches && rc == ORC_OK; m++) {
    if (msgs[m].type != ORC_MSG_RECORD_BATCH) return ORC_EIO;
    int64_t length, dict_id;
    int32_t nn, nb, comp, delta;
    rc = orc_decode_record_batch(buf + msgs[m].meta_off, msgs[m].meta_len, &length, nodes, MAX_FIELDS, &nn, bufs, MAX_BUFS, &nb,
                                 &comp, &dict_id, &delta);
    if (rc) break;
    if (comp != -1) return ORC_ENOTSUP;
    const uint8_t* body = buf + msgs[m].body_off;
    const int64_t n = length, nwords = (n + 63) / 64;
    /* ---- untimed: the DuckDB table the COPY reads, as 2048-row chunks per column (decode of this very batch) ---- */
    uint8_t* vec[MAX_FIELDS];
    uint64_t* val[MAX_FIELDS];
    int32_t width[MAX_FIELDS], first_buf[MAX_FIELDS];
    int32_t bi = 0;
    for (int32_t c = 0; c < nf; c++) {
      first_buf[c] = bi;
      orc_col_task t;
      memset(&t, 0, sizeof(t));
      t.kind = kind[c];
      t.param = param[c];
      t.validity = bufs[bi].length ? body + bufs[bi].offset : NULL;
      t.buf1 = body + bufs[bi + 1].offset;
      t.buf2 = nbuf[c] > 2 ? body + bufs[bi + 2].offset : NULL;
      t.buf2_len = nbuf[c] > 2 ? bufs[bi + 2].length : 0;
      t.null_count = nodes[c].null_count;
      t.nrows = n;
      t.ptr_base = (uint64_t)(uintptr_t)t.buf2;
      width[c] = orc_out_width(kind[c], param[c]);
      vec[c] = (uint8_t*)malloc((size_t)(n ? n : 1) * (size_t)width[c] + 16);
      val[c] = (uint64_t*)malloc((size_t)(nwords + 1) * 8);
      t.out_data = vec[c];
      t.out_validity = val[c];
      rc = orc_decode_column(&t, /*copy_direct*/ 1);
      if (rc) break;
      bi += nbuf[c];
    }
    if (rc) {
      for (int32_t c = 0; c < nf; c++) { free(vec[c]); free(val[c]); }
      break;
    }
    /* ---- timed: Serialize(ColumnDataCollection) ---- */
    const double t0 = now_seconds();
    int64_t body_cap = 0;
    for (int32_t c = 0; c < nf; c++)
      for (int32_t k = 0; k < nbuf[c]; k++) body_cap += ((bufs[first_buf[c] + k].length > (n + 7) / 8 ? bufs[first_buf[c] + k].length : (n + 7) / 8) + 7 + 8) & ~(int64_t)7;
    uint8_t* out_body = (uint8_t*)malloc((size_t)body_cap + 64);
    int64_t out_pos = 0;
    for (int32_t c = 0; c < nf; c++) {
      /* a10: the collection's chunks are concatenated into one DataChunk first */
      uint8_t* flat = (uint8_t*)malloc((size_t)(n ? n : 1) * (size_t)width[c] + 16);
      for (int64_t o = 0; o < n; o += ORC_VECTOR_SIZE) {
        const int64_t k = n - o < ORC_VECTOR_SIZE ? n - o : ORC_VECTOR_SIZE;
        memcpy(flat + o * width[c], vec[c] + o * width[c], (size_t)k * (size_t)width[c]);
      }
      st->bytes_in += n * width[c] + nwords * 8;
      /* a11: ArrowAppender -- validity (always emitted, starts as 0xFF), then the data buffers */
      const int64_t vbytes = (n + 7) / 8;
      uint8_t* bitmap = (uint8_t*)malloc((size_t)vbytes + 8);
      memset(bitmap, 0xFF, (size_t)vbytes + 8);
      int64_t nulls = 0;
      orc_enc_validity(nodes[c].null_count ? val[c] : NULL, n, 0, bitmap, &nulls);
      uint8_t* b1 = NULL;
      uint8_t* b2 = NULL;
      int64_t l1 = 0, l2 = 0;
      switch (kind[c]) {
        case ORC_K_COPY:
          l1 = n * width[c];
          b1 = (uint8_t*)malloc((size_t)l1 + 8);
          memcpy(b1, flat, (size_t)l1);
          break;
        case ORC_K_DEC128:
          l1 = n * 16;
          b1 = (uint8_t*)malloc((size_t)l1 + 8);
          orc_enc_decimal_widen(flat, width[c], n, b1);
          break;
        case ORC_K_BOOL:
          l1 = vbytes;
          b1 = (uint8_t*)malloc((size_t)l1 + 8);
          memset(b1, 0xFF, (size_t)l1);
          orc_enc_bool(flat, nodes[c].null_count ? val[c] : NULL, n, 0, b1);
          break;
        default: { /* ORC_K_STR32 */
          const uint8_t* heap = body + bufs[first_buf[c] + 2].offset;
          l1 = (n + 1) * 4;
          l2 = bufs[first_buf[c] + 2].length;
          b1 = (uint8_t*)malloc((size_t)l1 + 8);
          b2 = (uint8_t*)malloc((size_t)l2 + 8);
          ((int32_t*)b1)[0] = 0;
          rc |= orc_enc_varchar32(flat, nodes[c].null_count ? val[c] : NULL, n, 0, (uint64_t)(uintptr_t)heap, heap, (int32_t*)b1, b2);
          st->bytes_in += l2;
          break;
        }
      }
      /* a12: every buffer is copied into the message body, padded to 8 bytes */
      const uint8_t* parts[3] = {bitmap, b1, b2};
      const int64_t lens[3] = {vbytes, l1, l2};
      for (int32_t k = 0; k < nbuf[c]; k++) {
        memcpy(out_body + out_pos, parts[k], (size_t)lens[k]);
        const int64_t padded = (lens[k] + 7) & ~(int64_t)7;
        memset(out_body + out_pos + lens[k], 0, (size_t)(padded - lens[k]));
        if (verify) {
          /* the source stream was written by the same rules: same bytes (bitmap pad bits are 1 here, the source's may be 0) */
          const orc_buf* sb = &bufs[first_buf[c] + k];
          int64_t cmp = lens[k] < sb->length ? lens[k] : sb->length;
          if (k == 0) cmp = n / 8; /* whole bytes of the bitmap */
          if ((k > 0 && lens[k] != sb->length) || memcmp(out_body + out_pos, body + sb->offset, (size_t)cmp) != 0) st->mismatches++;
        }
        out_pos += padded;
      }
      free(bitmap);
      free(b1);
      free(b2);
      free(flat);
    }
    st->seconds += now_seconds() - t0;
    st->bytes_out += out_pos;
    st->checksum ^= fold(out_body, out_pos) * (uint64_t)(2 * done + 1);
    free(out_body);
    for (int32_t c = 0; c < nf; c++) { free(vec[c]); free(val[c]); }
    st->rows += n;
    st->batches++;
    done++;
  }
  return rc;
}
