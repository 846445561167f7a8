/*
 * mi_synth.h -- seeded synthetic TPC-H `lineitem` in Arrow IPC stream format.
 *
 * BENCH / TEST SUPPORT, not part of the drop-in boundary (that is mi_arrow_ipc.h).  It stands in for
 * `CALL dbgen(sf=...)` + `COPY lineitem TO 'lineitem.arrows' (FORMAT arrows)` of the reference's benchmark
 * (benchmark/lineitem.py:120-126,149-153), which cannot run here (no DuckDB).  The stream has the schema DuckDB
 * exports for lineitem (4 x int64 keys, 4 x decimal128(15,2), 2 x utf8 flags, 3 x date32, 3 x utf8), record batches
 * of `rows_per_batch` rows (default 122880 = the writer's row group, write_arrow_stream.cpp:33), buffers 8-byte
 * aligned/padded like nanoarrow's encoder, validity bitmaps present (DuckDB-writer style) or omitted (pyarrow style).
 * Every value is a pure function of (seed, global row, column): any batch can be generated independently on any rank.
 */
#ifndef MI_SYNTH_H
#define MI_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_synth_options {
  double scale_factor;     /* 1 -> 6001215 rows, 10 -> 59986052, 100 -> 600037902 */
  uint64_t seed;
  int64_t rows_per_batch;  /* 0 -> 122880 */
  int64_t n_rows;          /* 0 -> derived from scale_factor */
  int64_t first_row;       /* global row number of this stream's first row (a multiple of rows_per_batch): lets every
                              rank generate its own row-group shard of one big table */
  int32_t with_validity;   /* 1: every column carries an all-ones bitmap (DuckDB's writer always emits them) */
  int32_t n_threads;       /* 0 -> hardware concurrency (capped at 32) */
} mi_synth_options;

/* Rows / batches / exact stream size (schema message + batches + EOS).  batch_offsets (optional, n_batches + 1
 * entries) receives the stream position of every RecordBatch message and of the EOS marker. */
int mi_synth_lineitem_layout(const mi_synth_options* o, int64_t* n_rows, int64_t* n_batches, int64_t* stream_size,
                             int64_t* batch_offsets, int64_t batch_offsets_cap);
/* Writes the whole stream into out[0, stream_size).  Multi-threaded over record batches. */
int mi_synth_lineitem_fill(const mi_synth_options* o, uint8_t* out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
