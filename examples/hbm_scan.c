/* The headline measurement through the C ABI alone: a TPC-H lineitem.arrows stream resident in HBM, every column of every
 * record batch transcoded to DuckDB vectors, rows/s and achieved HBM GB/s per kernel -- what bench.py reports, without
 * Python, torch or HIP headers on the client side.  The layout / task planner behind mi_hbm_open is the scan operator's.
 *
 *   gcc -std=c99 -Iinclude examples/hbm_scan.c -Lduckdb-arrow_amd -lmi_arrow_ipc -Wl,-rpath,$PWD/duckdb-arrow_amd -o hbm_scan
 *   ./hbm_scan [scale_factor=10] [steps=20] [file.arrows]      (without a file: the seeded synthetic table of mi_synth.h)
 */
#define _POSIX_C_SOURCE 199309L
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mi_arrow_ipc.h"
#include "mi_synth.h"

static void check(int rc, const char* what) {
  if (rc != MI_OK) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, mi_last_error());
    exit(1);
  }
}

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char** argv) {
  const double sf = argc > 1 ? atof(argv[1]) : 10.0;
  const int steps = argc > 2 ? atoi(argv[2]) : 20;
  uint8_t* stream = NULL;
  int64_t size = 0;
  if (argc > 3) { /* an uncompressed IPC stream from a file */
    FILE* f = fopen(argv[3], "rb");
    if (!f) { perror(argv[3]); return 1; }
    fseek(f, 0, SEEK_END);
    size = ftell(f);
    fseek(f, 0, SEEK_SET);
    stream = (uint8_t*)malloc((size_t)size);
    if (fread(stream, 1, (size_t)size, f) != (size_t)size) { perror("read"); return 1; }
    fclose(f);
  } else {
    mi_synth_options so;
    memset(&so, 0, sizeof(so));
    so.scale_factor = sf;
    so.seed = 42;
    so.with_validity = 1;
    int64_t rows = 0, batches = 0;
    check(mi_synth_lineitem_layout(&so, &rows, &batches, &size, NULL, 0), "mi_synth_lineitem_layout");
    stream = (uint8_t*)malloc((size_t)size);
    check(mi_synth_lineitem_fill(&so, stream, size), "mi_synth_lineitem_fill");
  }

  mi_ctx* ctx = NULL;
  check(mi_ctx_create(0, &ctx), "mi_ctx_create");
  mi_hbm_options o;
  memset(&o, 0, sizeof(o)); /* every vector materialised, device pointers, library-owned memory */
  mi_hbm* h = NULL;
  check(mi_hbm_open(ctx, stream, size, &o, &h), "mi_hbm_open");
  mi_hbm_layout lay;
  check(mi_hbm_layout_get(h, &lay), "mi_hbm_layout_get");
  int64_t bytes_read = 0, bytes_written = 0, rows = 0, tiles = 0;
  check(mi_hbm_stats(h, &bytes_read, &bytes_written, &rows, &tiles), "mi_hbm_stats");
  printf("%" PRId64 " rows in %d messages, %d tasks, %" PRId64 " tiles; %.2f B/row read + %.2f B/row written\n", lay.n_rows, lay.n_batches,
         lay.n_tasks, tiles, (double)bytes_read / (double)lay.n_rows, (double)bytes_written / (double)lay.n_rows);

  uint32_t status = 0;
  for (int i = 0; i < 3; i++) check(mi_hbm_launch(h, NULL), "mi_hbm_launch"); /* warmup */
  check(mi_hbm_status(h, &status), "mi_hbm_status");                          /* waits for the stream */
  if (status) { check(mi_status_to_error(status), "device status"); }
  const double t0 = now();
  for (int i = 0; i < steps; i++) check(mi_hbm_launch(h, NULL), "mi_hbm_launch");
  check(mi_hbm_status(h, &status), "mi_hbm_status");
  const double ms = (now() - t0) / steps * 1e3;
  if (status) { check(mi_status_to_error(status), "device status"); }
  printf("%.3f ms per step = %.2f G rows/s, %.0f GB/s of algorithmic traffic (%.1f %% of 8 TB/s)\n", ms, (double)lay.n_rows / ms / 1e6,
         (double)(bytes_read + bytes_written) / ms / 1e6, (double)(bytes_read + bytes_written) / ms / 1e6 / 80.0);

  float per_class[MI_NUM_KERNEL_CLASSES];
  check(mi_hbm_launch_timed(h, NULL, per_class), "mi_hbm_launch_timed");
  for (int c = 0; c < MI_NUM_KERNEL_CLASSES; c++) {
    int64_t r = 0, w = 0, n = 0, t = 0;
    const char* name = "";
    check(mi_hbm_class_stats(h, c, &r, &w, &n, &t, &name), "mi_hbm_class_stats");
    if (t) printf("  %-22s %7.3f ms  %6.0f GB/s\n", name, per_class[c], (double)(r + w) / per_class[c] / 1e6);
  }
  /* one value back, to show where the vectors are: l_orderkey of the first row of the first record batch */
  for (int32_t i = 0; i < lay.n_nodes; i++) {
    const mi_hbm_node* nd = &lay.nodes[i];
    if (nd->parent < 0 && !lay.batches[nd->batch].is_dictionary && nd->out_width == 8 && nd->data_off >= 0) {
      int64_t v = 0;
      check(mi_hbm_fetch(h, 0, nd->data_off, 8, &v), "mi_hbm_fetch");
      printf("  %s[0] = %" PRId64 "\n", nd->name, v);
      break;
    }
  }
  mi_hbm_close(h);
  mi_ctx_destroy(ctx);
  free(stream);
  return 0;
}
