/* TPC-H Q6 over lineitem.arrows through the C ABI alone -- what the host glue of a DuckDB extension (or any C program)
 * links against: no C++, no torch, no HIP headers.
 *
 *   gcc -std=c99 -Iinclude examples/q6.c -Lduckdb-arrow_amd -lmi_arrow_ipc -Wl,-rpath,$PWD/duckdb-arrow_amd -o q6
 *   ./q6 lineitem.arrows [more files...]
 *
 * SELECT sum(l_extendedprice * l_discount) FROM read_arrow(files)
 *  WHERE l_shipdate >= DATE '1994-01-01' AND l_shipdate < DATE '1995-01-01'
 *    AND l_discount BETWEEN 0.05 AND 0.07 AND l_quantity < 24          (benchmark/lineitem.py:22-34 of the reference)
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi_arrow_ipc.h"

static void check(int rc, const char* what) {
  if (rc != MI_OK) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, mi_last_error());
    exit(1);
  }
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s file.arrows [file.arrows ...]\n", argv[0]);
    return 2;
  }
  mi_ctx* ctx = NULL;
  check(mi_ctx_create(0, &ctx), "mi_ctx_create");

  mi_scan_options opts;
  memset(&opts, 0, sizeof(opts)); /* defaults reproduce the reference */
  mi_scan* scan = NULL;
  check(mi_scan_open_files(ctx, (const char* const*)(argv + 1), argc - 1, &opts, &scan), "mi_scan_open_files");

  int32_t n_fields = 0;
  check(mi_scan_bind(scan, NULL, 0, &n_fields), "mi_scan_bind");
  mi_field* fields = (mi_field*)calloc((size_t)n_fields, sizeof(mi_field));
  check(mi_scan_bind(scan, fields, n_fields, &n_fields), "mi_scan_bind");
  printf("%d columns:", n_fields);
  for (int32_t i = 0; i < n_fields; i++) printf(" %s %s%s", fields[i].name, fields[i].duck_type, i + 1 < n_fields ? "," : "\n");

  /* stored integers: DATE = days since 1970-01-01, DECIMAL(15,2) 0.05 = 5 */
  const mi_range_filter filters[3] = {{"l_shipdate", 8766, 9131}, {"l_discount", 5, 8}, {"l_quantity", INT64_MIN, 2400}};
  mi_sum_product_result r;
  check(mi_scan_sum_product(scan, "l_extendedprice", "l_discount", filters, 3, &r), "mi_scan_sum_product");
  if (r.sum_hi != 0 && r.sum_hi != -1) {
    printf("revenue does not fit 64 bits: hi=%" PRId64 " lo=%" PRIu64 "\n", r.sum_hi, r.sum_lo);
  } else {
    const int64_t scaled = (int64_t)r.sum_lo; /* DECIMAL(15,2) * DECIMAL(15,2): scale 4 */
    printf("revenue = %" PRId64 ".%04" PRId64 "  (%" PRId64 " of %" PRId64 " rows pass)\n", scaled / 10000, scaled % 10000,
           r.rows_selected, r.rows_scanned);
  }
  free(fields);
  mi_scan_close(scan);
  mi_ctx_destroy(ctx);
  return 0;
}
