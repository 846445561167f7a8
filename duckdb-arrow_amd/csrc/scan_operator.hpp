// scan_operator.hpp -- the scan TableFunction body: read_arrow / scan_arrow_ipc on the MI355X path.
//
// Mirrors, for this path, what the reference wires together from DuckDB pieces:
//   bind      ArrowFileScan::ArrowFileScan            src/file_scanner/arrow_file_scan.cpp:9-23
//             ScanArrowIPCFunction::ScanArrowIPCBind   src/scanner/scan_arrow_ipc.cpp:20-48
//             ArrowMultiFileInfo::BindReader           src/file_scanner/arrow_multi_file_info.cpp:54-70
//   init      ArrowFileScan::TryInitializeScan         src/file_scanner/arrow_file_scan.cpp:30-67
//             ArrowIPCStreamFactory::Produce (projection)  src/ipc/stream_factory.cpp:14-30
//   scan      ArrowFileScan::Scan -> ArrowTableFunction::ArrowScanFunction   src/file_scanner/arrow_file_scan.cpp:68-72
//             (<= 2048 rows per call, output cardinality 0 = exhausted)
// The per-value work of ArrowToDuckDB runs in the HIP kernels; this class is the pipeline around them:
// record-batch body -> pinned slot -> hipMemcpyAsync H2D (copy stream) -> class kernels (compute stream) ->
// hipMemcpyAsync D2H (copy-back stream) -> DataChunks that alias the pinned output slot.
#pragma once

#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <memory>
#include <string>
#include <vector>

#include "engine.hpp"
#include "ipc_stream_reader.hpp"

namespace miarrow {

void DeduplicateColumns(std::vector<std::string>& names);  // ipc_stream_reader.cpp

struct ScanColumn {
  std::string name;
  ArrowField field;           // arrow field of the first file that has it
  bool is_filename = false;   // `filename` option (README.md:104-107)
  bool is_hive = false;       // hive partition key
  std::string hive_key;
};

class ArrowScan {
 public:
  ArrowScan(Context* ctx, std::vector<std::string> paths, const mi_scan_options& opts);
  ArrowScan(Context* ctx, std::vector<ArrowIPCBuffer> buffers, const mi_scan_options& opts);
  ~ArrowScan();

  //! Bind: schema of the scan (names deduplicated) -- "Provided table/dataframe must have at least one column"
  const std::vector<ScanColumn>& Bind();
  //! Init: projection pushdown (column names, output order)
  void Init(const std::vector<std::string>& projected);
  //! Range filter pushed into the scan (K6; the reference leaves filters to DuckDB: read_arrow.cpp:47-48)
  void SetFilterRange(const std::string& column, int64_t lo, int64_t hi);
  //! One DataChunk (<= 2048 rows); size 0 when exhausted
  void Next(mi_data_chunk* out);
  double Progress();

 private:
  struct Source {
    std::string path;                          // empty for buffers
    std::unique_ptr<IPCStreamReader> reader;
    std::vector<int32_t> out_to_file_column;   // per output column: index in this file's projected batch, -1 = absent
    std::map<std::string, std::string> hive;   // key -> value parsed from the path
    bool opened = false;
  };
  //! One immutable version of a decoded dictionary (dict_len + 1 entries, the last one NULL).  Record batches keep the
  //! version they were enqueued with, so a later replacement / delta never changes what an in-flight batch sees.
  struct DictState {
    void* d_data = nullptr;        // decoded values on the device
    void* d_validity = nullptr;
    void* h_data = nullptr;        // pinned host copy (host consumers)
    void* h_validity = nullptr;
    std::vector<std::shared_ptr<void>> d_heaps;      // device copies of the dictionary bodies (long string payload)
    std::vector<std::shared_ptr<void>> host_bodies;  // host bodies: long dictionary strings point into them
    int64_t dict_len = 0;
    int32_t kind = 0, out_width = 0;
    ~DictState();
  };
  struct Slot {
    // one record batch in flight
    uint8_t* d_in = nullptr;   size_t d_in_cap = 0;    // body in HBM
    uint8_t* d_out = nullptr;  size_t d_out_cap = 0;   // decoded vectors in HBM
    uint8_t* h_out = nullptr;  size_t h_out_cap = 0;   // decoded vectors, pinned
    std::unique_ptr<Plan> plan;
    uint32_t* h_status = nullptr;                      // pinned copy of the plan's device status word
    hipEvent_t h2d_done = nullptr, compute_done = nullptr, d2h_done = nullptr;
    bool busy = false;
    DecodedBatch batch;
    int32_t source = 0;
    int64_t batch_index = 0;
    int64_t nrows = 0;
    std::vector<size_t> col_data_off, col_valid_off;   // per output column, offsets into d_out / h_out
    //! decoded field nodes of the batch (nested columns have children); col_root[c] = node of output column c (-1: absent)
    struct NodeOut {
      size_t data_off = 0, valid_off = 0;
      int32_t kind = 0, width = 0, arrow_type = 0;
      int64_t param = 0, nrows = 0;
      std::vector<int64_t> win;       // first row (in this node's row space) of every top-level 2048-row window, + end
      std::vector<int32_t> children;
      std::shared_ptr<DictState> dict;
      const uint8_t* alias = nullptr;  // zero_copy_direct: the values live in the record-batch body, validity = all valid
    };
    std::vector<std::pair<int64_t, int64_t>> upload;   // body byte ranges the kernels read (everything unless aliasing)
    std::vector<NodeOut> node_out;
    std::vector<int32_t> col_root;
    uint8_t* h_aux = nullptr;  size_t h_aux_cap = 0;   // pinned: list window tables, string-view buffer tables
    uint8_t* d_aux = nullptr;  size_t d_aux_cap = 0;
    size_t sel_off = 0, sel_count_off = 0;             // filter outputs
    std::shared_ptr<void> external_body;               // buffer sources: nothing to own, body is caller memory
    std::vector<std::shared_ptr<DictState>> col_dict;  // per output column: the dictionary version this batch uses
  };

  void OpenSource(size_t i);
  void BuildOutputSchema();
  //! takes the next fetched record batch and enqueues its GPU work; false when nothing could be submitted (no free
  //! slot, nothing fetched yet while `may_block` is false, or every source is exhausted)
  bool SubmitNextBatch(bool may_block);
  // ---- read-ahead: a producer thread walks the sources (open, column mapping, projection, sharding, pread into pinned
  // staging buffers) a few record batches ahead of the consumer, so file I/O overlaps whatever the consumer does between
  // two Next() calls; the reference reads synchronously inside the scan call (ipc_file_stream_reader.cpp:71-94)
  struct Fetched {
    DecodedBatch batch;
    int32_t source = 0;
    int64_t ordinal = 0;
    bool end = false;                 // every source is exhausted
    std::exception_ptr error;         // raised where the consumer reaches it, after the batches read before it
  };
  struct Staging {                    // pinned body buffers, leased to one record batch at a time
    uint8_t* p = nullptr;
    size_t cap = 0;
    bool leased = false;
  };
  void StartProducer();
  void StopProducer();
  void ProducerLoop();
  void PrepareSource(size_t si);      // per-file column mapping + reader projection (was inline in SubmitNextBatch)
  std::shared_ptr<void> LeaseStaging(size_t bytes, uint8_t** ptr);
  static constexpr int kReadAhead = 3;              // fetched batches waiting for a slot
  static constexpr int kStaging = 3 + kReadAhead + 1;  // in flight on the GPU + waiting + the one being read
  std::thread producer;
  std::mutex q_mu;
  std::condition_variable q_cv;
  std::deque<Fetched> fetched;
  Staging staging[kStaging];
  bool producer_started = false, producer_stop = false;
  void EnqueueBatch(Slot& s);
 public:
  //! sum(a * b) over rows passing the range filters, all on the GPU; drains the scan (mi_scan_sum_product)
  void SumProduct(const std::string& a, const std::string& b, const std::vector<std::string>& filter_columns,
                  const std::vector<int64_t>& lo, const std::vector<int64_t>& hi, mi_sum_product_result* out);
 private:
  int32_t AddNode(Slot& s, const DecodedBatch& b, int32_t ni, std::vector<int64_t> win, bool win_is_tiles, int64_t parent_valid_off,
                  int32_t parent_div, size_t* off, std::vector<mi_col_task>* tasks, std::vector<uint64_t>* aux,
                  std::vector<std::pair<size_t, size_t>>* aux_fixups);
  void BuildVector(Slot& s, int32_t node, size_t window, uint8_t* base, mi_vector* out);
  void DecodeDictionary(Source& src, const DecodedBatch& b);
  void EnsureSlotBuffers(Slot& s, size_t in_bytes, size_t out_bytes);
  Slot* FreeSlot();

  Context* ctx;
  mi_scan_options opts;
  std::vector<Source> sources;
  std::vector<ArrowIPCBuffer> buffers;
  bool is_buffers = false;
  bool bound = false, initialized = false;
  std::vector<ScanColumn> all_columns;   // bind result
  std::vector<ScanColumn> out_columns;   // after projection
  std::vector<std::string> projected_names;

  // pipeline
  static constexpr int kSlots = 3;
  Slot slots[kSlots];
  std::vector<int> inflight;             // slot indices in submission order
  size_t cur_source = 0;
  int64_t next_batch_ordinal = 0;        // global record-batch ordinal (sharding + order)
  bool exhausted = false;
  // consumer cursor
  int cur_slot = -1;
  int64_t cur_row = 0;
  std::vector<mi_vector> chunk_vectors;
  std::vector<mi_vector> child_pool;     // children of nested vectors of the current chunk
  size_t child_pool_used = 0;
  int32_t filter_node = -1;              // field node of the filter column in the batch being enqueued
  // fused aggregate (mi_scan_sum_product): output columns the kernel reads, bounds, device accumulator
  struct Aggregate {
    bool on = false;
    int32_t col_a = -1, col_b = -1;
    std::vector<int32_t> filter_cols;
    std::vector<int64_t> lo, hi;
    unsigned long long* d_acc = nullptr;   // {sum lo, sum hi, rows selected}
    int64_t rows_scanned = 0;
  } agg;
  // constant columns (filename / hive): 2048 string_t each, host
  std::vector<std::vector<mi_string_t>> const_vectors;
  std::vector<mi_validity_t> all_valid;
  // dictionaries by id
  std::map<int64_t, std::shared_ptr<DictState>> dicts;
  // filter
  bool has_filter = false;
  std::string filter_column;
  int filter_out_col = -1;
  int64_t filter_lo = 0, filter_hi = 0;
  // progress
  int64_t total_bytes = 0, consumed_bytes = 0;
};

}  // namespace miarrow
