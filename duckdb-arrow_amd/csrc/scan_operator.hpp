// scan_operator.hpp -- the scan TableFunction body: read_arrow / scan_arrow_ipc on the MI355X path.
//
// Takes the place of what the reference wires together from DuckDB pieces:
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

#include <atomic>
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <memory>
#include <string>
#include <vector>

#include "batch_planner.hpp"
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

//! One leaf of a pushed-down predicate after normalisation (scan_filter.cpp): every comparison on an integer-like column
//! is an inclusive range (optionally negated), plus IS [NOT] NULL and IN-lists.
struct FilterLeaf {
  std::string column;
  int32_t op = 0;                 // device::kLeaf*
  int64_t lo = 0, hi = 0;         // kLeafRange: lo <= v <= hi ...
  bool lo_open = true, hi_open = true;  // ... where an open end has no bound at all (uint64 columns reach past INT64_MAX)
  bool negate = false;
  std::vector<int64_t> in_values;
  bool is_string = false;         // kLeafStrIn: the column is VARCHAR / BLOB, the row passes when it equals one of ...
  std::vector<std::string> str_values;   // ... these byte strings (negate: none of them)
  //! kLeafStrRange: str_values = {lower, upper}; lo_open / hi_open = no such bound, the two below = the bound itself passes
  bool lo_incl = false, hi_incl = false;
  int32_t out_col = -1;           // resolved at Init: index into the scan's filter columns
};
//! Conjunctive normal form: every clause is an OR of leaves, the filter is the AND of its clauses.
using FilterCnf = std::vector<std::vector<FilterLeaf>>;
FilterCnf NormaliseFilter(const mi_filter_node* nodes, int32_t n_nodes, int32_t root);  // scan_filter.cpp

//! Vectors of one DataChunk (the storage behind mi_data_chunk.columns)
struct ChunkStorage {
  std::vector<mi_vector> vectors;
  std::vector<mi_vector> child_pool;
  size_t child_pool_used = 0;
};

//! A record batch whose decoded vectors are complete and held for the caller (ArrowScan::AcquireBatch)
struct BatchRef {
  int slot = -1;
  int64_t batch_index = 0;
  int64_t nrows = 0;
  int32_t source = 0;
  int32_t n_windows = 0;     // chunks BuildChunk can produce (compacted batches: of the rows that survived the filter)
  int64_t selected = 0;      // rows passing the pushed-down filter (== nrows without one)
  int64_t chunk_rows = 0;    // rows its chunks hold together: nrows, or `selected` when the batch was compacted
};

//! A decoded top-level column of an acquired record batch where it lies in HBM (the fused COPY reads it there)
struct DeviceColumnView {
  bool flat = false;                  // a leaf vector without dictionary / children; the fields below are set only then
  const uint8_t* d_data = nullptr;    // DuckDB vector data (row 0 of the batch)
  const uint8_t* d_validity = nullptr;  // validity words, NULL when the column has no NULLs
  int32_t kind = 0, width = 0;
  int64_t null_count = 0;
  const uint8_t* d_heap = nullptr;    // string kinds: device copy of the Arrow data buffer ...
  uint64_t ptr_base = 0;              // ... and the pointer value its byte 0 has inside the string_t rows
  const uint8_t* h_offsets = nullptr; // string kinds: the Arrow offsets / validity bitmap in the host body
  const uint8_t* h_validity = nullptr;
  int32_t offset_width = 0;
};

class ScanBase {
 public:
  virtual ~ScanBase() = default;
  virtual const std::vector<ScanColumn>& Bind() = 0;
  virtual void Init(const std::vector<std::string>& projected) = 0;
  virtual void SetFilter(FilterCnf cnf) = 0;
  virtual void Next(mi_data_chunk* out) = 0;
  virtual void Count(int64_t* rows, int64_t* selected, int64_t* chunks) = 0;
  virtual void SumProduct(const std::string& a, const std::string& b, const std::vector<std::string>& filter_columns,
                          const std::vector<int64_t>& lo, const std::vector<int64_t>& hi, mi_sum_product_result* out) = 0;
  virtual double Progress() = 0;
  virtual void Stats(mi_scan_stats* out) = 0;   // adds to *out
};

constexpr int kMaxDepth = 16;   // pipeline slots of a scan at most (24 and 32 were measured with 40 hardware queues: slower for both codecs of K8)

class ArrowScan : public ScanBase {
 public:
  ArrowScan(Context* ctx, std::vector<std::string> paths, const mi_scan_options& opts);
  ArrowScan(Context* ctx, std::vector<ArrowIPCBuffer> buffers, const mi_scan_options& opts);
  ~ArrowScan() override;

  //! Bind: schema of the scan (names deduplicated) -- "Provided table/dataframe must have at least one column"
  const std::vector<ScanColumn>& Bind() override;
  //! Init: projection pushdown (column names, output order)
  void Init(const std::vector<std::string>& projected) override;
  //! Predicate pushed into the scan (K6; the reference leaves filters to DuckDB: read_arrow.cpp:47-48)
  void SetFilter(FilterCnf cnf) override;
  //! One DataChunk (<= 2048 rows); size 0 when exhausted
  void Next(mi_data_chunk* out) override;
  void Count(int64_t* rows, int64_t* selected, int64_t* chunks) override;
  double Progress() override;
  void Stats(mi_scan_stats* out) override;
  //! sum(a * b) over rows passing the range filters, all on the GPU; drains the scan (mi_scan_sum_product)
  void SumProduct(const std::string& a, const std::string& b, const std::vector<std::string>& filter_columns,
                  const std::vector<int64_t>& lo, const std::vector<int64_t>& hi, mi_sum_product_result* out) override;

  // ---- batch-level pull (what Next() is built on; the COPY pump and the multi-device scan use it directly) ----
  //! Waits for the next record batch in order; false when the scan is exhausted.  The batch stays valid (its slot is
  //! not recycled) until ReleaseBatch; at most pipeline_depth - 1 batches may be held at once.
  bool AcquireBatch(BatchRef* out);
  //! Chunk `window` (rows [2048 w, 2048 (w+1)) of the batch; fewer when compacted) -> out, vectors in `storage`.
  //! Thread-safe for different batches.
  void BuildChunk(const BatchRef& ref, int32_t window, ChunkStorage* storage, mi_data_chunk* out);
  void ReleaseBatch(const BatchRef& ref);
  //! true once every source is read and no batch is in flight (AcquireBatch returns false both then and when every slot is
  //! held by the caller: release one and ask again)
  bool Exhausted() const { return exhausted && inflight.empty(); }
  void EnsurePipelineDepth(int depth);
  bool HostConsumer() const { return !opts.device_resident; }
  size_t NumOutputColumns() const { return out_columns.size(); }
  const std::vector<ScanColumn>& OutputColumns() const { return out_columns; }
  bool Initialized() const { return initialized; }
  bool HasFilter() const { return has_filter; }
  //! Host consumers only: decoded vectors of the record batches enqueued from now on stay in HBM (no copy into the pinned
  //! output slot) until EnsureHostVectors asks for them -- a consumer that reads them on the GPU (the fused COPY) sets this
  void KeepVectorsOnDevice(bool on) { keep_on_device = on; }
  void EnsureHostVectors(const BatchRef& ref);
  void DeviceColumn(const BatchRef& ref, size_t column, DeviceColumnView* out) const;

 private:
  struct Source {
    std::string path;                          // empty for buffers
    std::unique_ptr<IPCStreamReader> reader;
    std::vector<int32_t> out_to_file_column;   // per output column: index in this file's projected batch, -1 = absent
    std::vector<int32_t> filter_to_file_column;  // per filter-only column (not in the projection)
    std::map<std::string, std::string> hive;   // key -> value parsed from the path
    bool opened = false, prepared = false;
    std::vector<std::string> wanted;           // the reader projection PrepareSource settled on (the extra producers' readers take it too)
  };
  //! One immutable version of a decoded dictionary (dict_len + 1 entries, the last one NULL).  Record batches keep the
  //! version they were enqueued with, so a later replacement / delta never changes what an in-flight batch sees.
  struct DictState {
    void* d_data = nullptr;        // decoded values on the device
    void* d_validity = nullptr;
    void* h_data = nullptr;        // pinned host copy (host consumers)
    void* h_validity = nullptr;    // == h_words for host consumers
    uint64_t* h_words = nullptr;   // pinned: the validity words as built on the host (uploaded from here)
    uint32_t* h_status = nullptr;  // pinned: status word of the decode of the values, checked with the first batch that uses them
    hipEvent_t uploaded = nullptr; // the dictionary body is in HBM
    std::unique_ptr<Plan> decode_plan;   // kept until the version dies: its status word is read asynchronously
    std::vector<std::shared_ptr<void>> d_heaps;      // device copies of the dictionary bodies (long string payload)
    std::vector<std::shared_ptr<void>> host_bodies;  // host bodies: long dictionary strings point into them
    int64_t dict_len = 0;
    int32_t kind = 0, out_width = 0;
    //! string-valued dictionaries: the values themselves (empty + not valid for NULL entries), for pushed-down string
    //! predicates -- the dictionary is matched once, on the host, the rows by index (K6, kLeafDictMap)
    std::vector<std::string> host_strings;
    std::vector<char> host_valid;
    std::map<size_t, std::shared_ptr<void>> match_maps;   // filter leaf -> device byte per entry: 0 no, 1 yes, 2 NULL
    ~DictState();
  };
  struct Slot {
    int64_t tr_enqueued_ns = 0;       // MI_SCAN_TRACE: when the batch was submitted
    // one record batch in flight
    uint8_t* d_in = nullptr;   size_t d_in_cap = 0;    // body in HBM
    uint8_t* d_out = nullptr;  size_t d_out_cap = 0;   // decoded vectors in HBM
    uint8_t* h_out = nullptr;  size_t h_out_cap = 0;   // decoded vectors, pinned
    std::unique_ptr<Plan> plan;                        // full-width decode (all columns, or the filter columns when compacting)
    std::unique_ptr<Plan> gather_plan;                 // compaction: every projected column through the selection vector
    uint32_t* h_status = nullptr;                      // pinned copy of the plans' device status words
    hipEvent_t h2d_done = nullptr, compute_done = nullptr, d2h_done = nullptr, filter_done = nullptr;
    bool busy = false;
    DecodedBatch batch;
    int32_t source = 0;
    int64_t batch_index = 0;
    int64_t nrows = 0;
    BatchPlanner planner{PlannerOptions{}};            // layout + tasks of the projected columns
    std::vector<int32_t> col_root;                     // per output column: planner node (-1: absent in this file)
    std::vector<std::pair<size_t, size_t>> absent;     // per output column absent in this file: {data_off, valid_off}
    std::vector<std::shared_ptr<DictState>> node_dict; // per planner node: the dictionary version this batch uses
    uint8_t* h_aux = nullptr;  size_t h_aux_cap = 0;   // pinned: list window tables, string-view buffer tables, filter program
    uint8_t* d_aux = nullptr;  size_t d_aux_cap = 0;
    size_t sel_off = 0, sel_count_off = 0;             // filter outputs (arena offsets)
    std::vector<int32_t> filter_root;                  // per filter column: planner node of its full-width decoded vector
    uint32_t* h_counts = nullptr; size_t h_counts_cap = 0;  // pinned: rows selected per 2048-row window
    size_t d2h_bytes = 0;                              // bytes that travel back to the host
    size_t stage_a_bytes = 0;                          // arena bytes of the full-width arrays (+ sel, counts)
    bool compact = false;                              // chunks hold only the selected rows (dense arrays behind stage A's)
    bool host_vectors = false;                         // h_out holds the decoded vectors
    // K8: a record batch whose LZ4 buffers are decompressed in HBM (kernels_lz4.hip)
    uint8_t* d_comp = nullptr;  size_t d_comp_cap = 0; // the compressed body
    uint8_t* d_lz4 = nullptr;   size_t d_lz4_cap = 0;  // block / buffer tables, sequence descriptors, links, counters
    uint8_t* h_lz4 = nullptr;   size_t h_lz4_cap = 0;  // pinned copy of the tables
    hipStream_t lz4_stream = nullptr;                  // decompression of this slot overlaps the other slots' copies and kernels
    hipEvent_t lz4_done = nullptr;
    uint8_t* h_mirror = nullptr; size_t h_mirror_cap = 0;   // host consumers: pinned image of the decompressed body; only the
                                                            // string payload buffers are filled (D2H), string_t rows point into it
    bool lz4_counted = true;
    bool lz4_stream_shared = false;
    bool needs_stage_b = false;                        // compaction: the gather + copy back wait for the counts
    uint8_t* compact_region = nullptr;                 // device address of the dense arrays
  };

  void OpenSource(size_t i);
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
  void ProducerLoop(int p);
  void PrepareSource(size_t si);      // per-file column mapping + reader projection
  std::shared_ptr<void> LeaseStaging(size_t bytes, uint8_t** ptr);
  static constexpr int kReadAhead = 3;              // fetched batches waiting for a slot (per producer: 2 when there are several)
  static constexpr int kMaxProducers = 4;
  //! Several read-ahead threads for file scans without dictionaries: producer p reads the record batches j of this scan's
  //! share with j mod P == p (every producer walks every header, bodies that are not its own are stepped over unread -- the
  //! rank / world rule once more, inside the process), so the pread of one body overlaps the header walk, staging lease and
  //! pread of the next ones.  The consumer takes them back in order: batch j from queue j mod P.
  int n_producers = 1;
  std::vector<std::thread> producers;
  std::mutex q_mu;
  std::condition_variable q_cv;
  std::vector<std::deque<Fetched>> fetched;         // one queue per producer
  int64_t next_fetch = 0;                           // j of the batch the consumer takes next
  std::exception_ptr producer_error;                // the first failure of any producer
  std::vector<std::vector<std::unique_ptr<IPCStreamReader>>> extra_readers;   // [producer - 1][source]
  std::vector<Staging> staging;                     // in flight on the GPU + waiting + the one being read
  bool producer_started = false, producer_stop = false;
  void InitSlot(Slot& s);
  void EnqueueBatch(Slot& s);
  void EnqueueStageB(Slot& s);
  void EnqueueLz4(Slot& s);
  void UploadAux(Slot& s, const std::vector<uint64_t>& aux);
  void BuildVector(const Slot& s, int32_t node, size_t window, int64_t compact_rows, uint8_t* base, ChunkStorage* st, mi_vector* out);
  void DecodeDictionary(Source& src, const DecodedBatch& b);
  void EnsureSlotBuffers(Slot& s, size_t in_bytes, size_t out_bytes);
  void EnsureHostOut(Slot& s, size_t bytes);
  Slot* FreeSlot();

  Context* ctx;
  mi_scan_options opts;
  std::vector<Source> sources;
  std::vector<ArrowIPCBuffer> buffers;
  bool is_buffers = false;
  bool bound = false, initialized = false;
  std::vector<ScanColumn> all_columns;   // bind result
  std::vector<ScanColumn> out_columns;   // after projection
  std::vector<ScanColumn> filter_only_columns;  // filter columns outside the projection: decoded, never emitted
  std::vector<std::string> projected_names;

  // pipeline
  std::vector<Slot> slots;
  std::deque<int> inflight;              // slot indices in submission order (not yet acquired)
  size_t cur_source = 0;
  bool exhausted = false;
  // consumer cursor of Next()
  BatchRef cur_ref;
  bool have_cur = false;
  int32_t cur_window = 0;
  ChunkStorage next_storage;
  // fused aggregate (mi_scan_sum_product): output columns the kernel reads, bounds, device accumulator
  struct Aggregate {
    bool on = false;
    int32_t col_a = -1, col_b = -1;
    std::vector<int32_t> filter_cols;
    std::vector<int64_t> lo, hi;
    unsigned long long* d_acc = nullptr;   // {sum lo, sum hi, rows selected}
    int64_t rows_scanned = 0;
  } agg;
  // constant columns (filename / hive): 2048 string_t each per source, host
  std::map<std::pair<int32_t, size_t>, std::vector<mi_string_t>> const_vectors;
  std::mutex const_mu;
  std::vector<mi_validity_t> all_valid;
  // dictionaries by id
  std::map<int64_t, std::shared_ptr<DictState>> dicts;
  // filter
  bool has_filter = false;
  FilterCnf filter;
  //! filter column k -> output column (>= 0) or ~index into filter_only_columns (< 0)
  std::vector<int32_t> filter_columns;
  // Buffers a slot has outgrown.  hipFree / hipHostFree wait for the device to go idle -- with the other slots' record batches
  // in flight that is a pipeline stall of milliseconds -- so they are kept until the scan closes (growth is geometric: at
  // most twice the final sizes in all).
  std::vector<void*> retired_device, retired_host;
  std::mutex retire_mu;   // the pipeline thread and the producers (staging buffers) both retire
  void RetireDevice(void* p) { std::lock_guard<std::mutex> lk(retire_mu); retired_device.push_back(p); }
  void RetireHost(void* p) { std::lock_guard<std::mutex> lk(retire_mu); retired_host.push_back(p); }
  static size_t GrowCap(size_t need, size_t cap) { return std::max(need + need / 4, cap + cap / 2); }   // record batches of a file differ by a few percent
  std::vector<void*> d_in_lists;         // per leaf (clause order): its IN-list in HBM, or NULL
  bool compact = false;
  bool keep_on_device = false;
  mi_scan_stats stats{};
  // MI_SCAN_TRACE: where the host threads' time went (seconds), printed when the scan closes (diagnostics only)
  bool trace = false;
  std::atomic<int64_t> tr_read_ns{0}, tr_push_wait_ns{0}, tr_lease_wait_ns{0};
  int64_t tr_latency_ns = 0, tr_inflight_sum = 0, tr_k8_prep_ns = 0, tr_k8_launch_ns = 0, tr_enqueue_ns = 0, tr_fetch_wait_ns = 0, tr_event_wait_ns = 0, tr_poll_ns = 0;
};

//! read_arrow over several GPUs of one process: one ArrowScan per context, record batch k of the file list
//! goes to sub-scan k mod N, chunks come back in record-batch order (k-way merge on batch_index).  Draining calls
//! (Count, SumProduct) run every sub-scan on its own thread.
class MultiDeviceScan : public ScanBase {
 public:
  MultiDeviceScan(const std::vector<Context*>& ctxs, std::vector<std::string> paths, const mi_scan_options& opts);
  const std::vector<ScanColumn>& Bind() override;
  void Init(const std::vector<std::string>& projected) override;
  void SetFilter(FilterCnf cnf) override;
  void Next(mi_data_chunk* out) override;
  void Count(int64_t* rows, int64_t* selected, int64_t* chunks) override;
  void SumProduct(const std::string& a, const std::string& b, const std::vector<std::string>& filter_columns,
                  const std::vector<int64_t>& lo, const std::vector<int64_t>& hi, mi_sum_product_result* out) override;
  double Progress() override;
  void Stats(mi_scan_stats* out) override;

 private:
  void ForEachParallel(const std::function<void(size_t)>& fn);
  std::vector<std::unique_ptr<ArrowScan>> subs;
  std::vector<mi_data_chunk> pending;   // one chunk per sub-scan, valid until that sub-scan's next Next()
  std::vector<char> have, done;
  int last_emitted = -1;
};

}  // namespace miarrow
