// mi_scan_arrow_ipc.cpp -- scan_arrow_ipc(LIST(STRUCT(ptr POINTER, size UBIGINT))) on the MI355X path.
//
// Replaces src/scanner/scan_arrow_ipc.cpp:20-64 of the reference: same name, argument type, bind behaviour (schema of the
// buffers, deduplicated names, "Provided table/dataframe must have at least one column") and flags; the function body --
// ArrowTableFunction::ArrowScanFunction over a BufferIPCStreamFactory there -- is mi_scan_* here: the buffers' record
// batches go to HBM, are transcoded by the HIP kernels and come back as DuckDB vectors (<= 2048 rows per call).
// The caller keeps the IPC buffers alive for the scan, as src/include/table_function/scan_arrow_ipc.hpp:18-23 demands:
// string_t rows longer than 12 bytes point into them (the reference's zero-copy SetVectorString does the same).
#include "mi_glue_common.hpp"

#include "duckdb/main/config.hpp"
#include "duckdb/main/extension_util.hpp"
#include "duckdb/main/query_result.hpp"

namespace duckdb {
namespace ext_nanoarrow {

struct MiScanIPCBindData : public TableFunctionData {
  vector<mi_ipc_buffer> buffers;  // == vector<ArrowIPCBuffer>
  vector<mi_field> fields;
  vector<string> names;           // deduplicated, as returned to the binder
  vector<LogicalType> types;
  MiPushedFilter pushed;          // filters taken from DuckDB at optimisation time (pushdown_complex_filter)
};

struct MiScanIPCGlobalState : public GlobalTableFunctionState {
  unique_ptr<MiCtx> ctx;
  MiScan scan;
  mutex lock;
  vector<LogicalType> scanned_types;  // of the projected columns, in output order
  vector<MiDictionaryCache> dictionaries;
  //! one puller: the library pipelines record batches itself (pread / H2D / kernels / D2H on its own streams); use
  //! mi_scan_options.rank / world to give several DuckDB threads their own scans over disjoint record batches
  idx_t MaxThreads() const override { return 1; }
};

static unique_ptr<FunctionData> MiScanArrowIPCBind(ClientContext& context, TableFunctionBindInput& input,
                                                   vector<LogicalType>& return_types, vector<string>& names) {
  auto res = make_uniq<MiScanIPCBindData>();
  for (auto& buffer_struct : ListValue::GetChildren(input.inputs[0])) {
    auto& unpacked = StructValue::GetChildren(buffer_struct);
    mi_ipc_buffer b;
    b.ptr = unpacked[0].GetPointer();
    b.size = unpacked[1].GetValue<uint64_t>();
    res->buffers.push_back(b);
  }
  // the schema message is parsed on the host: no GPU is touched at bind time
  mi_reader* reader = nullptr;
  MiCheck(mi_reader_open_buffers(res->buffers.data(), NumericCast<int32_t>(res->buffers.size()), &reader));
  try {
    MiSchemaToDuck(reader, res->fields, names, return_types);
  } catch (...) {
    mi_reader_close(reader);
    throw;
  }
  mi_reader_close(reader);
  QueryResult::DeduplicateColumns(names);
  if (return_types.empty()) {
    throw InvalidInputException("Provided table/dataframe must have at least one column");
  }
  res->names = names;
  res->types = return_types;
  return std::move(res);
}

static unique_ptr<GlobalTableFunctionState> MiScanArrowIPCInitGlobal(ClientContext& context, TableFunctionInitInput& input) {
  auto& bind = input.bind_data->Cast<MiScanIPCBindData>();
  auto g = make_uniq<MiScanIPCGlobalState>();
  g->ctx = make_uniq<MiCtx>(0);
  mi_scan_options opts;
  memset(&opts, 0, sizeof(opts));  // the defaults reproduce the reference (plain columns alias the caller's buffers)
  MiCheck(mi_scan_open_buffers(g->ctx->h, bind.buffers.data(), NumericCast<int32_t>(bind.buffers.size()), &opts, &g->scan.h));
  // the library binds the same schema again (and deduplicates the names the same way): projection is by name
  int32_t n_fields = 0;
  MiCheck(mi_scan_bind(g->scan.h, nullptr, 0, &n_fields));
  MiApplyFilter(g->scan.h, bind.pushed);
  vector<const char*> projected;
  for (auto& col : input.column_ids) {
    if (col == COLUMN_IDENTIFIER_ROW_ID) {
      throw NotImplementedException("scan_arrow_ipc has no row ids");
    }
    projected.push_back(bind.names[col].c_str());
    g->scanned_types.push_back(bind.types[col]);
  }
  MiCheck(mi_scan_init(g->scan.h, projected.data(), NumericCast<int32_t>(projected.size())));
  return std::move(g);
}

//! == ArrowTableFunction::ArrowScanFunction (scan_arrow_ipc.cpp:56): one DataChunk per call, cardinality 0 = exhausted
static void MiScanArrowIPCFunction(ClientContext& context, TableFunctionInput& data, DataChunk& output) {
  auto& g = data.global_state->Cast<MiScanIPCGlobalState>();
  lock_guard<mutex> guard(g.lock);
  // vectors alias the scan's pinned result slot (and the caller's IPC buffers): valid until the next call, the lifetime
  // the reference's ArrowArray-backed vectors have (the array is released when the next one is fetched)
  MiScanIntoChunk(g.scan.h, g.scanned_types, output, g.dictionaries);
}

//! Takes the filters the kernels can evaluate (K6) out of DuckDB's hands; the rest stays above the scan.
static void MiScanArrowIPCPushdown(ClientContext& context, LogicalGet& get, FunctionData* bind_data_p,
                                   vector<unique_ptr<Expression>>& filters) {
  auto& bind = bind_data_p->Cast<MiScanIPCBindData>();
  MiFilterTranslator translator(get, bind.names, bind.pushed);
  translator.Take(filters);
}

static double MiScanArrowIPCProgress(ClientContext& context, const FunctionData* bind_data,
                                     const GlobalTableFunctionState* global_state) {
  auto& g = global_state->Cast<MiScanIPCGlobalState>();
  return mi_scan_progress(g.scan.h);
}

TableFunction MiScanArrowIPCFunctionDefinition() {
  child_list_t<LogicalType> buffer_struct {{"ptr", LogicalType::POINTER}, {"size", LogicalType::UBIGINT}};
  TableFunction fun("scan_arrow_ipc", {LogicalType::LIST(LogicalType::STRUCT(buffer_struct))}, MiScanArrowIPCFunction,
                    MiScanArrowIPCBind, MiScanArrowIPCInitGlobal);
  fun.projection_pushdown = true;
  // the reference leaves both off (scan_arrow_ipc.cpp:60-61).  Table filters stay off here too; the predicates the K6 kernel
  // evaluates are taken through pushdown_complex_filter, expression by expression, so DuckDB keeps whatever is not taken
  fun.filter_pushdown = false;
  fun.filter_prune = false;
  fun.pushdown_complex_filter = MiScanArrowIPCPushdown;
  fun.table_scan_progress = MiScanArrowIPCProgress;
  return fun;
}

void MiRegisterScanArrowIPC(DatabaseInstance& db) {
  ExtensionUtil::RegisterFunction(db, MiScanArrowIPCFunctionDefinition());
}

}  // namespace ext_nanoarrow
}  // namespace duckdb
