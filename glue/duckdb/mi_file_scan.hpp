// mi_file_scan.hpp -- the per-file reader and the MultiFileReaderInterface behind read_arrow on the MI355X path (shared by
// mi_read_arrow.cpp, which registers the function, and mi_write_arrow_stream.cpp, whose COPY FROM binds through it).
//
// Replaces src/scanner/read_arrow.cpp:43-86, src/file_scanner/arrow_file_scan.cpp:9-72 and
// src/file_scanner/arrow_multi_file_info.cpp of the reference.  DuckDB's MultiFileFunction keeps doing what it does there
// (file lists and globs, union_by_name, filename, hive_partitioning, cross-file casts); the per-file reader -- the
// reference's ArrowFileScan, which wraps ArrowTableFunction::ArrowScanFunction over a FileIPCStreamFactory -- is MiFileScan:
// one mi_scan per file whose record batches are read (projected preads into pinned memory), DMA'd to HBM, transcoded by the
// HIP kernels and handed back as DuckDB vectors.
#pragma once

#include "mi_glue_common.hpp"

#include "duckdb/common/multi_file/base_file_reader.hpp"
#include "duckdb/common/multi_file/multi_file_function.hpp"
#include "duckdb/main/config.hpp"
#include "duckdb/main/extension_util.hpp"
#include "duckdb/main/query_result.hpp"
#include "duckdb/parser/expression/constant_expression.hpp"
#include "duckdb/parser/expression/function_expression.hpp"
#include "duckdb/parser/tableref/table_function_ref.hpp"

namespace duckdb {
namespace ext_nanoarrow {

//! scanner options: none yet, like the reference (ArrowFileReaderOptions)
class MiFileReaderOptions : public BaseFileReaderOptions {};

struct MiFileLocalState : public LocalTableFunctionState {
  explicit MiFileLocalState(ExecutionContext& execution_context_p) : execution_context(execution_context_p) {}
  ExecutionContext& execution_context;
  //! the scan of the file this thread currently reads (one reader thread per file: arrow_file_scan.cpp:35-42)
  unique_ptr<MiCtx> ctx;
  unique_ptr<MiScan> scan;
  vector<LogicalType> scanned_types;
  vector<MiDictionaryCache> dictionaries;
};

struct MiFileGlobalState : public GlobalTableFunctionState {
  MiFileGlobalState(ClientContext& context_p, const MultiFileGlobalState& global_state_p)
      : global_state(global_state_p), context(context_p) {}
  const MultiFileGlobalState& global_state;
  ClientContext& context;
  mutex lock;
  set<idx_t> files;  // file_list_idx of the files a thread has taken
};

//! One .arrows / .arrow file (stream or file format; ZSTD / LZ4_FRAME bodies; dictionaries are refused like the reference)
class MiFileScan : public BaseFileReader {
 public:
  MiFileScan(ClientContext& context, const string& file_name) : BaseFileReader(file_name) {
    mi_reader* reader = nullptr;
    MiCheck(mi_reader_open_file(file_name.c_str(), &reader));
    try {
      MiSchemaToDuck(reader, fields, names, types);
    } catch (...) {
      mi_reader_close(reader);
      throw;
    }
    mi_reader_close(reader);
    QueryResult::DeduplicateColumns(names);
    if (types.empty()) {
      throw InvalidInputException("Provided table/dataframe must have at least one column");
    }
    columns = MultiFileColumnDefinition::ColumnsFromNamesAndTypes(names, types);
  }

  string GetReaderType() const override { return "ARROW"; }
  const vector<string>& GetNames() const { return names; }
  const vector<LogicalType>& GetTypes() const { return types; }

  bool TryInitializeScan(ClientContext& context, GlobalTableFunctionState& gstate_p, LocalTableFunctionState& lstate_p) override {
    auto& gstate = gstate_p.Cast<MiFileGlobalState>();
    auto& lstate = lstate_p.Cast<MiFileLocalState>();
    {
      lock_guard<mutex> guard(gstate.lock);
      if (gstate.files.find(file_list_idx.GetIndex()) != gstate.files.end()) {
        return false;  // another thread reads this file (the record batches of ONE file are pipelined inside the library)
      }
      gstate.files.insert(file_list_idx.GetIndex());
    }
    if (!lstate.ctx) {
      lstate.ctx = make_uniq<MiCtx>(0);
    }
    lstate.scan = make_uniq<MiScan>();
    lstate.dictionaries.clear();
    mi_scan_options opts;
    memset(&opts, 0, sizeof(opts));
    const char* path = GetFileName().c_str();
    MiCheck(mi_scan_open_files(lstate.ctx->h, &path, 1, &opts, &lstate.scan->h));
    int32_t n_fields = 0;
    MiCheck(mi_scan_bind(lstate.scan->h, nullptr, 0, &n_fields));
    // projection: the columns MultiFileReader asks of THIS file (its own column order), by name
    const auto& ids = column_indexes.empty() ? gstate.global_state.column_indexes : column_indexes;
    vector<const char*> projected;
    lstate.scanned_types.clear();
    for (auto& index : ids) {
      const auto col = index.GetPrimaryIndex();
      if (col >= names.size()) {
        throw InternalException("read_arrow: column index %llu outside the schema of \"%s\"", col, GetFileName());
      }
      projected.push_back(names[col].c_str());
      lstate.scanned_types.push_back(types[col]);
    }
    MiCheck(mi_scan_init(lstate.scan->h, projected.data(), NumericCast<int32_t>(projected.size())));
    return true;
  }

  void Scan(ClientContext& context, GlobalTableFunctionState& global_state, LocalTableFunctionState& local_state,
            DataChunk& chunk) override {
    auto& lstate = local_state.Cast<MiFileLocalState>();
    MiScanIntoChunk(lstate.scan->h, lstate.scanned_types, chunk, lstate.dictionaries);
  }

  double Progress(LocalTableFunctionState& local_state) const {
    auto& lstate = local_state.Cast<MiFileLocalState>();
    return lstate.scan && lstate.scan->h ? mi_scan_progress(lstate.scan->h) : 100.0;
  }

  shared_ptr<BaseUnionData> GetUnionData(idx_t file_idx) override {
    auto data = make_shared_ptr<BaseUnionData>(GetFileName());
    data->names = names;
    data->types = types;
    return data;
  }

 private:
  vector<mi_field> fields;
  vector<string> names;
  vector<LogicalType> types;
};

struct MiMultiFileData final : public TableFunctionData {};

//! Same hooks as the reference's ArrowMultiFileInfo (src/include/file_scanner/arrow_multi_file_info.hpp:55-135): the ones that
//! are no-ops there are no-ops here.
struct MiMultiFileInfo : MultiFileReaderInterface {
  static unique_ptr<MultiFileReaderInterface> InitializeInterface(ClientContext& context, MultiFileReader& reader,
                                                                   MultiFileList& file_list) {
    return make_uniq<MiMultiFileInfo>();
  }
  unique_ptr<BaseFileReaderOptions> InitializeOptions(ClientContext& context, optional_ptr<TableFunctionInfo> info) override {
    return make_uniq<MiFileReaderOptions>();
  }
  bool ParseCopyOption(ClientContext& context, const string& key, const vector<Value>& values, BaseFileReaderOptions& options,
                       vector<string>& expected_names, vector<LogicalType>& expected_types) override {
    return false;  // the scanner has no options of its own
  }
  bool ParseOption(ClientContext& context, const string& key, const Value& val, MultiFileOptions& file_options,
                   BaseFileReaderOptions& options) override {
    return false;
  }
  void FinalizeCopyBind(ClientContext& context, BaseFileReaderOptions& options, const vector<string>& expected_names,
                        const vector<LogicalType>& expected_types) override {}
  unique_ptr<TableFunctionData> InitializeBindData(MultiFileBindData& multi_file_data,
                                                   unique_ptr<BaseFileReaderOptions> options) override {
    return make_uniq<MiMultiFileData>();
  }
  //! schema of the first file, or of all of them with union_by_name (arrow_multi_file_info.cpp:54-70)
  void BindReader(ClientContext& context, vector<LogicalType>& return_types, vector<string>& names,
                  MultiFileBindData& bind_data) override {
    MiFileReaderOptions options;
    if (bind_data.file_options.union_by_name) {
      bind_data.reader_bind = bind_data.multi_file_reader->BindUnionReader(context, return_types, names, *bind_data.file_list,
                                                                           bind_data, options, bind_data.file_options);
    } else {
      bind_data.reader_bind = bind_data.multi_file_reader->BindReader(context, return_types, names, *bind_data.file_list,
                                                                      bind_data, options, bind_data.file_options);
    }
    D_ASSERT(names.size() == return_types.size());
  }
  void FinalizeBindData(MultiFileBindData& multi_file_data) override {}
  void GetBindInfo(const TableFunctionData& bind_data, BindInfo& info) override {}
  //! one thread per file (arrow_multi_file_info.cpp:77-86): inside a file the library overlaps read, copy and kernels itself
  optional_idx MaxThreads(const MultiFileBindData& bind_data, const MultiFileGlobalState& global_state,
                          FileExpandResult expand_result) override {
    if (expand_result == FileExpandResult::MULTIPLE_FILES) {
      return optional_idx();
    }
    return 1;
  }
  unique_ptr<GlobalTableFunctionState> InitializeGlobalState(ClientContext& context, MultiFileBindData& bind_data,
                                                             MultiFileGlobalState& global_state) override {
    return make_uniq<MiFileGlobalState>(context, global_state);
  }
  unique_ptr<LocalTableFunctionState> InitializeLocalState(ExecutionContext& context,
                                                           GlobalTableFunctionState& function_state) override {
    return make_uniq<MiFileLocalState>(context);
  }
  shared_ptr<BaseFileReader> CreateReader(ClientContext& context, GlobalTableFunctionState& gstate, BaseUnionData& union_data,
                                          const MultiFileBindData& bind_data) override {
    return make_shared_ptr<MiFileScan>(context, union_data.GetFileName());
  }
  shared_ptr<BaseFileReader> CreateReader(ClientContext& context, GlobalTableFunctionState& gstate, const OpenFileInfo& file,
                                          idx_t file_idx, const MultiFileBindData& bind_data) override {
    return make_shared_ptr<MiFileScan>(context, file.path);
  }
  shared_ptr<BaseFileReader> CreateReader(ClientContext& context, const OpenFileInfo& file, BaseFileReaderOptions& options,
                                          const MultiFileOptions& file_options) override {
    return make_shared_ptr<MiFileScan>(context, file.path);
  }
  static void FinalizeReader(ClientContext& context, BaseFileReader& reader, GlobalTableFunctionState&) {}
  static void FinishFile(ClientContext& context, GlobalTableFunctionState& global_state, BaseFileReader& reader) {}
  void FinishReading(ClientContext& context, GlobalTableFunctionState& global_state,
                     LocalTableFunctionState& local_state) override {
    auto& lstate = local_state.Cast<MiFileLocalState>();
    lstate.scan.reset();  // gives the pinned slots and the HBM of the last file back
  }
  unique_ptr<NodeStatistics> GetCardinality(const MultiFileBindData& bind_data, idx_t file_count) override {
    return make_uniq<NodeStatistics>();
  }
  static unique_ptr<BaseStatistics> GetStatistics(ClientContext& context, BaseFileReader& reader, const string& name) {
    return nullptr;
  }
  static double GetProgressInFile(ClientContext& context, const BaseFileReader& reader) {
    return 0;  // the scan handle lives in the thread's local state (MiFileScan::Progress); a file is either open or done
  }
  void GetVirtualColumns(ClientContext& context, MultiFileBindData& bind_data, virtual_column_map_t& result) override {
    if (result.find(COLUMN_IDENTIFIER_EMPTY) != result.end()) {
      result.erase(COLUMN_IDENTIFIER_EMPTY);
    }
  }
};

}  // namespace ext_nanoarrow
}  // namespace duckdb
