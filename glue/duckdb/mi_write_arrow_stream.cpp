// mi_write_arrow_stream.cpp -- COPY ... TO 'out.arrows' (FORMAT ARROWS | ARROW) on the MI355X path.
//
// Replaces src/writer/write_arrow_stream.cpp:54-272 of the reference: same CopyFunction registration (names "arrows" and
// "arrow", every callback), same options and BinderException texts (parsed by mi_write_options_*), same execution modes.
// What the callbacks do changes: the reference buffers chunks in a ColumnDataCollection, concatenates them into one
// DataChunk and runs ArrowAppender + nanoarrow's encoder on the CPU (column_data_collection_serializer.cpp:80-115); here
// a sink thread's rows are staged in pinned memory and every row group is encoded by the K7 kernels into the IPC body that
// is written (mi_writer_local_*), one serializer + HIP stream per sink thread.
#include "mi_file_scan.hpp"

#include "duckdb/common/multi_file/multi_file_function.hpp"
#include "duckdb/common/types/column/column_data_collection.hpp"
#include "duckdb/function/copy_function.hpp"
#include "duckdb/main/config.hpp"
#include "duckdb/main/extension_util.hpp"

namespace duckdb {
namespace ext_nanoarrow {

TableFunction MiReadArrowFunction();  // mi_read_arrow.cpp

namespace {

struct MiWriteBindData : public TableFunctionData {
  vector<LogicalType> sql_types;
  vector<string> column_names;
  mi_write_options options;
};

struct MiWriteGlobalState : public GlobalFunctionData {
  unique_ptr<MiCtx> ctx;
  mi_writer* writer = nullptr;
  ~MiWriteGlobalState() override {
    if (writer) {
      mi_writer_close(writer);
    }
  }
};

struct MiWriteLocalState : public LocalFunctionData {
  mi_writer_local* local = nullptr;
  MiChunkView view;
  ~MiWriteLocalState() override {
    if (local) {
      mi_writer_local_destroy(local);
    }
  }
};

//! ArrowWriteBind (write_arrow_stream.cpp:54-125): the option loop, delegated; the library words the BinderExceptions
unique_ptr<FunctionData> MiWriteBind(ClientContext& context, CopyFunctionBindInput& input, const vector<string>& names,
                                     const vector<LogicalType>& sql_types) {
  D_ASSERT(names.size() == sql_types.size());
  auto bind = make_uniq<MiWriteBindData>();
  auto check_bind = [](int rc) {
    if (rc == MI_EINVAL) {
      throw BinderException(mi_last_error());
    }
    MiCheck(rc);
  };
  MiCheck(mi_write_options_init(&bind->options));
  bind->options.preserve_insertion_order = DBConfig::GetConfig(context).options.preserve_insertion_order ? 1 : 0;
  bind->options.arrow_large_buffer_size = context.GetClientProperties().arrow_offset_size == ArrowOffsetSize::LARGE ? 1 : 0;
  for (auto& option : input.info.options) {
    const auto loption = StringUtil::Lower(option.first);
    if (loption == "kv_metadata" && option.second.size() == 1) {
      auto& kv_struct = option.second[0];
      if (kv_struct.type().id() != LogicalTypeId::STRUCT) {
        throw BinderException("Expected kv_metadata argument to be a STRUCT");
      }
      auto& values = StructValue::GetChildren(kv_struct);
      for (idx_t i = 0; i < values.size(); i++) {
        // BLOB values are written raw, everything else as its string form (write_arrow_stream.cpp:95-101)
        const string value = values[i].type().id() == LogicalTypeId::BLOB ? StringValue::Get(values[i]) : values[i].ToString();
        check_bind(mi_write_options_add_kv(&bind->options, StructType::GetChildName(kv_struct.type(), i).c_str(), value.data(),
                                           NumericCast<int32_t>(value.size())));
      }
      continue;
    }
    // exactly one argument per option: a NULL value makes the library raise "<NAME> requires exactly one argument"
    string value;
    if (option.second.size() == 1) {
      auto v = option.second[0];
      if (loption == "row_group_size_bytes" && v.type().id() == LogicalTypeId::VARCHAR) {
        value = std::to_string(DBConfig::ParseMemoryLimit(v.ToString()));  // '2MB'
      } else {
        value = v.ToString();
      }
    }
    check_bind(mi_write_options_set(&bind->options, option.first.c_str(), option.second.size() == 1 ? value.c_str() : nullptr));
  }
  check_bind(mi_write_options_finalize(&bind->options));
  bind->sql_types = sql_types;
  bind->column_names = names;
  return std::move(bind);
}

//! ArrowWriteInitializeGlobal (:127-139): creates the file and writes the Schema message
unique_ptr<GlobalFunctionData> MiWriteInitializeGlobal(ClientContext& context, FunctionData& bind_data, const string& file_path) {
  auto& bind = bind_data.Cast<MiWriteBindData>();
  auto g = make_uniq<MiWriteGlobalState>();
  g->ctx = make_uniq<MiCtx>(0);
  auto fields = DuckToMiFields(bind.column_names, bind.sql_types);
  MiCheck(mi_writer_open(g->ctx->h, file_path.c_str(), fields.data(), NumericCast<int32_t>(fields.size()), &bind.options, &g->writer));
  return std::move(g);
}

//! ArrowWriteInitializeLocal (:176-180): every sink thread gets its own staging buffer, serializer and HIP stream.  The
//! writer is only known at sink time (the local state is created before the global one in some plans): created lazily.
unique_ptr<LocalFunctionData> MiWriteInitializeLocal(ExecutionContext& context, FunctionData& bind_data) {
  return make_uniq<MiWriteLocalState>();
}

//! ArrowWriteSink (:141-159): append; the library flushes a row group through the K7 kernels when row_group_size /
//! row_group_size_bytes is reached
void MiWriteSink(ExecutionContext& context, FunctionData& bind_data, GlobalFunctionData& gstate, LocalFunctionData& lstate,
                 DataChunk& input) {
  auto& global = gstate.Cast<MiWriteGlobalState>();
  auto& local = lstate.Cast<MiWriteLocalState>();
  if (!local.local) {
    MiCheck(mi_writer_local_create(global.writer, &local.local));
  }
  DuckChunkToMi(input, local.view);
  MiCheck(mi_writer_local_sink(local.local, &local.view.chunk));
}

//! ArrowWriteCombine (:161-167): the rows left in the thread's buffer become its last row group
void MiWriteCombine(ExecutionContext& context, FunctionData& bind_data, GlobalFunctionData& gstate, LocalFunctionData& lstate) {
  auto& local = lstate.Cast<MiWriteLocalState>();
  if (local.local) {
    MiCheck(mi_writer_local_combine(local.local));
  }
}

//! ArrowWriteFinalize (:169-174): end-of-stream marker, close
void MiWriteFinalize(ClientContext& context, FunctionData& bind_data, GlobalFunctionData& gstate) {
  auto& global = gstate.Cast<MiWriteGlobalState>();
  MiCheck(mi_writer_finalize(global.writer));
}

CopyFunctionExecutionMode MiWriteExecutionMode(bool preserve_insertion_order, bool supports_batch_index) {
  if (!preserve_insertion_order) {
    return CopyFunctionExecutionMode::PARALLEL_COPY_TO_FILE;
  }
  if (supports_batch_index) {
    return CopyFunctionExecutionMode::BATCH_COPY_TO_FILE;
  }
  return CopyFunctionExecutionMode::REGULAR_COPY_TO_FILE;
}

idx_t MiWriteDesiredBatchSize(ClientContext& context, FunctionData& bind_data) {
  return NumericCast<idx_t>(bind_data.Cast<MiWriteBindData>().options.row_group_size);
}

bool MiWriteRotateFiles(FunctionData& bind_data, const optional_idx& file_size_bytes) {
  return file_size_bytes.IsValid() || bind_data.Cast<MiWriteBindData>().options.row_groups_per_file > 0;
}

bool MiWriteRotateNextFile(GlobalFunctionData& gstate, FunctionData& bind_data, const optional_idx& file_size_bytes) {
  auto& global = gstate.Cast<MiWriteGlobalState>();
  return mi_writer_rotate_next_file(global.writer, file_size_bytes.IsValid() ? NumericCast<int64_t>(file_size_bytes.GetIndex()) : -1) != 0;
}

//! BATCH_COPY_TO_FILE: prepare_batch runs concurrently and may not touch the writer (:225-238) -- it gets a serializer of its
//! own (own pinned staging + HIP stream) and returns the finished header||body message; flush_batch appends it (:240-245).
struct MiWriteBatchData : public PreparedBatchData {
  mi_writer* serializer = nullptr;
  const uint8_t* blob = nullptr;
  int64_t size = 0;
  ~MiWriteBatchData() override {
    if (serializer) {
      mi_writer_close(serializer);
    }
  }
};

unique_ptr<PreparedBatchData> MiWritePrepareBatch(ClientContext& context, FunctionData& bind_data, GlobalFunctionData& gstate,
                                                  unique_ptr<ColumnDataCollection> collection) {
  auto& bind = bind_data.Cast<MiWriteBindData>();
  auto& global = gstate.Cast<MiWriteGlobalState>();
  auto batch = make_uniq<MiWriteBatchData>();
  auto fields = DuckToMiFields(bind.column_names, bind.sql_types);
  MiCheck(mi_ipc_serializer_create(global.ctx->h, fields.data(), NumericCast<int32_t>(fields.size()), &batch->serializer));
  // the chunks of the collection, flattened, as one array of mi_data_chunk: one record batch comes back
  vector<unique_ptr<DataChunk>> chunks;
  vector<unique_ptr<MiChunkView>> views;
  vector<mi_data_chunk> c_chunks;
  for (auto& chunk : collection->Chunks()) {
    chunks.push_back(make_uniq<DataChunk>());
    chunks.back()->Initialize(Allocator::DefaultAllocator(), chunk.GetTypes());
    chunks.back()->Reference(chunk);
    views.push_back(make_uniq<MiChunkView>());
    DuckChunkToMi(*chunks.back(), *views.back());
    c_chunks.push_back(views.back()->chunk);
  }
  MiCheck(mi_ipc_serialize_chunks(batch->serializer, c_chunks.data(), NumericCast<int32_t>(c_chunks.size()), &batch->blob, &batch->size));
  collection->Reset();
  return std::move(batch);
}

void MiWriteFlushBatch(ClientContext& context, FunctionData& bind_data, GlobalFunctionData& gstate, PreparedBatchData& batch_p) {
  auto& global = gstate.Cast<MiWriteGlobalState>();
  auto& batch = batch_p.Cast<MiWriteBatchData>();
  MiCheck(mi_writer_append_message(global.writer, batch.blob, batch.size));
}

}  // namespace

void MiRegisterArrowStreamCopyFunction(DatabaseInstance& db) {
  CopyFunction function("arrows");
  function.copy_to_bind = MiWriteBind;
  function.copy_to_initialize_global = MiWriteInitializeGlobal;
  function.copy_to_initialize_local = MiWriteInitializeLocal;
  function.copy_to_sink = MiWriteSink;
  function.copy_to_combine = MiWriteCombine;
  function.copy_to_finalize = MiWriteFinalize;
  function.execution_mode = MiWriteExecutionMode;
  function.copy_from_bind = MultiFileFunction<MiMultiFileInfo>::MultiFileBindCopy;
  function.copy_from_function = MiReadArrowFunction();
  function.prepare_batch = MiWritePrepareBatch;
  function.flush_batch = MiWriteFlushBatch;
  function.desired_batch_size = MiWriteDesiredBatchSize;
  function.rotate_files = MiWriteRotateFiles;
  function.rotate_next_file = MiWriteRotateNextFile;
  function.extension = "arrows";
  ExtensionUtil::RegisterFunction(db, function);
  function.name = "arrow";
  function.extension = "arrow";
  ExtensionUtil::RegisterFunction(db, function);
}

}  // namespace ext_nanoarrow
}  // namespace duckdb
