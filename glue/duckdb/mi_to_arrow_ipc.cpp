// mi_to_arrow_ipc.cpp -- to_arrow_ipc(TABLE) -> rows of (ipc BLOB, header BOOLEAN) on the MI355X path.
//
// Replaces src/writer/to_arrow_ipc.cpp:72-182 of the reference: same in-out table function (first row = the Schema message
// with header = true, emitted by exactly one thread; then one RecordBatch message per 120 x 2048 rows, or per input chunk
// when operator caching is off; the tail in the final call), same output layout (header || body in one BLOB).  The
// reference appends every input chunk to an ArrowAppender on the CPU and encodes with nanoarrow; here the thread's chunks
// are staged and the K7 kernels encode the whole record batch (mi_ipc_serialize_chunks).
#include "mi_glue_common.hpp"

#include "duckdb/common/types/column/column_data_collection.hpp"
#include "duckdb/execution/physical_operator.hpp"
#include "duckdb/main/extension_util.hpp"

namespace duckdb {
namespace ext_nanoarrow {

namespace {

constexpr idx_t MI_TO_IPC_VECTORS_PER_MESSAGE = 120;  // ToArrowIPCFunction::DEFAULT_CHUNK_SIZE (to_arrow_ipc.hpp:28)

struct MiToIPCBindData : public TableFunctionData {
  vector<LogicalType> types;
  vector<string> names;
  const idx_t chunk_size = MI_TO_IPC_VECTORS_PER_MESSAGE * STANDARD_VECTOR_SIZE;
};

struct MiToIPCGlobalState : public GlobalTableFunctionState {
  atomic<bool> sent_schema {false};
  mutex lock;
};

struct MiToIPCLocalState : public LocalTableFunctionState {
  unique_ptr<MiCtx> ctx;
  mi_writer* serializer = nullptr;
  //! the rows waiting for their message: input chunks are only valid during the call that brings them
  unique_ptr<ColumnDataCollection> buffer;
  ColumnDataAppendState append_state;
  idx_t current_count = 0;
  bool checked_schema = false;
  ~MiToIPCLocalState() override {
    if (serializer) {
      mi_writer_close(serializer);
    }
  }
};

unique_ptr<FunctionData> MiToIPCBind(ClientContext& context, TableFunctionBindInput& input, vector<LogicalType>& return_types,
                                     vector<string>& names) {
  auto result = make_uniq<MiToIPCBindData>();
  return_types.emplace_back(LogicalType::BLOB);
  names.emplace_back("ipc");
  return_types.emplace_back(LogicalType::BOOLEAN);
  names.emplace_back("header");
  result->types = input.input_table_types;
  result->names = input.input_table_names;
  return std::move(result);
}

unique_ptr<GlobalTableFunctionState> MiToIPCInitGlobal(ClientContext& context, TableFunctionInitInput& input) {
  return make_uniq<MiToIPCGlobalState>();
}

unique_ptr<LocalTableFunctionState> MiToIPCInitLocal(ExecutionContext& context, TableFunctionInitInput& input,
                                                     GlobalTableFunctionState* global_state) {
  auto& bind = input.bind_data->Cast<MiToIPCBindData>();
  auto local = make_uniq<MiToIPCLocalState>();
  local->ctx = make_uniq<MiCtx>(0);
  auto fields = DuckToMiFields(bind.names, bind.types);
  MiCheck(mi_ipc_serializer_create(local->ctx->h, fields.data(), NumericCast<int32_t>(fields.size()), &local->serializer));
  return std::move(local);
}

//! the message goes out as the one row of `output`: column 0 = the bytes, column 1 = "this is the schema"
void MiEmitMessage(const uint8_t* blob, int64_t size, bool is_header, DataChunk& output) {
  auto& vec = output.data[0];
  FlatVector::GetData<string_t>(vec)[0] = StringVector::AddStringOrBlob(vec, const_char_ptr_cast(blob), NumericCast<idx_t>(size));
  output.data[1].SetValue(0, Value::BOOLEAN(is_header));
  output.SetCardinality(1);
}

//! every buffered chunk -> one record batch on the GPU -> header || body
void MiSerializeBuffered(MiToIPCLocalState& local, DataChunk& output) {
  vector<unique_ptr<DataChunk>> chunks;
  vector<unique_ptr<MiChunkView>> views;
  vector<mi_data_chunk> c_chunks;
  for (auto& chunk : local.buffer->Chunks()) {
    chunks.push_back(make_uniq<DataChunk>());
    chunks.back()->Initialize(Allocator::DefaultAllocator(), chunk.GetTypes());
    chunks.back()->Reference(chunk);
    views.push_back(make_uniq<MiChunkView>());
    DuckChunkToMi(*chunks.back(), *views.back());
    c_chunks.push_back(views.back()->chunk);
  }
  const uint8_t* blob = nullptr;
  int64_t size = 0;
  MiCheck(mi_ipc_serialize_chunks(local.serializer, c_chunks.data(), NumericCast<int32_t>(c_chunks.size()), &blob, &size));
  MiEmitMessage(blob, size, false, output);
  local.buffer.reset();
  local.current_count = 0;
}

OperatorResultType MiToIPCFunction(ExecutionContext& context, TableFunctionInput& data_p, DataChunk& input, DataChunk& output) {
  auto& bind = data_p.bind_data->Cast<MiToIPCBindData>();
  auto& local = data_p.local_state->Cast<MiToIPCLocalState>();
  auto& global = data_p.global_state->Cast<MiToIPCGlobalState>();
  bool sending_schema = false;
  if (!local.checked_schema) {
    if (!global.sent_schema) {
      lock_guard<mutex> guard(global.lock);
      if (!global.sent_schema) {  // this call sends the schema; the other threads send record batches only
        global.sent_schema = true;
        sending_schema = true;
      }
    }
    local.checked_schema = true;
  }
  if (sending_schema) {
    const uint8_t* blob = nullptr;
    int64_t size = 0;
    MiCheck(mi_ipc_serialize_schema(local.serializer, &blob, &size));
    MiEmitMessage(blob, size, true, output);
    return OperatorResultType::HAVE_MORE_OUTPUT;  // the same input chunk comes back for its rows
  }
  if (!local.buffer) {
    local.buffer = make_uniq<ColumnDataCollection>(context.client, bind.types);
    local.buffer->InitializeAppend(local.append_state);
  }
  local.buffer->Append(local.append_state, input);
  local.current_count += input.size();
  const bool caching_disabled = !PhysicalOperator::OperatorCachingAllowed(context);
  if (caching_disabled || local.current_count >= bind.chunk_size) {
    MiSerializeBuffered(local, output);
  }
  return OperatorResultType::NEED_MORE_INPUT;
}

OperatorFinalizeResultType MiToIPCFunctionFinal(ExecutionContext& context, TableFunctionInput& data_p, DataChunk& output) {
  auto& local = data_p.local_state->Cast<MiToIPCLocalState>();
  if (local.buffer && local.current_count > 0) {
    MiSerializeBuffered(local, output);
  }
  return OperatorFinalizeResultType::FINISHED;
}

}  // namespace

void MiRegisterToArrowIPC(DatabaseInstance& db) {
  TableFunction fun("to_arrow_ipc", {LogicalType::TABLE}, nullptr, MiToIPCBind, MiToIPCInitGlobal, MiToIPCInitLocal);
  fun.in_out_function = MiToIPCFunction;
  fun.in_out_function_final = MiToIPCFunctionFinal;
  ExtensionUtil::RegisterFunction(db, fun);
}

}  // namespace ext_nanoarrow
}  // namespace duckdb
