// mi_read_arrow.cpp -- registers read_arrow('file.arrows' | [files]) and the .arrows / .arrow replacement scan
// (src/scanner/read_arrow.cpp:43-86 of the reference); the reader itself is mi_file_scan.hpp.
#include "mi_file_scan.hpp"

namespace duckdb {
namespace ext_nanoarrow {

TableFunction MiReadArrowFunction() {
  MultiFileFunction<MiMultiFileInfo> read_arrow("read_arrow");
  read_arrow.projection_pushdown = true;
  read_arrow.filter_pushdown = false;  // read_arrow.cpp:47-48; see mi_scan_arrow_ipc.cpp for the route predicates can take
  read_arrow.filter_prune = false;
  return static_cast<TableFunction>(read_arrow);
}

//! FROM 'file.arrows' -> read_arrow('file.arrows') (read_arrow.cpp:50-73)
static unique_ptr<TableRef> MiReadArrowReplacement(ClientContext& context, ReplacementScanInput& input,
                                                   optional_ptr<ReplacementScanData> data) {
  auto table_name = ReplacementScan::GetFullPath(input);
  if (!ReplacementScan::CanReplace(table_name, {"arrows", "arrow"})) {
    return nullptr;
  }
  vector<unique_ptr<ParsedExpression>> arguments;
  arguments.push_back(make_uniq<ConstantExpression>(Value(table_name)));
  auto ref = make_uniq<TableFunctionRef>();
  ref->function = make_uniq<FunctionExpression>("read_arrow", std::move(arguments));
  if (!FileSystem::HasGlob(table_name)) {
    ref->alias = FileSystem::GetFileSystem(context).ExtractBaseName(table_name);
  }
  return std::move(ref);
}

void MiRegisterReadArrow(DatabaseInstance& db) {
  auto function = MiReadArrowFunction();
  ExtensionUtil::RegisterFunction(db, function);
  function.arguments = {LogicalType::LIST(LogicalType::VARCHAR)};  // ['file_1.arrow', 'file_2.arrow']
  ExtensionUtil::RegisterFunction(db, function);
  DBConfig::GetConfig(db).replacement_scans.emplace_back(MiReadArrowReplacement);
}

}  // namespace ext_nanoarrow
}  // namespace duckdb
