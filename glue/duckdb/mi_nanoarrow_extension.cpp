// mi_nanoarrow_extension.cpp -- the extension entry with the MI355X bodies behind the reference's names.
//
// Replaces src/nanoarrow_extension.cpp:33-66: nanoarrow_version(), read_arrow (+ LIST(VARCHAR) overload and the .arrows /
// .arrow replacement scan), COPY ... (FORMAT ARROWS | ARROW), scan_arrow_ipc and to_arrow_ipc are registered under the same
// names; a database that loads this build instead of the reference's sees the same SQL surface.
#define DUCKDB_EXTENSION_MAIN
#include "mi_glue_common.hpp"

#include "duckdb.hpp"
#include "duckdb/function/scalar_function.hpp"
#include "duckdb/main/extension.hpp"
#include "duckdb/main/extension_util.hpp"

namespace duckdb {

namespace ext_nanoarrow {
void MiRegisterReadArrow(DatabaseInstance& db);                // mi_read_arrow.cpp
void MiRegisterScanArrowIPC(DatabaseInstance& db);             // mi_scan_arrow_ipc.cpp
void MiRegisterArrowStreamCopyFunction(DatabaseInstance& db);  // mi_write_arrow_stream.cpp
void MiRegisterToArrowIPC(DatabaseInstance& db);               // mi_to_arrow_ipc.cpp
}  // namespace ext_nanoarrow

namespace {

//! SELECT nanoarrow_version() -- the format level this build reads and writes (test/sql/nanoarrow.test:15-18)
void MiNanoarrowVersion(DataChunk& args, ExpressionState& state, Vector& result) {
  result.SetValue(0, StringVector::AddString(result, mi_nanoarrow_version()));
  result.SetVectorType(VectorType::CONSTANT_VECTOR);
}

void MiLoadInternal(DatabaseInstance& db) {
  if (mi_device_count() <= 0) {
    // the scan and COPY paths have no CPU fallback: say so when the extension is loaded, not at the first query
    throw IOException("nanoarrow (MI355X build): no HIP device is visible to this process");
  }
  ExtensionUtil::RegisterFunction(db, ScalarFunction("nanoarrow_version", {}, LogicalType::VARCHAR, MiNanoarrowVersion));
  ext_nanoarrow::MiRegisterReadArrow(db);
  ext_nanoarrow::MiRegisterArrowStreamCopyFunction(db);
  ext_nanoarrow::MiRegisterScanArrowIPC(db);
  ext_nanoarrow::MiRegisterToArrowIPC(db);
}

}  // namespace

class NanoarrowExtension : public Extension {
 public:
  void Load(DuckDB& db) override { MiLoadInternal(*db.instance); }
  std::string Name() override { return "nanoarrow"; }
  std::string Version() const override {
#ifdef EXT_VERSION_NANOARROW
    return EXT_VERSION_NANOARROW;
#else
    return mi_version();
#endif
  }
};

}  // namespace duckdb

extern "C" {

DUCKDB_EXTENSION_API void nanoarrow_init(duckdb::DatabaseInstance& db) {
  duckdb::DuckDB db_wrapper(db);
  db_wrapper.LoadExtension<duckdb::NanoarrowExtension>();
}

DUCKDB_EXTENSION_API const char* nanoarrow_version() { return duckdb::DuckDB::LibraryVersion(); }
}

#ifndef DUCKDB_EXTENSION_MAIN
#error DUCKDB_EXTENSION_MAIN not defined
#endif
