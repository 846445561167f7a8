// mi_glue_common.hpp -- what the four DuckDB-facing sources of the nanoarrow extension share once their bodies call
// libmi_arrow_ipc.so (include/mi_arrow_ipc.h) instead of nanoarrow + DuckDB's CPU ArrowToDuckDB / ArrowAppender.
//
// Written against the DuckDB >= 1.3 extension API the reference uses (duckdb/common/multi_file/*, TableFunction,
// CopyFunction).  Compiled only when a DuckDB source tree is supplied (CMakeLists.txt: -DDUCKDB_DIR=...): this repository
// holds no DuckDB headers and no stand-ins for them.
//
// Replaces: THROW_NOT_OK (src/include/nanoarrow_errors.hpp:10-23), the vector wiring of ArrowToDuckDB that the reference
// reaches through ArrowTableFunction::ArrowScanFunction (src/scanner/scan_arrow_ipc.cpp:56,
// src/file_scanner/arrow_file_scan.cpp:68-72), and the DataChunk -> Arrow hand-over of
// ColumnDataCollectionSerializer::Serialize (src/writer/column_data_collection_serializer.cpp:80-115).
#pragma once

#include "duckdb/common/exception.hpp"
#include "duckdb/common/types/data_chunk.hpp"
#include "duckdb/common/types/selection_vector.hpp"
#include "duckdb/common/types/vector.hpp"
#include "duckdb/common/types/vector_buffer.hpp"
#include "duckdb/function/table_function.hpp"
#include "duckdb/parser/parsed_data/create_table_function_info.hpp"
#include "duckdb/planner/expression/bound_columnref_expression.hpp"
#include "duckdb/planner/expression/bound_comparison_expression.hpp"
#include "duckdb/planner/expression/bound_conjunction_expression.hpp"
#include "duckdb/planner/expression/bound_constant_expression.hpp"
#include "duckdb/planner/expression/bound_operator_expression.hpp"
#include "duckdb/planner/operator/logical_get.hpp"

extern "C" {
#include "mi_arrow_ipc.h"
}

namespace duckdb {
namespace ext_nanoarrow {

// ------------------------------------------------------------------------------------------------ errors
//! status + mi_last_error() -> the exception the reference raises for the same condition
inline void MiCheck(int rc) {
  if (rc == MI_OK) {
    return;
  }
  const string msg = mi_last_error();
  switch (rc) {
    case MI_EIO:
      throw IOException(msg);
    case MI_ENOTSUP:
      throw NotImplementedException(msg);
    case MI_ERANGE:
      throw ConversionException(msg);
    case MI_ENOMEM:
      throw OutOfMemoryException(msg);
    case MI_ENODEV:
      throw IOException(msg);  // no GPU: this path has no CPU fallback
    default:
      throw InvalidInputException(msg);  // MI_EINVAL carries the Internal / InvalidInput / Binder texts
  }
}

// ------------------------------------------------------------------------------------------------ handles
struct MiCtx {
  mi_ctx* h = nullptr;
  explicit MiCtx(int device = 0) { MiCheck(mi_ctx_create(device, &h)); }
  ~MiCtx() {
    if (h) {
      mi_ctx_destroy(h);
    }
  }
  MiCtx(const MiCtx&) = delete;
  MiCtx& operator=(const MiCtx&) = delete;
};

struct MiScan {
  mi_scan* h = nullptr;
  ~MiScan() {
    if (h) {
      mi_scan_close(h);
    }
  }
};

// ------------------------------------------------------------------------------------------------ schema
//! mi_reader_schema -> names + LogicalTypes (the job of ArrowTableFunction::PopulateArrowTableType in the reference)
inline void MiSchemaToDuck(mi_reader* reader, vector<mi_field>& fields, vector<string>& names, vector<LogicalType>& types) {
  int32_t n = 0;
  MiCheck(mi_reader_schema(reader, nullptr, 0, &n));
  fields.resize(NumericCast<idx_t>(n));
  MiCheck(mi_reader_schema(reader, fields.data(), n, &n));
  for (auto& f : fields) {
    if (f.kind == 0) {
      throw NotImplementedException("Column '%s' (%s) is not decoded by the MI355X scan path", f.name, f.format);
    }
    names.emplace_back(f.name);
    types.push_back(TransformStringToLogicalType(f.duck_type));
  }
}

//! names + LogicalTypes -> the fields mi_writer_open / mi_ipc_serializer_create take (name + duck_type is all they read)
inline vector<mi_field> DuckToMiFields(const vector<string>& names, const vector<LogicalType>& types) {
  vector<mi_field> fields(names.size());
  for (idx_t i = 0; i < names.size(); i++) {
    memset(&fields[i], 0, sizeof(mi_field));
    const string type_name = types[i].ToString();
    if (names[i].size() >= sizeof(fields[i].name) || type_name.size() >= sizeof(fields[i].duck_type)) {
      throw NotImplementedException("Column name or type of '%s' is too long for the Arrow IPC writer", names[i]);
    }
    memcpy(fields[i].name, names[i].data(), names[i].size());
    memcpy(fields[i].duck_type, type_name.data(), type_name.size());
  }
  return fields;
}

// ------------------------------------------------------------------------------------------------ scan -> DataChunk
//! Keeps whatever a zero-copy vector points into alive for as long as DuckDB holds the vector: the reference attaches the
//! ArrowArray to its vectors the same way (ArrowAuxiliaryData); the foreign-memory idiom is the VectorBuffer subclass of
//! src/include/writer/to_arrow_ipc.hpp:16-23.
class MiLeaseBuffer : public VectorBuffer {
 public:
  explicit MiLeaseBuffer(shared_ptr<void> lease_p) : VectorBuffer(VectorBufferType::OPAQUE_BUFFER), lease(std::move(lease_p)) {}

 private:
  shared_ptr<void> lease;
};

//! One decoded dictionary as a DuckDB vector (dict_len + 1 entries, the last one NULL), cached while the library hands out
//! the same dictionary version -- ColumnArrowToDuckDBDictionary caches it per ArrowArray for the same reason.
struct MiDictionaryCache {
  const void* version = nullptr;
  unique_ptr<Vector> base;
};

inline void MiSetValidity(Vector& vec, const mi_vector& mv, idx_t count) {
  auto& mask = FlatVector::Validity(vec);
  if (!mv.validity) {
    mask.Reset();  // NULL = every row valid: DuckDB's unset ValidityMask
    return;
  }
  if (mv.validity_shift == 0) {
    mask.Initialize(reinterpret_cast<validity_t*>(mv.validity), count);  // aliases the result slot, like the data
    return;
  }
  // child windows of lists start at any row: the words are shared with the rows in front, so the bits are re-based
  ValidityMask window(reinterpret_cast<validity_t*>(mv.validity), NumericCast<idx_t>(mv.validity_shift) + count);
  mask.Initialize(count);
  mask.Slice(window, NumericCast<idx_t>(mv.validity_shift), count);
}

//! mi_vector (tree) -> DuckDB vector of logical type `type`, zero-copy.  `dicts` has one slot per (column, node).
inline void MiVectorToDuck(const mi_vector& mv, const LogicalType& type, Vector& vec, idx_t count,
                           vector<MiDictionaryCache>& dicts, idx_t& dict_slot) {
  if (mv.kind == MI_K_DICT) {
    // indices were turned into a selection vector on the GPU (NULL -> dict_len); the values are decoded once per version
    if (dict_slot >= dicts.size()) {
      dicts.resize(dict_slot + 1);
    }
    auto& cache = dicts[dict_slot++];
    if (cache.version != mv.dictionary || !cache.base) {
      cache.base = make_uniq<Vector>(type, data_ptr_cast(const_cast<void*>(mv.dictionary)));
      FlatVector::Validity(*cache.base)
          .Initialize(reinterpret_cast<validity_t*>(const_cast<mi_validity_t*>(mv.dictionary_validity)),
                      NumericCast<idx_t>(mv.dict_len) + 1);
      cache.version = mv.dictionary;
    }
    SelectionVector sel(reinterpret_cast<sel_t*>(mv.data));
    vec.Slice(*cache.base, sel, count);
    return;
  }
  switch (type.InternalType()) {
    case PhysicalType::LIST: {
      // list_entry_t rows are relative to the child vector this chunk carries (mi_vector.children[0], already windowed)
      FlatVector::SetData(vec, data_ptr_cast(mv.data));
      MiSetValidity(vec, mv, count);
      D_ASSERT(mv.n_children == 1);
      const auto& child = mv.children[0];
      auto& entry = ListVector::GetEntry(vec);
      MiVectorToDuck(child, ListType::GetChildType(type), entry, NumericCast<idx_t>(child.count), dicts, dict_slot);
      ListVector::SetListSize(vec, NumericCast<idx_t>(child.count));
      break;
    }
    case PhysicalType::STRUCT: {
      MiSetValidity(vec, mv, count);
      auto& entries = StructVector::GetEntries(vec);
      D_ASSERT(NumericCast<idx_t>(mv.n_children) == entries.size());
      for (idx_t i = 0; i < entries.size(); i++) {
        MiVectorToDuck(mv.children[i], StructType::GetChildType(type, i), *entries[i], NumericCast<idx_t>(mv.children[i].count),
                       dicts, dict_slot);
      }
      break;
    }
    case PhysicalType::ARRAY: {
      MiSetValidity(vec, mv, count);
      D_ASSERT(mv.n_children == 1);
      auto& entry = ArrayVector::GetEntry(vec);
      MiVectorToDuck(mv.children[0], ArrayType::GetChildType(type), entry, NumericCast<idx_t>(mv.children[0].count), dicts,
                     dict_slot);
      break;
    }
    default:
      // fixed width, string_t (long strings point into the record-batch body the lease keeps alive), BOOLEAN bytes
      FlatVector::SetData(vec, data_ptr_cast(mv.data));
      MiSetValidity(vec, mv, count);
      break;
  }
}

//! One mi_scan_next -> `output`.  Returns false when the scan is exhausted.  With a pushed-down filter the chunk is sliced
//! by the selection vector the GPU produced; chunks in which no row passed are skipped here, because an empty DataChunk
//! means "exhausted" to DuckDB.
inline bool MiScanIntoChunk(mi_scan* scan, const vector<LogicalType>& types, DataChunk& output,
                            vector<MiDictionaryCache>& dicts) {
  while (true) {
    mi_data_chunk ch;
    MiCheck(mi_scan_next(scan, &ch));
    if (ch.size == 0) {
      output.SetCardinality(0);
      return false;
    }
    if (ch.sel && ch.sel_count == 0) {
      continue;
    }
    D_ASSERT(NumericCast<idx_t>(ch.n_columns) == output.ColumnCount());
    idx_t dict_slot = 0;
    for (idx_t c = 0; c < output.ColumnCount(); c++) {
      MiVectorToDuck(ch.columns[c], types[c], output.data[c], NumericCast<idx_t>(ch.size), dicts, dict_slot);
    }
    output.SetCardinality(NumericCast<idx_t>(ch.size));
    if (ch.sel) {
      SelectionVector sel(reinterpret_cast<sel_t*>(const_cast<mi_sel_t*>(ch.sel)));
      output.Slice(sel, NumericCast<idx_t>(ch.sel_count));
    }
    return true;
  }
}

// ------------------------------------------------------------------------------------------------ DataChunk -> sink
//! A flattened DataChunk as the mi_data_chunk the writer entry points take; `pool` owns the child arrays.
struct MiChunkView {
  mi_data_chunk chunk;
  vector<mi_vector> columns;
  vector<unique_ptr<vector<mi_vector>>> pool;
};

inline void DuckVectorToMi(Vector& vec, idx_t count, mi_vector& out, MiChunkView& view) {
  memset(&out, 0, sizeof(out));
  out.count = NumericCast<int64_t>(count);
  const auto& type = vec.GetType();
  auto add_children = [&](idx_t n) -> mi_vector* {
    view.pool.push_back(make_uniq<vector<mi_vector>>(n));
    out.children = view.pool.back()->data();
    out.n_children = NumericCast<int32_t>(n);
    return view.pool.back()->data();
  };
  switch (type.InternalType()) {
    case PhysicalType::LIST: {
      out.data = FlatVector::GetData(vec);  // list_entry_t{offset, length}; the sink gathers the children in list order
      out.validity = reinterpret_cast<mi_validity_t*>(FlatVector::Validity(vec).GetData());
      auto kids = add_children(1);
      DuckVectorToMi(ListVector::GetEntry(vec), ListVector::GetListSize(vec), kids[0], view);
      break;
    }
    case PhysicalType::STRUCT: {
      out.validity = reinterpret_cast<mi_validity_t*>(FlatVector::Validity(vec).GetData());
      auto& entries = StructVector::GetEntries(vec);
      auto kids = add_children(entries.size());
      for (idx_t i = 0; i < entries.size(); i++) {
        DuckVectorToMi(*entries[i], count, kids[i], view);
      }
      break;
    }
    case PhysicalType::ARRAY: {
      out.validity = reinterpret_cast<mi_validity_t*>(FlatVector::Validity(vec).GetData());
      auto kids = add_children(1);
      DuckVectorToMi(ArrayVector::GetEntry(vec), count * ArrayType::GetSize(type), kids[0], view);
      break;
    }
    default:
      out.data = FlatVector::GetData(vec);
      out.validity = reinterpret_cast<mi_validity_t*>(FlatVector::Validity(vec).GetData());  // nullptr = all valid
      break;
  }
}

inline void DuckChunkToMi(DataChunk& input, MiChunkView& view) {
  input.Flatten();  // nested children included: the sink reads flat vectors only (ArrowAppender takes UnifiedVectorFormat)
  view.pool.clear();
  view.columns.assign(input.ColumnCount(), mi_vector {});
  for (idx_t c = 0; c < input.ColumnCount(); c++) {
    DuckVectorToMi(input.data[c], input.size(), view.columns[c], view);
  }
  memset(&view.chunk, 0, sizeof(view.chunk));
  view.chunk.size = NumericCast<int64_t>(input.size());
  view.chunk.sel_count = view.chunk.size;
  view.chunk.source_rows = view.chunk.size;
  view.chunk.n_columns = NumericCast<int32_t>(input.ColumnCount());
  view.chunk.columns = view.columns.data();
}

// ------------------------------------------------------------------------------------------------ filter pushdown
//! Filters DuckDB offers to a scan (pushdown_complex_filter) -> mi_filter_node trees.  The reference registers
//! filter_pushdown = false (src/scanner/read_arrow.cpp:47-48, src/scanner/scan_arrow_ipc.cpp:60-61): DuckDB's own filter
//! above the scan yields the same rows.  An expression is taken only when every leaf of it translates; whatever is not
//! taken stays in `filters`, i.e. above the scan.
struct MiPushedFilter {
  vector<mi_filter_node> nodes;  // nodes[0] = AND over the taken expressions
  // storage the nodes point into
  vector<unique_ptr<string>> strings;
  vector<unique_ptr<vector<int64_t>>> in_lists;
  vector<unique_ptr<vector<const char*>>> str_ptrs;
  vector<unique_ptr<vector<int32_t>>> str_lens;
  bool Empty() const { return nodes.size() <= 1; }
};

inline bool MiConstantToInt(const Value& v, int64_t& out) {
  if (v.IsNull()) {
    return false;
  }
  switch (v.type().InternalType()) {
    case PhysicalType::BOOL:
      out = v.GetValueUnsafe<bool>() ? 1 : 0;
      return true;
    case PhysicalType::INT8:
      out = v.GetValueUnsafe<int8_t>();
      return true;
    case PhysicalType::INT16:
      out = v.GetValueUnsafe<int16_t>();
      return true;
    case PhysicalType::INT32:
      out = v.GetValueUnsafe<int32_t>();  // DATE is stored as int32 days, DECIMAL(<=9) as int32
      return true;
    case PhysicalType::INT64:
      out = v.GetValueUnsafe<int64_t>();  // TIMESTAMP / TIME micros, DECIMAL(<=18) as the stored integer
      return true;
    case PhysicalType::UINT8:
      out = v.GetValueUnsafe<uint8_t>();
      return true;
    case PhysicalType::UINT16:
      out = v.GetValueUnsafe<uint16_t>();
      return true;
    case PhysicalType::UINT32:
      out = v.GetValueUnsafe<uint32_t>();
      return true;
    case PhysicalType::UINT64: {
      const auto u = v.GetValueUnsafe<uint64_t>();
      if (u > NumericCast<uint64_t>(NumericLimits<int64_t>::Maximum())) {
        return false;  // constants travel as int64
      }
      out = NumericCast<int64_t>(u);
      return true;
    }
    default:
      return false;  // floats, hugeints, intervals: compared by DuckDB above the scan
  }
}

class MiFilterTranslator {
 public:
  MiFilterTranslator(const LogicalGet& get_p, const vector<string>& column_names_p, MiPushedFilter& out_p)
      : get(get_p), column_names(column_names_p), out(out_p) {
    if (out.nodes.empty()) {
      mi_filter_node root;
      memset(&root, 0, sizeof(root));
      root.op = MI_F_AND;
      out.nodes.push_back(root);
    }
  }

  //! Takes what it can out of `filters`; the root's children are the taken expressions.
  void Take(vector<unique_ptr<Expression>>& filters) {
    vector<vector<mi_filter_node>> taken;
    for (idx_t i = 0; i < filters.size();) {
      vector<mi_filter_node> tree;
      if (Translate(*filters[i], tree)) {
        taken.push_back(std::move(tree));
        filters.erase_at(i);
      } else {
        i++;
      }
    }
    if (taken.empty()) {
      return;
    }
    // children of a node must be contiguous: the roots of the taken trees first, then every tree's descendants
    const auto first = NumericCast<int32_t>(out.nodes.size());
    out.nodes[0].first_child = out.nodes[0].n_children == 0 ? first : out.nodes[0].first_child;
    if (out.nodes[0].n_children != 0) {
      throw InternalException("MiFilterTranslator::Take may be called once per scan");
    }
    out.nodes[0].n_children = NumericCast<int32_t>(taken.size());
    out.nodes.resize(out.nodes.size() + taken.size());
    for (idx_t t = 0; t < taken.size(); t++) {
      Place(taken[t], 0, NumericCast<idx_t>(first) + t);
    }
  }

 private:
  // a translated expression, children by index inside its own vector (node 0 = its root)
  bool Translate(const Expression& expr, vector<mi_filter_node>& tree) {
    tree.emplace_back();
    return TranslateInto(expr, tree, 0);
  }

  const char* ColumnOf(const Expression& e) {
    if (e.GetExpressionClass() != ExpressionClass::BOUND_COLUMN_REF) {
      return nullptr;
    }
    auto& ref = e.Cast<BoundColumnRefExpression>();
    if (ref.depth != 0 || ref.binding.table_index != get.table_index) {
      return nullptr;
    }
    const auto& ids = get.GetColumnIds();
    if (ref.binding.column_index >= ids.size()) {
      return nullptr;
    }
    const auto col = ids[ref.binding.column_index].GetPrimaryIndex();
    if (col >= column_names.size()) {
      return nullptr;  // virtual columns (filename, ...) are produced above the reader
    }
    return column_names[col].c_str();
  }

  bool SetConstant(mi_filter_node& n, const Value& v) {
    if (v.type().id() == LogicalTypeId::VARCHAR || v.type().id() == LogicalTypeId::BLOB) {
      if (v.IsNull()) {
        return false;
      }
      // = <> < <= > >= : byte-wise, like string_t's own comparison (a column with a non-default collation never gets here: its
      // comparisons arrive wrapped in collation functions, which are not column references)
      out.strings.push_back(make_uniq<string>(StringValue::Get(v)));
      n.str_value = out.strings.back()->data();
      n.str_len = NumericCast<int32_t>(out.strings.back()->size());
      return true;
    }
    return MiConstantToInt(v, n.value);
  }

  static int32_t FlipComparison(int32_t op) {
    switch (op) {
      case MI_F_LT:
        return MI_F_GT;
      case MI_F_LE:
        return MI_F_GE;
      case MI_F_GT:
        return MI_F_LT;
      case MI_F_GE:
        return MI_F_LE;
      default:
        return op;
    }
  }

  bool TranslateInto(const Expression& expr, vector<mi_filter_node>& tree, idx_t at) {
    memset(&tree[at], 0, sizeof(mi_filter_node));
    switch (expr.GetExpressionClass()) {
      case ExpressionClass::BOUND_COMPARISON: {
        auto& cmp = expr.Cast<BoundComparisonExpression>();
        int32_t op;
        switch (cmp.GetExpressionType()) {
          case ExpressionType::COMPARE_EQUAL:
            op = MI_F_EQ;
            break;
          case ExpressionType::COMPARE_NOTEQUAL:
            op = MI_F_NE;
            break;
          case ExpressionType::COMPARE_LESSTHAN:
            op = MI_F_LT;
            break;
          case ExpressionType::COMPARE_LESSTHANOREQUALTO:
            op = MI_F_LE;
            break;
          case ExpressionType::COMPARE_GREATERTHAN:
            op = MI_F_GT;
            break;
          case ExpressionType::COMPARE_GREATERTHANOREQUALTO:
            op = MI_F_GE;
            break;
          default:
            return false;  // IS [NOT] DISTINCT FROM
        }
        const Expression* col = cmp.left.get();
        const Expression* constant = cmp.right.get();
        if (col->GetExpressionClass() == ExpressionClass::BOUND_CONSTANT) {
          std::swap(col, constant);
          op = FlipComparison(op);
        }
        const char* name = ColumnOf(*col);
        if (!name || constant->GetExpressionClass() != ExpressionClass::BOUND_CONSTANT) {
          return false;
        }
        tree[at].op = op;
        tree[at].column = name;
        return SetConstant(tree[at], constant->Cast<BoundConstantExpression>().value);
      }
      case ExpressionClass::BOUND_OPERATOR: {
        auto& opx = expr.Cast<BoundOperatorExpression>();
        const auto kind = opx.GetExpressionType();
        if (kind == ExpressionType::OPERATOR_IS_NULL || kind == ExpressionType::OPERATOR_IS_NOT_NULL) {
          const char* name = opx.children.size() == 1 ? ColumnOf(*opx.children[0]) : nullptr;
          if (!name) {
            return false;
          }
          tree[at].op = kind == ExpressionType::OPERATOR_IS_NULL ? MI_F_IS_NULL : MI_F_IS_NOT_NULL;
          tree[at].column = name;
          return true;
        }
        if (kind != ExpressionType::COMPARE_IN || opx.children.size() < 2) {
          return false;
        }
        const char* name = ColumnOf(*opx.children[0]);
        if (!name) {
          return false;
        }
        tree[at].op = MI_F_IN;
        tree[at].column = name;
        tree[at].n_values = NumericCast<int32_t>(opx.children.size() - 1);
        const bool strings = opx.children[1]->return_type.id() == LogicalTypeId::VARCHAR ||
                             opx.children[1]->return_type.id() == LogicalTypeId::BLOB;
        if (strings) {
          out.str_ptrs.push_back(make_uniq<vector<const char*>>());
          out.str_lens.push_back(make_uniq<vector<int32_t>>());
        } else {
          out.in_lists.push_back(make_uniq<vector<int64_t>>());
        }
        for (idx_t i = 1; i < opx.children.size(); i++) {
          if (opx.children[i]->GetExpressionClass() != ExpressionClass::BOUND_CONSTANT) {
            return false;
          }
          const auto& v = opx.children[i]->Cast<BoundConstantExpression>().value;
          if (v.IsNull()) {
            return false;  // x IN (..., NULL): three-valued, left to DuckDB
          }
          if (strings) {
            out.strings.push_back(make_uniq<string>(StringValue::Get(v)));
            out.str_ptrs.back()->push_back(out.strings.back()->data());
            out.str_lens.back()->push_back(NumericCast<int32_t>(out.strings.back()->size()));
          } else {
            int64_t x;
            if (!MiConstantToInt(v, x)) {
              return false;
            }
            out.in_lists.back()->push_back(x);
          }
        }
        if (strings) {
          tree[at].str_values = out.str_ptrs.back()->data();
          tree[at].str_lens = out.str_lens.back()->data();
        } else {
          tree[at].values = out.in_lists.back()->data();
        }
        return true;
      }
      case ExpressionClass::BOUND_CONJUNCTION: {
        auto& conj = expr.Cast<BoundConjunctionExpression>();
        const auto first = tree.size();
        tree.resize(tree.size() + conj.children.size());
        tree[at].op = conj.GetExpressionType() == ExpressionType::CONJUNCTION_AND ? MI_F_AND : MI_F_OR;
        tree[at].first_child = NumericCast<int32_t>(first);
        tree[at].n_children = NumericCast<int32_t>(conj.children.size());
        for (idx_t i = 0; i < conj.children.size(); i++) {
          if (!TranslateInto(*conj.children[i], tree, first + i)) {
            return false;
          }
        }
        return true;
      }
      default:
        return false;
    }
  }

  //! copies node `from` of `tree` to out.nodes[to], its children (contiguous in `tree` already) behind the current end
  void Place(const vector<mi_filter_node>& tree, idx_t from, idx_t to) {
    mi_filter_node n = tree[from];
    if (n.op == MI_F_AND || n.op == MI_F_OR) {
      const auto first = out.nodes.size();
      out.nodes.resize(out.nodes.size() + NumericCast<idx_t>(n.n_children));
      for (idx_t i = 0; i < NumericCast<idx_t>(n.n_children); i++) {
        Place(tree, NumericCast<idx_t>(n.first_child) + i, first + i);
      }
      n.first_child = NumericCast<int32_t>(first);
    }
    out.nodes[to] = n;
  }

  const LogicalGet& get;
  const vector<string>& column_names;
  MiPushedFilter& out;
};

//! mi_scan_set_filter between bind and init.  MI_ENOTSUP here means the library narrowed what it accepts after the
//! translator was written: an internal error of the glue, not a user error (the filters were already taken from DuckDB).
inline void MiApplyFilter(mi_scan* scan, const MiPushedFilter& f) {
  if (f.Empty()) {
    return;
  }
  MiCheck(mi_scan_set_filter(scan, f.nodes.data(), NumericCast<int32_t>(f.nodes.size()), 0));
}

}  // namespace ext_nanoarrow
}  // namespace duckdb
