// batch_planner.hpp -- ONE planner for both ways a record batch reaches the kernels: the scan operator's pipeline slots
// (scan_operator.cpp) and the HBM-resident super-batch (hbm_stream.cpp, what bench.py times).
//
// For every field node of a decoded record batch (depth first, like Arrow's own node order) the planner
//   * reserves the node's DuckDB vector in an output arena (data array + validity words),
//   * emits the transcode task (mi_col_task) that fills it from the Arrow buffers of the message body,
//   * records what a DataChunk needs later (PlannedNode: offsets, child windows, dictionary id, aliasing).
// It restates, for the GPU layout, what DuckDB's ArrowToDuckDB does per column when the reference calls it
// (src/scanner/scan_arrow_ipc.cpp:56, src/file_scanner/arrow_file_scan.cpp:68-72): DirectConversion = alias or copy,
// GetValidityMask = validity words (left unset for a column without NULLs), ConvertArrowListOffsets = child windows.
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "ipc_stream_reader.hpp"

namespace miarrow {

//! One field node of one record batch after planning.
struct PlannedNode {
  size_t data_off = 0;       // arena offset of the vector data (unused when aliased)
  int64_t valid_off = -1;    // arena offset of the validity words; -1: not materialised = all valid
  int32_t kind = 0, width = 0, arrow_type = 0, depth = 0;
  int64_t param = 0, nrows = 0, null_count = 0;
  int32_t parent = -1;                 // index into BatchPlanner::nodes
  std::vector<int64_t> win;            // first row (in this node's row space) of every top-level 2048-row window, + end
  std::vector<int32_t> children;       // indices into BatchPlanner::nodes
  int64_t dict_id = -1;                // MI_K_DICT: which dictionary the selection vector indexes
  int64_t alias_body_off = -1;         // >= 0: zero-copy DirectConversion, the values are the body bytes at this offset
  uint64_t ptr_base = 0;               // string kinds: string_t long pointers = ptr_base + offset inside the data buffer
  int64_t heap_size = 0;               //               bytes of that data buffer
  int32_t source_node = -1;            // index into DecodedBatch::nodes
  int32_t task = -1;                   // index into BatchPlanner::tasks (-1: no task -- empty, aliased or nothing to do)
};

struct PlannerOptions {
  size_t array_align = 256;            // every output array starts on this boundary of the arena
  bool zero_copy_direct = false;       // plain fixed-width columns without NULLs alias the body instead of being copied
  bool unset_all_valid = false;        // columns with null_count == 0 get no validity words (the reference leaves the mask unset)
};

//! Where one batch's bytes are and will be seen.
struct BatchPlacement {
  const DecodedBatch* batch = nullptr;
  const uint8_t* in_base = nullptr;    // DEVICE address of body byte 0 (what the kernels read)
  uint64_t consumer_base = 0;          // address the consumer of the vectors sees for body byte 0 (string_t pointers)
  const std::vector<char>* no_alias = nullptr;  // per DecodedBatch node: 1 = materialise even when aliasable (filter columns)
  int64_t alloc_rows = -1;             // >= 0: reserve this many rows per top-level column instead of its length (compaction)
  //! length of the decoded dictionary a MI_K_DICT node indexes; throws when the id is unknown
  std::function<int64_t(int64_t dict_id)> dict_len;
};

class BatchPlanner {
 public:
  explicit BatchPlanner(const PlannerOptions& o) : opts(o) {}

  PlannerOptions opts;
  std::vector<PlannedNode> nodes;
  //! out_data / out_validity / out_aux hold ARENA OFFSETS (+1, 0 = none) and STRVIEW / nested-list buf2 an AUX WORD INDEX
  //! until Rebase() turns them into device addresses
  std::vector<mi_col_task> tasks;
  std::vector<uint64_t> aux;                              // list window tables, string-view buffer tables
  std::vector<std::pair<int64_t, int64_t>> upload;        // body byte ranges the tasks of the LAST batch read
  size_t arena_bytes = 0;                                 // bytes reserved so far

  void Clear();
  //! Plans one top-level column (and its descendants); returns its index in `nodes`.  extra_rows: slots reserved after
  //! the column's rows (a decoded dictionary keeps one NULL entry at index dict_len).
  int32_t AddColumn(const BatchPlacement& where, int32_t decoded_node, int64_t extra_rows = 0);
  //! Reserves an all-NULL column of `n` rows (union_by_name: column absent from a file); returns {data_off, valid_off}
  std::pair<size_t, size_t> AddAbsentColumn(int64_t n, int32_t width);
  //! Raw reservation (selection vectors, counters)
  size_t Reserve(size_t bytes);
  //! Arena / aux offsets -> device addresses for the tasks [first_task, tasks.size())
  void Rebase(size_t first_task, uint8_t* arena_base, const uint8_t* aux_base);

 private:
  void Alloc(int64_t rows, int32_t width, bool with_validity, size_t* data_off, int64_t* valid_off);
  int32_t AddNode(const BatchPlacement& where, int32_t ni, std::vector<int64_t> win, bool win_is_tiles, int64_t parent_valid_off,
                  int32_t parent_div, int32_t parent, int64_t extra_rows);
  std::vector<std::pair<size_t, size_t>> aux_fixups;      // (task index, first aux word)
  friend class HbmStream;
};

}  // namespace miarrow
