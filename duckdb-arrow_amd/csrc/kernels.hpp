// kernels.hpp -- launch interface of the gfx950 transcode kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "../../include/mi_arrow_ipc.h"

namespace miarrow {
namespace device {

constexpr int kBlockThreads = 256;   // 4 waves of 64
constexpr int kTileRows = 2048;      // rows per tile == DuckDB STANDARD_VECTOR_SIZE: one tile is one output vector
#ifndef MI_COPY_TILE_ROWS
#define MI_COPY_TILE_ROWS 2048
#endif
#ifndef MI_DEC_TILE_ROWS
#define MI_DEC_TILE_ROWS 2048
#endif
constexpr int kCopyTileRows = MI_COPY_TILE_ROWS;  // copy / dec128 tiles may span several vectors (multiples of 2048)
constexpr int kDecTileRows = MI_DEC_TILE_ROWS;
int TileRowsOfClass(int cls);

// Kernel classes: a plan groups its tasks by class and launches one kernel per non-empty class.
enum KernelClass { kClassCopy = 0, kClassDec128 = 1, kClassString = 2, kClassMisc = 3, kClassEncFixed = 4, kClassEncString = 5, kClassGather = 6,
                   kNumClasses = 7 };
int ClassOfKind(int32_t kind);  // -1 for an unknown kind
//! class of a task: gather mode (mi_col_task.sel) has its own kernel; -1 when the kind is unknown or cannot be gathered
int ClassOfTask(const mi_col_task& t);
bool KindCanGather(int32_t kind);

// One launch over a device-resident task table slice.  `tile_begin[i]` = first tile of task i within the slice,
// tile_begin[n_tasks] = total_tiles.  `status` accumulates MI_ST_* bits.
// `tile_task[tile]` = task index (within the slice) of every tile of the slice.  One workgroup per tile.
hipError_t LaunchTranscode(int cls, const mi_col_task* d_tasks, const uint32_t* d_tile_begin, const uint32_t* d_tile_task,
                           int32_t n_tasks, uint32_t total_tiles, uint32_t* d_status, uint32_t misc_groups, hipStream_t stream);
// which transcode_misc kernel owns a kind of the misc class: 0 / 3 common flat kinds (one wave per tile; 3 = 8-byte inputs),
// 1 nested, 2 rare flat
int MiscGroupOfKind(int32_t kind);

//! Fused consumer: sum(a * b) over the rows that pass up to 4 conjunctive range filters, straight
//! from the decoded vectors in HBM.  acc = {sum low 64 bits, sum high 64 bits (two's complement), rows selected}.
struct AggSumProductArgs {
  const void* fcol[4];
  const uint64_t* fvalid[4];
  int32_t fwidth[4];
  int64_t lo[4], hi[4];
  int32_t n_filters;
  const void* a;
  const void* b;
  const uint64_t* avalid;
  const uint64_t* bvalid;
  int32_t awidth, bwidth;
  int64_t nrows;
};
hipError_t LaunchAggSumProduct(const AggSumProductArgs& args, unsigned long long* d_acc, int num_cus, hipStream_t stream);

// K8: LZ4_FRAME buffers of one record batch, decompressed in HBM (kernels_lz4.hip).  Positions are 31-bit: a compressed or
// decompressed body of 2 GiB or more takes the host decompressor.
struct Lz4BlockDev {
  uint32_t comp_off, comp_size;   // the block's bytes inside the compressed body
  uint32_t buffer;                // index into Lz4Args::buffers
  uint32_t stored;                // 1 = the block holds its bytes uncompressed
  uint32_t seq_base, seq_cap;     // its slice of the sequence-descriptor scratch, 256 equal parts: LZ4 256 x (ceil(comp_size / 256) / 3 + 2), ZSTD count + 1 rounded up
};
struct Lz4BufferDev {
  uint64_t out_off, out_len;      // where the buffer lies in the decompressed body, and its declared length
  uint32_t first_block, n_blocks;
  uint32_t block_max, _pad;       // the frame's maximum block size (BD byte)
};
struct Lz4Args {
  const uint8_t* comp;            // compressed body (device)
  uint8_t* out;                   // decompressed body (device), out_size bytes
  uint64_t out_size;
  uint64_t max_buffer_len;        // longest decompressed LZ4 buffer: bounds the number of resolve rounds
  const Lz4BlockDev* blocks;
  const Lz4BufferDev* buffers;
  uint32_t n_blocks, n_buffers;
  uint32_t max_block_comp, _pad;  // largest compressed block; _pad: set by the launcher (which parse kernel owns which blocks)
  uint32_t min_block_comp;        // smallest compressed (not stored) block
  uint32_t max_buffer_blocks;     // most blocks any one buffer has
  void* seq;                      // 16 bytes per sequence, in the slices of lz4_parse's 256 lanes per block
  uint32_t* seq_off;              // 4 bytes per sequence
  uint32_t* lane_out;             // per block and lane: output position (inside the block) its slice begins at ...
  uint32_t* lane_nseq;            // ... and its number of sequences
  uint32_t* link;                 // 4 bytes per decompressed byte (rounded up to 16 bytes); no preset needed
  uint32_t* block_out_size;       // per block
  uint32_t* block_nseq;
  uint64_t* block_out_base;
  uint32_t* chunk_base;           // per block (+ 1): first chunk (<= 8 KiB of one block's output) of the block, k8_chunk_map
  uint32_t* buffer_ok;            // per buffer
  uint8_t* mark;                  // one byte per decompressed byte (rounded up to 16), zeroed: 1 = an open word of another tile points here
  uint32_t* skel;                 // the skeleton list: up to one entry per decompressed byte
  uint32_t* round_left;           // 40 words, zeroed
  uint32_t* status;               // MI_ST_DECOMPRESS
  // ZSTD batches (NULL for LZ4): one zstd::BlockInfo per block (zstd_format.hpp); the decoded literals of a block go to
  // literals[BlockInfo::lit_pos ..] -- the same allocation as `comp`, behind the compressed body, so that a sequence's literal
  // source is one position whichever section it came from
  const void* zblocks;
  uint8_t* literals;
  uint32_t* rep_state;            // ZSTD: 4 words per block and slice -- the slice's effect on the repeat offsets, then (zstd_layout) the offsets it starts from
};
inline uint32_t Lz4SeqCapacity(uint32_t comp_size) { return 256u * (((comp_size + 255u) / 256u) / 3u + 2u); }
hipError_t LaunchLz4Decompress(const Lz4Args& args, int num_cus, hipStream_t stream);

// Late materialisation: tasks with mi_col_task.sel decode only the selected rows, compacted per window (kernels_gather.hip)
// d_window_base: total_tiles words of scratch (first output row of every window, filled by the launch)
hipError_t LaunchGather(const mi_col_task* d_tasks, const uint32_t* d_tile_begin, const uint32_t* d_tile_task, int32_t n_tasks,
                        uint32_t total_tiles, int64_t* d_window_base, uint32_t* d_status, hipStream_t stream);

// K6: pushed-down predicate -> one ascending selection vector per 2048-row window (kernels_filter.hip).  The predicate is
// in conjunctive normal form: leaves in clause order, kLeafEndsClause on the last leaf of every clause (a clause is the
// OR of its leaves, the filter the AND of its clauses).
constexpr int kLeafRange = 1;       // lo <= v <= hi on the stored integer (negated with kLeafNegate)
constexpr int kLeafIsNull = 2;
constexpr int kLeafIsNotNull = 3;
constexpr int kLeafIn = 4;          // v is one of in_values[0 .. n_in)
constexpr int kLeafStrIn = 5;       // string_t column: the row equals one of n_in byte strings.  in_values holds 3 words per
                                    // constant: {len | dword1 << 32, dword2 | dword3 << 32, device address of the bytes} where
                                    // dword1..3 are the string_t image (inline bytes, or prefix + 0 + 0 for > 12 bytes)
constexpr int kLeafDictMap = 6;     // dictionary-encoded string column: data = the uint32 selection vector, in_values = one byte per
                                    // dictionary entry (n_in of them; 0 no, 1 yes, 2 NULL entry), lo = 0 match, 1 no match, 2 IS NULL, 3 IS NOT NULL
constexpr int kLeafStrRange = 7;    // string_t column: lower bound <(=) row <(=) upper bound, byte-wise.  in_values: two constants in kLeafStrIn's
                                    // layout (lower, upper); n_in = bit 0 has lower, bit 1 lower inclusive, bit 2 has upper, bit 3 upper inclusive
constexpr int kLeafUnsigned = 1;    // flags: the column holds unsigned integers
constexpr int kLeafNegate = 2;      //        NOT (range / in-list); NULL still fails
constexpr int kLeafEndsClause = 4;
constexpr int kLeafBias = 8;        //        uint64 column: values and constants are compared after x ^ 2^63
constexpr int kMaxFilterLeaves = 24;
struct FilterLeafDev {
  const void* data;                 // decoded fixed-width vector (device), NULL for IS [NOT] NULL
  const uint64_t* validity;         // validity words (device) or NULL = all valid
  const int64_t* in_values;         // kLeafIn (device)
  int64_t lo, hi;
  int32_t op, width, flags, n_in;
};
struct FilterProgram {
  int32_t n_leaves;
  int32_t _pad;
  FilterLeafDev leaves[kMaxFilterLeaves];
};
hipError_t LaunchFilterProgram(const FilterProgram& prog, int64_t nrows, mi_sel_t* sel_out, uint32_t* count_out, hipStream_t stream);
// lo <= v < hi on one column (mi_filter_range)
hipError_t LaunchFilterRange(const void* values, int32_t width, const void* validity, int64_t nrows, int64_t lo,
                             int64_t hi, mi_sel_t* sel_out, uint32_t* count_out, hipStream_t stream);

// K7 launches.  The string / list kernel is a single pass (decoupled look-back across the tiles of a column); it needs
// 2 * total_tiles + 1 state words (zeroed by the launch).
hipError_t LaunchEncodeFixed(const mi_col_task* d_tasks, const uint32_t* d_tile_begin, const uint32_t* d_tile_task,
                             int32_t n_tasks, uint32_t total_tiles, int64_t* d_null_counts, hipStream_t stream);
hipError_t LaunchEncodeString(const mi_col_task* d_tasks, const uint32_t* d_tile_begin, const uint32_t* d_tile_task,
                              int32_t n_tasks, uint32_t total_tiles, int64_t* d_tile_state, int64_t* d_null_counts,
                              uint32_t* d_status, bool has_lists, hipStream_t stream);

}  // namespace device
}  // namespace miarrow
