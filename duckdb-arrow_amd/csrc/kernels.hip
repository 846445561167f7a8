// kernels.hip -- hand-written gfx950 (CDNA4) kernels for the Arrow IPC <-> DuckDB vector transcode.
//
// Everything here is HBM-bound integer / byte work ("no MFMA"): the design rules that matter are
// coalesced 16-byte-per-lane accesses, enough bytes in flight per CU, a handful of launches over a device-resident
// task table for ANY number of record batches (a 122880-row record batch is ~40 MB of traffic = ~7 us at HBM speed,
// so per-column-per-batch launches would be launch-bound), and a persistent grid of 8 x 256-thread workgroups per CU
// that strides over 2048-row tiles.
//
// One TILE = 2048 rows of one column of one record batch = exactly one DuckDB vector (STANDARD_VECTOR_SIZE), so
// tile t of a column writes vector t: data at out_data + t*2048*width, validity words at out_validity + t*32.
// Tiles are numbered across all tasks of a kernel class; workgroup b handles tiles b, b+G, b+2G, ... so that at any
// moment the resident workgroups stream adjacent tiles (adjacent HBM pages / channels).  Workgroups b and b+8 share
// an XCD (and its L2); neighbouring tiles only share the cache lines at their seams, so no XCD remap is needed.
//
// Kernel classes: each class is its own kernel so that it gets the register budget of its own inner loop (a single
// switch over all kinds needed 154 VGPRs = 3 waves/SIMD).  A plan groups its tasks by class (engine.cpp).
//
// Semantics restated per kernel from DuckDB's ArrowToDuckDB / ArrowAppender (call sites in the reference:
// src/scanner/scan_arrow_ipc.cpp:56, src/file_scanner/arrow_file_scan.cpp:68-72,
// src/writer/column_data_collection_serializer.cpp:85); canonical values for slots upstream leaves undefined:
// NULL rows of converted columns = 0, validity pad bits = 1.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

#include <algorithm>
#include <cstdlib>
#include <string>

namespace miarrow {
namespace device {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Pointers inside a task descriptor are loaded from memory, so the compiler only knows them as generic ("flat")
// addresses.  Everything they point to is HBM: casting to the global address space turns flat_load/flat_store into
// global_load/global_store (no LDS aperture check, vmcnt-only accounting).
template <typename T>
using gptr = T __attribute__((address_space(1)))*;
template <typename T>
__device__ __forceinline__ gptr<const T> GC(const void* p) {
  return (gptr<const T>)p;
}
template <typename T>
__device__ __forceinline__ gptr<T> GM(void* p) {
  return (gptr<T>)p;
}

__device__ __forceinline__ void raise(uint32_t* status, uint32_t bits) {
  if (bits) atomicOr(status, bits);
}

// Finds the task that owns a tile: largest i with tile_begin[i] <= tile.  The tile index is wave-uniform, so the
// search runs on the scalar unit (s_load) and the task descriptor lands in SGPRs.
__device__ __forceinline__ int find_task(const uint32_t* __restrict__ tile_begin, int n_tasks, uint32_t tile) {
  int lo = 0, hi = n_tasks;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_begin[mid] <= tile) lo = mid; else hi = mid;
  }
  return lo;
}

// tile_task (optional): task index of every tile, so the owner of a tile is one scalar load instead of a ~13-step
// dependent binary search
#define MI_TILE_PROLOGUE() MI_TILE_PROLOGUE_ROWS(kTileRows)
#define MI_TILE_PROLOGUE_ROWS(TILE_ROWS)                                                            \
  const int ti = __builtin_amdgcn_readfirstlane(                                                    \
      tile_task ? static_cast<int>(tile_task[tile]) : find_task(tile_begin, n_tasks, tile));       \
  const mi_col_task& t = tasks[ti];                                                                 \
  const int64_t row0 = static_cast<int64_t>(tile - tile_begin[ti]) * (TILE_ROWS);                   \
  const int64_t left = t.nrows - row0;                                                              \
  const int n = left < (TILE_ROWS) ? static_cast<int>(left) : (TILE_ROWS);

// ---------------------------------------------------------------------------------------------------- K1
// Validity bitmap -> DuckDB validity_t words for one tile.  Word w of the tile holds rows [64w, 64w+64); the source
// bit position is row_offset + row0 + 64w, realigned with a 64-bit funnel shift when it is not word aligned (the CPU
// path's "copy n+1 bytes and shift right by o%8").  One lane per output word.
// With a parent (out_aux = validity words of the struct / fixed_size_list vector that owns this column) the parent's
// NULLs propagate into the child (ArrowToDuckDBStruct / ArrowToDuckDBArray).  `s_valid` (LDS, 32 words) receives the
// combined words so the data lanes canonicalise NULL rows with the same mask; returns whether any row can be NULL.
__device__ __forceinline__ bool tile_needs_mask(const mi_col_task& t) {
  return (t.validity != nullptr && t.null_count != 0) || t.out_aux != nullptr;
}

template <int T = kBlockThreads>  // threads of the workgroup that owns the tile
__device__ __forceinline__ void tile_validity(const mi_col_task& t, int64_t row0, int n, uint64_t* s_valid = nullptr) {
  const bool need_mask = s_valid != nullptr && tile_needs_mask(t);
  if (t.out_validity == nullptr && !need_mask) return;
  const int nwords = (n + 63) >> 6;
  for (int lane = threadIdx.x; lane < nwords; lane += T) {
    uint64_t w = ~0ull;
    if (t.validity != nullptr && t.null_count != 0) {
      gptr<const uint64_t> W = GC<uint64_t>(t.validity);
      const int64_t bit = t.row_offset + row0 + 64 * lane;
      const int64_t q = bit >> 6;
      const int sh = static_cast<int>(bit & 63);
      const int64_t last_q = (t.row_offset + t.nrows - 1) >> 6;  // last 8-byte word that holds a bit of this column
      const uint64_t lo = W[q];
      const uint64_t hi = (sh != 0 && q + 1 <= last_q) ? W[q + 1] : 0ull;
      w = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
    }
    if (t.out_aux != nullptr) {
      gptr<const uint64_t> P = GC<uint64_t>(t.out_aux);
      const int64_t r = row0 + 64 * lane;
      if (t.flags <= 1) {
        w &= P[r >> 6];  // same row numbering, same word
      } else {
        const int64_t div = t.flags;
        uint64_t pw = 0;
        for (int i = 0; i < 64; i++) {
          const int64_t pr = (r + i) / div;
          pw |= ((P[pr >> 6] >> (pr & 63)) & 1ull) << i;
        }
        w &= pw;
      }
    }
    const int rem = n - 64 * lane;
    if (rem < 64) w |= ~0ull << rem;  // canonical pad bits
    if (t.out_validity != nullptr) GM<uint64_t>(t.out_validity)[(row0 >> 6) + lane] = w;
    if (need_mask) s_valid[lane] = w;
  }
  if (need_mask) __syncthreads();
}

// Row validity for the data lanes: the combined tile mask in LDS (null_count == 0 and no parent => every row valid,
// the bitmap is not even read: GetValidityMask).
__device__ __forceinline__ bool row_valid(const uint64_t* s_valid, bool need_mask, int r) {
  return !need_mask || ((s_valid[r >> 6] >> (r & 63)) & 1);
}

// ---------------------------------------------------------------------------------------------------- K3a
// Fixed-width direct conversion: a coalesced copy of n*width bytes.  The destination is 16-byte aligned (tiles start
// at multiples of 2048 rows of a 16-byte aligned vector); the source is an IPC buffer, 8-byte aligned, so it takes
// either the 16-byte or the 8-byte lane path (wave-uniform choice).
template <typename V>
__device__ __forceinline__ void copy_vec(gptr<const uint8_t> src, gptr<uint8_t> dst, int bytes) {
  constexpr int VB = sizeof(V);
  const int nvec = bytes / VB;
  gptr<const V> s = (gptr<const V>)src;
  gptr<V> d = (gptr<V>)dst;
  int i = threadIdx.x;
  // 4 independent loads in flight per lane before the first store
#pragma clang loop unroll(disable)
  for (; i + 3 * kBlockThreads < nvec; i += 4 * kBlockThreads) {
    V a = s[i], b = s[i + kBlockThreads], c = s[i + 2 * kBlockThreads], e = s[i + 3 * kBlockThreads];
    d[i] = a;
    d[i + kBlockThreads] = b;
    d[i + 2 * kBlockThreads] = c;
    d[i + 3 * kBlockThreads] = e;
  }
#pragma clang loop unroll(disable)
  for (; i < nvec; i += kBlockThreads) d[i] = s[i];
#pragma clang loop unroll(disable) vectorize(disable)
  for (int j = nvec * VB + threadIdx.x; j < bytes; j += kBlockThreads) dst[j] = src[j];
}

// 16-byte vector whose loads may sit on any 4-byte boundary: gfx950 runs global memory in unaligned-access mode, so a
// global_load_dwordx4 from an 8-mod-16 address is legal; IPC buffers are only 8-byte aligned (after an odd-length
// offsets buffer every following buffer of the body is 8 mod 16), the destination vectors are always 16-byte aligned.
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));

template <bool NT>
__device__ __forceinline__ u32x4 ld16(gptr<const u32x4_a4> p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <bool NT>
__device__ __forceinline__ void st16(gptr<u32x4> p, u32x4 v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <bool NT>
__device__ __forceinline__ void copy_vec16_unaligned_src(gptr<const uint8_t> src, gptr<uint8_t> dst, int bytes) {
  const int nvec = bytes / 16;
  gptr<const u32x4_a4> s = (gptr<const u32x4_a4>)src;
  gptr<u32x4> d = (gptr<u32x4>)dst;
  int i = threadIdx.x;
#pragma clang loop unroll(disable)
  for (; i + 3 * kBlockThreads < nvec; i += 4 * kBlockThreads) {
    u32x4 a = ld16<NT>(s + i), b = ld16<NT>(s + i + kBlockThreads), c = ld16<NT>(s + i + 2 * kBlockThreads),
          e = ld16<NT>(s + i + 3 * kBlockThreads);
    st16<NT>(d + i, a);
    st16<NT>(d + i + kBlockThreads, b);
    st16<NT>(d + i + 2 * kBlockThreads, c);
    st16<NT>(d + i + 3 * kBlockThreads, e);
  }
#pragma clang loop unroll(disable)
  for (; i < nvec; i += kBlockThreads) st16<NT>(d + i, ld16<NT>(s + i));
#pragma clang loop unroll(disable) vectorize(disable)
  for (int j = nvec * 16 + threadIdx.x; j < bytes; j += kBlockThreads) dst[j] = src[j];
}

__device__ __forceinline__ void copy_bytes(gptr<const uint8_t> src, gptr<uint8_t> dst, int bytes, int variant = 0) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst);
  if (variant >= 1 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0) {
    if (variant == 2) copy_vec16_unaligned_src<true>(src, dst, bytes);
    else copy_vec16_unaligned_src<false>(src, dst, bytes);
    return;
  }
  if ((a & 15) == 0) copy_vec<u32x4>(src, dst, bytes);
  else if ((a & 7) == 0) copy_vec<u32x2>(src, dst, bytes);
  else if ((a & 3) == 0) copy_vec<uint32_t>(src, dst, bytes);
  else copy_vec<uint8_t>(src, dst, bytes);
}

template <int VARIANT>
__global__ __launch_bounds__(kBlockThreads) void transcode_copy(const mi_col_task* __restrict__ tasks,
                                                                const uint32_t* __restrict__ tile_begin,
                                                                const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                uint32_t total_tiles, uint32_t* __restrict__ status) {
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE_ROWS(kCopyTileRows);
    tile_validity(t, row0, n);
    const int w = static_cast<int>(t.param);
    copy_bytes(GC<uint8_t>(t.buf1) + (t.row_offset + row0) * w, GM<uint8_t>(t.out_data) + row0 * w, n * w, VARIANT);
  }
}

// ---------------------------------------------------------------------------------------------------- K3b
// decimal128 {u64 lower, i64 upper} -> int16/32/64 for valid rows (Hugeint::TryCast: value fits by precision);
// NULL rows canonical 0.  Each lane reads the whole 16-byte value (the upper half is what proves the range).
template <typename OUT, int VARIANT>
__device__ __forceinline__ void tile_dec128(const mi_col_task& t, int64_t row0, int n, uint32_t* status, const uint64_t* s_valid) {
  gptr<const uint8_t> src = GC<uint8_t>(t.buf1) + (t.row_offset + row0) * 16;
  gptr<OUT> out = GM<OUT>(t.out_data) + row0;
  const bool has_nulls = tile_needs_mask(t);
  const bool a16 = VARIANT >= 1 || (reinterpret_cast<uintptr_t>(src) & 15) == 0;
  uint32_t err = 0;
#pragma unroll 4
  for (int r = threadIdx.x; r < n; r += kBlockThreads) {
    uint64_t lower;
    int64_t upper;
    if (a16) {
      const u32x4 v = ld16<VARIANT == 2>((gptr<const u32x4_a4>)(src + 16 * static_cast<int64_t>(r)));
      lower = static_cast<uint64_t>(v.x) | (static_cast<uint64_t>(v.y) << 32);
      upper = static_cast<int64_t>(static_cast<uint64_t>(v.z) | (static_cast<uint64_t>(v.w) << 32));
    } else {
      gptr<const uint64_t> p = (gptr<const uint64_t>)(src + 16 * static_cast<int64_t>(r));
      lower = p[0];
      upper = static_cast<int64_t>(p[1]);
    }
    OUT o = 0;
    if (row_valid(s_valid, has_nulls, r)) {
      o = static_cast<OUT>(lower);
      const int64_t sext = static_cast<int64_t>(o);
      if (static_cast<uint64_t>(sext) != lower || upper != (sext >> 63)) err = MI_ST_DECIMAL_RANGE;
    }
    if (VARIANT == 2) __builtin_nontemporal_store(o, out + r);
    else out[r] = o;
  }
  raise(status, err);
}

template <int VARIANT>
__global__ __launch_bounds__(kBlockThreads) void transcode_dec128(const mi_col_task* __restrict__ tasks,
                                                                  const uint32_t* __restrict__ tile_begin,
                                                                  const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                  uint32_t total_tiles, uint32_t* __restrict__ status) {
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE_ROWS(kDecTileRows);
    __shared__ uint64_t s_valid[kDecTileRows / 64];
    if (tile_needs_mask(t)) __syncthreads();  // the previous tile's lanes are done with the mask
    tile_validity(t, row0, n, s_valid);
    if (t.param == 8) tile_dec128<int64_t, VARIANT>(t, row0, n, status, s_valid);
    else if (t.param == 4) tile_dec128<int32_t, VARIANT>(t, row0, n, status, s_valid);
    else tile_dec128<int16_t, VARIANT>(t, row0, n, status, s_valid);
  }
}

// ---------------------------------------------------------------------------------------------------- K4
// Builds one string_t from payload bytes [a, a+len) of `data`.  The payload is fetched as aligned dwords and
// realigned with v_alignbyte_b32 (IPC buffers are 8-byte aligned and padded to 8, so the aligned dword that holds
// the last payload byte is always readable).  len <= 12: 12 inline bytes, zero padded.  Else 4-byte prefix + pointer.
__device__ __forceinline__ u32x4 make_string_t(gptr<const uint8_t> data, int64_t a, uint32_t len, uint64_t ptr_base) {
  const uint32_t take = len <= 12 ? len : 4;  // payload bytes that go into the struct
  const uint32_t mis = static_cast<uint32_t>(a & 3);
  gptr<const uint32_t> q = (gptr<const uint32_t>)(data + (a - mis));
  const uint32_t nwords = take ? (mis + take + 3) >> 2 : 0;  // 0..4 aligned dwords cover the payload
  const uint32_t w0 = nwords > 0 ? q[0] : 0;
  const uint32_t w1 = nwords > 1 ? q[1] : 0;
  const uint32_t w2 = nwords > 2 ? q[2] : 0;
  const uint32_t w3 = nwords > 3 ? q[3] : 0;
  const uint32_t o0 = __builtin_amdgcn_alignbyte(w1, w0, mis);
  const uint32_t o1 = __builtin_amdgcn_alignbyte(w2, w1, mis);
  const uint32_t o2 = __builtin_amdgcn_alignbyte(w3, w2, mis);
  u32x4 s;
  s.x = len;
  if (len <= 12) {
    // zero the bytes past len
    const uint32_t k0 = len >= 4 ? 4 : len, k1 = len >= 8 ? 4 : (len > 4 ? len - 4 : 0), k2 = len > 8 ? len - 8 : 0;
    s.y = k0 == 4 ? o0 : (o0 & ((1u << (8 * k0)) - 1u));
    s.z = k1 == 4 ? o1 : (o1 & ((1u << (8 * k1)) - 1u));
    s.w = k2 == 4 ? o2 : (o2 & ((1u << (8 * k2)) - 1u));
  } else {
    const uint64_t p = ptr_base + static_cast<uint64_t>(a);
    s.y = o0;
    s.z = static_cast<uint32_t>(p);
    s.w = static_cast<uint32_t>(p >> 32);
  }
  return s;
}

// utf8 / binary with int32 or int64 offsets.  Lane r reads off[r], off[r+1] (coalesced; the second read hits L1),
// validates them like NANOARROW_VALIDATION_LEVEL_FULL, and stores one 16-byte string_t (1 KiB per wave store).
template <typename OFF>
__device__ __forceinline__ void tile_string(const mi_col_task& t, int64_t row0, int n, uint32_t* status, const uint64_t* s_valid) {
  gptr<const OFF> off = GC<OFF>(t.buf1) + t.row_offset + row0;
  gptr<const uint8_t> data = GC<uint8_t>(t.buf2);
  gptr<u32x4> out = GM<u32x4>(t.out_data) + row0;
  const bool has_nulls = tile_needs_mask(t);
  const int64_t data_len = t.buf2_len;
  uint32_t err = 0;
#pragma unroll 2
  for (int r = threadIdx.x; r < n; r += kBlockThreads) {
    const int64_t a = static_cast<int64_t>(off[r]);
    const int64_t b = static_cast<int64_t>(off[r + 1]);
    u32x4 s = {0u, 0u, 0u, 0u};
    const bool sane = a >= 0 && b >= a && b <= data_len;
    if (!sane) {
      err |= MI_ST_BAD_OFFSETS;
    } else if (sizeof(OFF) == 8 && b > 0xFFFFFFFFll) {
      err |= MI_ST_STRING_TOO_LARGE;  // "DuckDB does not support Strings over 4GB"
    } else if (row_valid(s_valid, has_nulls, r)) {
      s = make_string_t(data, a, static_cast<uint32_t>(b - a), t.ptr_base);
    }
    out[r] = s;
  }
  raise(status, err);
}

// Variant 1: every lane owns the 8 rows {tid + 256k} of the tile and issues all its loads before the first store
// (8 independent offset loads, then up to 8 x 4 payload dwords), so one wave has 8 rows in flight instead of 2.
// off[r+1] comes from the neighbouring lane (one DPP/permute) except at the wave edge and at the last row.
template <typename OFF, int NT>
__device__ __forceinline__ void tile_string_deep(const mi_col_task& t, int64_t row0, int n, uint32_t* status, const uint64_t* s_valid) {
  constexpr int R = kTileRows / kBlockThreads;
  gptr<const OFF> off = GC<OFF>(t.buf1) + t.row_offset + row0;
  gptr<const uint8_t> data = GC<uint8_t>(t.buf2);
  gptr<u32x4> out = GM<u32x4>(t.out_data) + row0;
  const bool has_nulls = tile_needs_mask(t);
  const int64_t data_len = t.buf2_len;
  const int lane = threadIdx.x & 63;
  OFF a[R], b[R];
#pragma unroll
  for (int k = 0; k < R; k++) {
    const int r = threadIdx.x + k * kBlockThreads;
    a[k] = r < n ? (NT >= 2 ? __builtin_nontemporal_load(off + r) : off[r]) : 0;
  }
#pragma unroll
  for (int k = 0; k < R; k++) {
    const int r = threadIdx.x + k * kBlockThreads;
    const OFF from_neighbour = __shfl_down(a[k], 1, 64);
    const bool edge = lane == 63 || r == n - 1;
    b[k] = (r < n && edge) ? off[r + 1] : from_neighbour;
  }
  uint32_t err = 0;
  u32x4 s[R];
#pragma unroll
  for (int k = 0; k < R; k++) {
    const int r = threadIdx.x + k * kBlockThreads;
    s[k] = u32x4{0u, 0u, 0u, 0u};
    if (r < n) {
      const int64_t aa = static_cast<int64_t>(a[k]), bb = static_cast<int64_t>(b[k]);
      const bool sane = aa >= 0 && bb >= aa && bb <= data_len;
      if (!sane) {
        err |= MI_ST_BAD_OFFSETS;
      } else if (sizeof(OFF) == 8 && bb > 0xFFFFFFFFll) {
        err |= MI_ST_STRING_TOO_LARGE;
      } else if (row_valid(s_valid, has_nulls, r)) {
        s[k] = make_string_t(data, aa, static_cast<uint32_t>(bb - aa), t.ptr_base);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < R; k++) {
    const int r = threadIdx.x + k * kBlockThreads;
    if (r < n) {
      if (NT >= 1) __builtin_nontemporal_store(s[k], out + r);
      else out[r] = s[k];
    }
  }
  raise(status, err);
}

// fixed_size_binary(width) -> string_t
__device__ __forceinline__ void tile_fixed_binary(const mi_col_task& t, int64_t row0, int n, const uint64_t* s_valid) {
  gptr<const uint8_t> data = GC<uint8_t>(t.buf1);
  gptr<u32x4> out = GM<u32x4>(t.out_data) + row0;
  const bool has_nulls = tile_needs_mask(t);
  const int64_t width = t.param;
#pragma unroll 2
  for (int r = threadIdx.x; r < n; r += kBlockThreads) {
    u32x4 s = {0u, 0u, 0u, 0u};
    if (row_valid(s_valid, has_nulls, r))
      s = make_string_t(data, (t.row_offset + row0 + r) * width, static_cast<uint32_t>(width), t.ptr_base);
    out[r] = s;
  }
}

template <int VARIANT>
__global__ __launch_bounds__(kBlockThreads) void transcode_string(const mi_col_task* __restrict__ tasks,
                                                                  const uint32_t* __restrict__ tile_begin,
                                                                  const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                  uint32_t total_tiles, uint32_t* __restrict__ status) {
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE();
    __shared__ uint64_t s_valid[kTileRows / 64];
    if (tile_needs_mask(t)) __syncthreads();
    tile_validity(t, row0, n, s_valid);
    if (VARIANT >= 1) {
      if (t.kind == MI_K_STR32) tile_string_deep<int32_t, VARIANT - 1>(t, row0, n, status, s_valid);
      else if (t.kind == MI_K_STR64) tile_string_deep<int64_t, VARIANT - 1>(t, row0, n, status, s_valid);
      else tile_fixed_binary(t, row0, n, s_valid);
    } else {
      if (t.kind == MI_K_STR32) tile_string<int32_t>(t, row0, n, status, s_valid);
      else if (t.kind == MI_K_STR64) tile_string<int64_t>(t, row0, n, status, s_valid);
      else tile_fixed_binary(t, row0, n, s_valid);
    }
  }
}

// ---------------------------------------------------------------------------------------------------- K2
// Bit-packed bool -> one byte per row, all rows (valid or not).  Lane i expands rows [8i, 8i+8) = one source byte
// (two when the bit offset is not byte aligned) into one 8-byte store.
template <int T = kBlockThreads>
__device__ __forceinline__ void tile_bool(const mi_col_task& t, int64_t row0, int n) {
  gptr<const uint8_t> bits = GC<uint8_t>(t.buf1);
  gptr<uint8_t> out = GM<uint8_t>(t.out_data) + row0;
  const int64_t last_byte = (t.row_offset + t.nrows - 1) >> 3;
  for (int r = 8 * threadIdx.x; r < n; r += 8 * T) {  // 8 rows per lane and pass: one byte (two when unaligned) -> 8 bytes
    const int64_t bit = t.row_offset + row0 + r;
    const int64_t byte = bit >> 3;
    const int sh = static_cast<int>(bit & 7);
    uint32_t b = bits[byte];
    if (sh != 0 && byte + 1 <= last_byte) b |= static_cast<uint32_t>(bits[byte + 1]) << 8;
    b = (b >> sh) & 0xFFu;
    uint64_t y = (static_cast<uint64_t>(b) * 0x0101010101010101ull) & 0x8040201008040201ull;
    y = ((y + 0x7F7F7F7F7F7F7F7Full) >> 7) & 0x0101010101010101ull;
    if (r + 8 <= n) {
      *(gptr<uint64_t>)(out + r) = y;
    } else {
      for (int k = 0; r + k < n; k++) out[r + k] = static_cast<uint8_t>(y >> (8 * k));
    }
  }
}

// ---------------------------------------------------------------------------------------------------- K3c
template <int T = kBlockThreads>
__device__ __forceinline__ void tile_date64(const mi_col_task& t, int64_t row0, int n) {
  gptr<const int64_t> src = GC<int64_t>(t.buf1) + t.row_offset + row0;
  gptr<int32_t> out = GM<int32_t>(t.out_data) + row0;
#pragma unroll 2  // 64-bit division by a constant is register hungry; 4 copies cost the class one occupancy step
  for (int r = threadIdx.x; r < n; r += T) out[r] = static_cast<int32_t>(src[r] / 86400000ll);
}

template <int T = kBlockThreads>
__device__ __forceinline__ void tile_mul_i32(const mi_col_task& t, int64_t row0, int n, const uint64_t* s_valid) {
  gptr<const int32_t> src = GC<int32_t>(t.buf1) + t.row_offset + row0;
  gptr<int64_t> out = GM<int64_t>(t.out_data) + row0;
  const bool has_nulls = tile_needs_mask(t);
#pragma unroll 4
  for (int r = threadIdx.x; r < n; r += T) {
    const bool ok = row_valid(s_valid, has_nulls, r);
    out[r] = ok ? static_cast<int64_t>(src[r]) * t.param : 0;  // int32 * 1e6 cannot overflow int64
  }
}

template <int T = kBlockThreads>
__device__ __forceinline__ void tile_mul_i64(const mi_col_task& t, int64_t row0, int n, uint32_t* status, const uint64_t* s_valid) {
  gptr<const int64_t> src = GC<int64_t>(t.buf1) + t.row_offset + row0;
  gptr<int64_t> out = GM<int64_t>(t.out_data) + row0;
  const bool has_nulls = tile_needs_mask(t);
  uint32_t err = 0;
#pragma unroll 4
  for (int r = threadIdx.x; r < n; r += T) {
    int64_t v = 0;
    if (row_valid(s_valid, has_nulls, r)) {
      if (__builtin_mul_overflow(src[r], t.param, &v)) {  // TryMultiplyOperator => ConversionException
        v = 0;
        err = MI_ST_MUL_OVERFLOW;
      }
    }
    out[r] = v;
  }
  raise(status, err);
}

template <int T = kBlockThreads>
__device__ __forceinline__ void tile_div_i64(const mi_col_task& t, int64_t row0, int n) {
  gptr<const int64_t> src = GC<int64_t>(t.buf1) + t.row_offset + row0;
  gptr<int64_t> out = GM<int64_t>(t.out_data) + row0;
  const int64_t d = t.param;
  // the divisors the type mapping produces are 1000 (ns -> us) and powers of it: constant divisions are a multiply-high,
  // the generic 64-bit division (~100 instructions, dozens of registers) stays out of the unrolled loops
  if (d == 1000) {
#pragma unroll 4
    for (int r = threadIdx.x; r < n; r += T) out[r] = src[r] / 1000;  // all rows, like upstream
    return;
  }
#pragma unroll 1
  for (int r = threadIdx.x; r < n; r += T) out[r] = src[r] / d;
}

__device__ __forceinline__ void tile_duration(const mi_col_task& t, int64_t row0, int n, uint32_t* status, const uint64_t* s_valid) {
  gptr<const int64_t> src = GC<int64_t>(t.buf1) + t.row_offset + row0;
  gptr<u32x4> out = GM<u32x4>(t.out_data) + row0;
  const bool has_nulls = tile_needs_mask(t);
  uint32_t err = 0;
#pragma unroll 1
  for (int r = threadIdx.x; r < n; r += kBlockThreads) {
    int64_t micros = 0;
    if (t.param < 0) {
      micros = t.param == -1000 ? src[r] / 1000 : src[r] / (-t.param);
    } else if (row_valid(s_valid, has_nulls, r)) {
      if (__builtin_mul_overflow(src[r], t.param, &micros)) {
        micros = 0;
        err = MI_ST_MUL_OVERFLOW;
      }
    }
    u32x4 o;
    o.x = 0;  // months
    o.y = 0;  // days
    o.z = static_cast<uint32_t>(static_cast<uint64_t>(micros));
    o.w = static_cast<uint32_t>(static_cast<uint64_t>(micros) >> 32);
    out[r] = o;
  }
  raise(status, err);
}

__device__ __forceinline__ void tile_interval_months(const mi_col_task& t, int64_t row0, int n) {
  gptr<const int32_t> src = GC<int32_t>(t.buf1) + t.row_offset + row0;
  gptr<u32x4> out = GM<u32x4>(t.out_data) + row0;
#pragma unroll 4
  for (int r = threadIdx.x; r < n; r += kBlockThreads) out[r] = u32x4{static_cast<uint32_t>(src[r]), 0u, 0u, 0u};
}

__device__ __forceinline__ void tile_interval_mdn(const mi_col_task& t, int64_t row0, int n) {
  gptr<const uint8_t> src = GC<uint8_t>(t.buf1) + (t.row_offset + row0) * 16;
  gptr<u32x4> out = GM<u32x4>(t.out_data) + row0;
#pragma unroll 4
  for (int r = threadIdx.x; r < n; r += kBlockThreads) {
    const u32x4 v = *(gptr<const u32x4_a4>)(src + 16 * static_cast<int64_t>(r));
    const int64_t nanos = static_cast<int64_t>(static_cast<uint64_t>(v.z) | (static_cast<uint64_t>(v.w) << 32));
    const uint64_t micros = static_cast<uint64_t>(nanos / 1000);
    out[r] = u32x4{v.x, v.y, static_cast<uint32_t>(micros), static_cast<uint32_t>(micros >> 32)};
  }
}

// decimal32 / decimal64 -> the physical type of the declared precision, valid rows only (NULL -> 0)
template <typename SRC, typename DST>
__device__ __forceinline__ void tile_narrow(const mi_col_task& t, int64_t row0, int n, uint32_t* status, const uint64_t* s_valid) {
  gptr<const SRC> src = GC<SRC>(t.buf1) + t.row_offset + row0;
  gptr<DST> out = GM<DST>(t.out_data) + row0;
  const bool has_nulls = tile_needs_mask(t);
  uint32_t err = 0;
#pragma unroll 4
  for (int r = threadIdx.x; r < n; r += kBlockThreads) {
    DST o = 0;
    if (row_valid(s_valid, has_nulls, r)) {
      const SRC v = src[r];
      o = static_cast<DST>(v);
      if (static_cast<SRC>(o) != v) err = MI_ST_DECIMAL_RANGE;
    }
    out[r] = o;
  }
  raise(status, err);
}

// IEEE binary16 -> binary32, exact (subnormals normalised, inf / nan keep their payload)
__device__ __forceinline__ void tile_half_float(const mi_col_task& t, int64_t row0, int n) {
  gptr<const uint16_t> src = GC<uint16_t>(t.buf1) + t.row_offset + row0;
  gptr<uint32_t> out = GM<uint32_t>(t.out_data) + row0;
#pragma unroll 4
  for (int r = threadIdx.x; r < n; r += kBlockThreads) {
    const uint32_t h = src[r];
    const uint32_t sign = (h & 0x8000u) << 16, exp = (h >> 10) & 0x1F;
    uint32_t man = h & 0x3FF, f;
    if (exp == 0) {
      if (man == 0) {
        f = sign;
      } else {
        const int lz = __builtin_clz(man) - 21;  // shifts needed to bring the leading 1 to bit 10
        man = (man << lz) & 0x3FF;
        f = sign | (static_cast<uint32_t>(113 - lz) << 23) | (man << 13);
      }
    } else if (exp == 31) {
      f = sign | 0x7F800000u | (man << 13);
    } else {
      f = sign | ((exp + 112) << 23) | (man << 13);
    }
    out[r] = f;
  }
}

// arrow null type: every row NULL
__device__ __forceinline__ void tile_null(const mi_col_task& t, int64_t row0, int n) {
  gptr<uint8_t> out = GM<uint8_t>(t.out_data) + row0;
  for (int r = threadIdx.x; r < n; r += kBlockThreads) out[r] = 0;
  if (t.out_validity != nullptr) {
    const int nwords = (n + 63) >> 6;
    for (int w = threadIdx.x; w < nwords; w += kBlockThreads) GM<uint64_t>(t.out_validity)[(row0 >> 6) + w] = 0ull;
  }
}

// ---------------------------------------------------------------------------------------------------- nested
// list / large_list / map offsets -> list_entry_t{u64 offset, u64 length} (ConvertArrowListOffsets): the offset is
// relative to the first element of the row's top-level 2048-row window, because the child vector a chunk carries starts
// there.  For a top-level list the window is this tile; for a list nested inside lists buf2 holds the window starts in
// this list's own row space (computed on the host from the outer offsets).  Offsets are validated like FULL.
template <typename OFF>
__device__ __forceinline__ void tile_list(const mi_col_task& t, int64_t row0, int n, uint32_t* status) {
  gptr<const OFF> off = GC<OFF>(t.buf1) + t.row_offset;
  gptr<u32x4> out = GM<u32x4>(t.out_data) + row0;
  gptr<const int64_t> wins = GC<int64_t>(t.buf2);
  const int nwin = static_cast<int>(t.buf2_len);
  const int64_t child_len = t.param;
  uint32_t err = 0;
#pragma unroll 2
  for (int r = threadIdx.x; r < n; r += kBlockThreads) {
    const int64_t row = row0 + r;
    int64_t win_row = row0;  // top-level list: the tile is the window
    if (t.buf2 != nullptr) {
      int lo = 0, hi = nwin;  // largest k with wins[k] <= row
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (wins[mid] <= row) lo = mid; else hi = mid;
      }
      win_row = wins[lo];
    }
    const int64_t a = static_cast<int64_t>(off[row]), b = static_cast<int64_t>(off[row + 1]);
    const int64_t base = static_cast<int64_t>(off[win_row]);
    if (a < 0 || b < a || b > child_len || a < base) err = MI_ST_BAD_OFFSETS;
    const uint64_t o = static_cast<uint64_t>(a - base), l = static_cast<uint64_t>(b - a);
    out[r] = u32x4{static_cast<uint32_t>(o), static_cast<uint32_t>(o >> 32), static_cast<uint32_t>(l), static_cast<uint32_t>(l >> 32)};
  }
  raise(status, err);
}

// utf8_view / binary_view -> string_t.  Inline views (len <= 12) are already string_t shaped (the pad bytes are
// re-zeroed like the string_t constructor does); long views {len, prefix, buffer_index, offset} get the pointer
// bases[buffer_index] + offset from the per-task table of variadic data buffers.
__device__ __forceinline__ void tile_strview(const mi_col_task& t, int64_t row0, int n, uint32_t* status, const uint64_t* s_valid) {
  gptr<const u32x4_a4> src = (gptr<const u32x4_a4>)(GC<uint8_t>(t.buf1) + (t.row_offset + row0) * 16);
  gptr<u32x4> out = GM<u32x4>(t.out_data) + row0;
  gptr<const uint64_t> table = GC<uint64_t>(t.buf2);  // {address, length} pairs
  const int64_t nbuf = t.buf2_len;
  const bool has_nulls = tile_needs_mask(t);
  uint32_t err = 0;
#pragma unroll 2
  for (int r = threadIdx.x; r < n; r += kBlockThreads) {
    u32x4 s = {0u, 0u, 0u, 0u};
    if (row_valid(s_valid, has_nulls, r)) {
      const u32x4 v = src[r];
      const uint32_t len = v.x;
      if (len <= 12) {
        const uint32_t k0 = len >= 4 ? 4 : len, k1 = len >= 8 ? 4 : (len > 4 ? len - 4 : 0), k2 = len > 8 ? len - 8 : 0;
        s.x = len;
        s.y = k0 == 4 ? v.y : (v.y & ((1u << (8 * k0)) - 1u));
        s.z = k1 == 4 ? v.z : (v.z & ((1u << (8 * k1)) - 1u));
        s.w = k2 == 4 ? v.w : (v.w & ((1u << (8 * k2)) - 1u));
      } else {
        const int64_t bi = static_cast<int32_t>(v.z), bo = static_cast<int32_t>(v.w);
        if (bi < 0 || bi >= nbuf || bo < 0 || static_cast<uint64_t>(bo) + len > table[2 * bi + 1]) {
          err = MI_ST_BAD_OFFSETS;
        } else {
          const uint64_t p = table[2 * bi] + static_cast<uint64_t>(bo);
          s = u32x4{len, v.y, static_cast<uint32_t>(p), static_cast<uint32_t>(p >> 32)};
        }
      }
    }
    out[r] = s;
  }
  raise(status, err);
}

// ---------------------------------------------------------------------------------------------------- K5
// Dictionary indices -> sel_t; NULL -> dict_len (the extra NULL slot of the decoded dictionary).
template <int T = kBlockThreads>
__device__ __forceinline__ void tile_dict(const mi_col_task& t, int64_t row0, int n, uint32_t* status, const uint64_t* s_valid) {
  const int iw = static_cast<int>(t.param & 0xFF);
  const bool is_signed = ((t.param >> 8) & 1) != 0;
  gptr<const uint8_t> idx = GC<uint8_t>(t.buf1) + (t.row_offset + row0) * iw;
  gptr<uint32_t> out = GM<uint32_t>(t.out_data) + row0;
  const bool has_nulls = tile_needs_mask(t);
  const uint32_t dict_len = static_cast<uint32_t>(t.param2);
  uint32_t err = 0;
#pragma unroll 4
  for (int r = threadIdx.x; r < n; r += T) {
    uint32_t sel = dict_len;
    if (row_valid(s_valid, has_nulls, r)) {
      uint64_t v;
      switch (iw) {
        case 1: v = is_signed ? static_cast<uint64_t>(static_cast<int64_t>(((gptr<const int8_t>)idx)[r])) : idx[r]; break;
        case 2: v = is_signed ? static_cast<uint64_t>(static_cast<int64_t>(((gptr<const int16_t>)idx)[r]))
                              : ((gptr<const uint16_t>)idx)[r]; break;
        case 4: v = is_signed ? static_cast<uint64_t>(static_cast<int64_t>(((gptr<const int32_t>)idx)[r]))
                              : ((gptr<const uint32_t>)idx)[r]; break;
        default: v = ((gptr<const uint64_t>)idx)[r]; break;
      }
      if (v > 0xFFFFFFFFull) {  // "DuckDB only supports indices that fit on an uint32"
        err = MI_ST_INDEX_RANGE;
        v = dict_len;
      } else if (v >= dict_len) {  // the selection vector must never point past the dictionary's NULL slot
        err = MI_ST_DICT_INDEX;
        v = dict_len;
      }
      sel = static_cast<uint32_t>(v);
    }
    out[r] = sel;
  }
  raise(status, err);
}

// The common flat kinds (GROUP 0 of transcode_misc) move 2-16 KB per tile: with 256-thread workgroups a CU holds 8 such
// tiles and each is one chain of dependent round trips (task lookup, bitmap, data), which left the kernel latency bound
// (1.8-2.3 TB/s).  Here ONE WAVE owns a tile (32 rows per lane): 32 independent tiles per CU, no workgroup barriers.
constexpr int kLightThreads = 64;
__global__ __launch_bounds__(kLightThreads) void transcode_misc_light(const mi_col_task* __restrict__ tasks,
                                                                      const uint32_t* __restrict__ tile_begin,
                                                                      const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                      uint32_t total_tiles, uint32_t* __restrict__ status) {
  __shared__ uint64_t s_valid[kTileRows / 64];
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE();
    const bool mine = t.kind == MI_K_BOOL || t.kind == MI_K_DICT || t.kind == MI_K_DATE64 || t.kind == MI_K_MUL_I32 ||
                      t.kind == MI_K_MUL_I64 || t.kind == MI_K_DIV_I64;
    if (!mine) continue;  // uniform: another group's launch owns this tile
    if (tile_needs_mask(t)) __syncthreads();
    tile_validity<kLightThreads>(t, row0, n, s_valid);
    switch (t.kind) {
      case MI_K_BOOL: tile_bool<kLightThreads>(t, row0, n); break;
      case MI_K_DATE64: tile_date64<kLightThreads>(t, row0, n); break;
      case MI_K_MUL_I32: tile_mul_i32<kLightThreads>(t, row0, n, s_valid); break;
      case MI_K_MUL_I64: tile_mul_i64<kLightThreads>(t, row0, n, status, s_valid); break;
      case MI_K_DIV_I64: tile_div_i64<kLightThreads>(t, row0, n); break;
      case MI_K_DICT: tile_dict<kLightThreads>(t, row0, n, status, s_valid); break;
      default: break;
    }
  }
}

// GROUP 0: the common flat kinds (bool, dictionary indices, date64, timestamp unit casts) -- small per-row work whose
// throughput is set by how many tiles a CU keeps in flight, so they get their own register budget; GROUP 1: list
// entries, string views, struct validity; GROUP 2: the rare flat kinds (intervals, durations, decimal32/64, half
// floats, null).  A plan launches only the groups its tasks use.
template <int GROUP>
__global__ __launch_bounds__(kBlockThreads) void transcode_misc(const mi_col_task* __restrict__ tasks,
                                                                const uint32_t* __restrict__ tile_begin,
                                                                const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                uint32_t total_tiles, uint32_t* __restrict__ status) {
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE();
    __shared__ uint64_t s_valid[kTileRows / 64];
    const int group = (t.kind == MI_K_STRUCT || t.kind == MI_K_LIST32 || t.kind == MI_K_LIST64 || t.kind == MI_K_STRVIEW) ? 1
                      : (t.kind == MI_K_BOOL || t.kind == MI_K_DICT || t.kind == MI_K_DATE64 || t.kind == MI_K_MUL_I32 ||
                         t.kind == MI_K_MUL_I64 || t.kind == MI_K_DIV_I64) ? 0 : 2;
    if (group != GROUP) continue;  // uniform: another group's launch owns this tile
    if (tile_needs_mask(t)) __syncthreads();
    if (t.kind != MI_K_NULL) tile_validity(t, row0, n, s_valid);
    if (GROUP == 1) {
      switch (t.kind) {
        case MI_K_LIST32: tile_list<int32_t>(t, row0, n, status); break;
        case MI_K_LIST64: tile_list<int64_t>(t, row0, n, status); break;
        case MI_K_STRVIEW: tile_strview(t, row0, n, status, s_valid); break;
        default: break;  // MI_K_STRUCT: validity only
      }
    } else if (GROUP == 0) {
      switch (t.kind) {
        case MI_K_BOOL: tile_bool(t, row0, n); break;
        case MI_K_DATE64: tile_date64(t, row0, n); break;
        case MI_K_MUL_I32: tile_mul_i32(t, row0, n, s_valid); break;
        case MI_K_MUL_I64: tile_mul_i64(t, row0, n, status, s_valid); break;
        case MI_K_DIV_I64: tile_div_i64(t, row0, n); break;
        case MI_K_DICT: tile_dict(t, row0, n, status, s_valid); break;
        default: break;
      }
    } else {
      switch (t.kind) {
        case MI_K_NULL: tile_null(t, row0, n); break;
        case MI_K_INTERVAL_MONTHS: tile_interval_months(t, row0, n); break;
        case MI_K_INTERVAL_MDN: tile_interval_mdn(t, row0, n); break;
        case MI_K_HALF_FLOAT: tile_half_float(t, row0, n); break;
        case MI_K_NARROW: {
          const int sw = static_cast<int>(t.param & 0xFF), dw = static_cast<int>((t.param >> 8) & 0xFF);
          if (sw == 4) tile_narrow<int32_t, int16_t>(t, row0, n, status, s_valid);
          else if (dw == 2) tile_narrow<int64_t, int16_t>(t, row0, n, status, s_valid);
          else tile_narrow<int64_t, int32_t>(t, row0, n, status, s_valid);
          break;
        }
        case MI_K_DURATION: tile_duration(t, row0, n, status, s_valid); break;
        default: break;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------- K6
// Range filter lo <= v < hi AND valid -> ascending window-relative indices, one selection vector per 2048-row window.
// A workgroup takes kFilterWindows consecutive windows: lane i owns rows [8i, 8i+8) of each (16 to 64 contiguous bytes,
// loaded as one vector), all windows' loads are issued before anything depends on them (a single 8 KB window per
// workgroup left the kernel bound by the latency of that one round trip: 2.1 TB/s), then per window a DPP wave scan +
// one LDS exchange of the 4 wave totals gives every lane its output position: the selection vector comes out sorted
// without a second pass.
constexpr int kFilterWindows = 4;
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v);

template <typename T>
__global__ __launch_bounds__(kBlockThreads) void filter_range(const T* __restrict__ values_p,
                                                              const uint64_t* __restrict__ validity_p, int64_t nrows,
                                                              int64_t lo, int64_t hi, mi_sel_t* __restrict__ sel_out_p,
                                                              uint32_t* __restrict__ count_out_p) {
  typedef T vec8 __attribute__((ext_vector_type(8)));
  typedef vec8 vec8_a4 __attribute__((aligned(4)));
  __shared__ uint32_t wave_total[kFilterWindows][kBlockThreads / 64];
  gptr<const T> values = GC<T>(values_p);
  gptr<const uint64_t> validity = GC<uint64_t>(validity_p);
  gptr<mi_sel_t> sel_out = GM<mi_sel_t>(sel_out_p);
  gptr<uint32_t> count_out = GM<uint32_t>(count_out_p);
  const int64_t first_window = static_cast<int64_t>(blockIdx.x) * kFilterWindows;
  const int r = 8 * threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t mask[kFilterWindows];
#pragma unroll
  for (int w = 0; w < kFilterWindows; w++) {
    const int64_t row0 = (first_window + w) * kTileRows;
    const int64_t left = nrows - row0;
    const int n = left < kTileRows ? static_cast<int>(left < 0 ? 0 : left) : kTileRows;
    uint32_t m = 0;
    if (r + 8 <= n) {
      const uint32_t vbits = validity ? static_cast<uint32_t>((validity[(row0 + r) >> 6] >> ((row0 + r) & 63)) & 0xFF) : 0xFFu;
      const vec8 v = __builtin_nontemporal_load((gptr<const vec8_a4>)(values + row0 + r));
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int64_t x = static_cast<int64_t>(v[k]);
        if (x >= lo && x < hi) m |= 1u << k;
      }
      m &= vbits;
    } else if (r < n) {  // the table's last, partial vector
      const uint32_t vbits = validity ? static_cast<uint32_t>((validity[(row0 + r) >> 6] >> ((row0 + r) & 63)) & 0xFF) : 0xFFu;
      for (int k = 0; r + k < n; k++) {
        const int64_t x = static_cast<int64_t>(values[row0 + r + k]);
        if (x >= lo && x < hi) m |= 1u << k;
      }
      m &= vbits;
    }
    mask[w] = m;
  }
  uint32_t incl[kFilterWindows];
#pragma unroll
  for (int w = 0; w < kFilterWindows; w++) {
    incl[w] = wave_inclusive_scan_u32(__builtin_popcount(mask[w]));
    if (lane == 63) wave_total[w][wave] = incl[w];
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < kFilterWindows; w++) {
    const int64_t window = first_window + w;
    if (window * kTileRows >= nrows) break;  // uniform
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < kBlockThreads / 64; i++) {
      const uint32_t x = wave_total[w][i];
      if (i < wave) base += x;
      total += x;
    }
    uint32_t pos = base + incl[w] - __builtin_popcount(mask[w]);
    gptr<mi_sel_t> out = sel_out + window * kTileRows;
    uint32_t m = mask[w];
    while (m) {  // ascending set bits
      const int k = __builtin_ctz(m);
      m &= m - 1;
      out[pos++] = static_cast<mi_sel_t>(r + k);
    }
    if (threadIdx.x == 0) count_out[window] = total;
  }
}

// ==================================================================================================== K7 (encode)
// DuckDB vectors -> Arrow buffers, ArrowAppender semantics.  Task fields for encode kinds:
//   validity  = DuckDB validity words of the whole column (NULL = all valid)   buf1 = vector data
//   buf2      = string heap base (long string_t pointers are ptr - ptr_base into it)
//   out_validity = Arrow bitmap (ceil(n/8) bytes, always emitted, pad bits 1)  out_data = Arrow buffer 1
//   out_aux   = Arrow buffer 2 (string data)          param2 = index of this task's null counter

// K7a: DuckDB validity words have Arrow's bit order and polarity, so the bitmap is a byte copy of the words with
// the pad bits of the last byte forced to 1 (ResizeValidity fills with 0xFF) and NULLs counted on the way.
__device__ __forceinline__ void enc_tile_validity(const mi_col_task& t, int64_t row0, int n, int64_t* null_counts,
                                                  uint64_t* s_valid = nullptr) {
  const int lane = threadIdx.x;
  const int nwords = (n + 63) >> 6;
  if (lane >= nwords || (t.out_validity == nullptr && s_valid == nullptr)) return;
  uint64_t w = ~0ull;
  if (t.validity != nullptr) w = GC<uint64_t>(t.validity)[(row0 >> 6) + lane];
  const int rem = n - 64 * lane;
  if (rem < 64) w |= ~0ull << rem;
  if (s_valid) s_valid[lane] = w;
  if (t.out_validity == nullptr) return;
  const int nulls = 64 - __builtin_popcountll(w);
  if (nulls) atomicAdd(reinterpret_cast<unsigned long long*>(null_counts + t.param2), static_cast<unsigned long long>(nulls));
  gptr<uint8_t> out = GM<uint8_t>(t.out_validity) + (row0 >> 3) + 8 * lane;
  const int nbytes = rem >= 64 ? 8 : (rem + 7) >> 3;
  if (nbytes == 8 && (reinterpret_cast<uintptr_t>(out) & 7) == 0) {
    *(gptr<uint64_t>)out = w;
  } else {
    for (int k = 0; k < nbytes; k++) out[k] = static_cast<uint8_t>(w >> (8 * k));
  }
}

__device__ __forceinline__ bool enc_row_valid(gptr<const uint64_t> v, bool has, int64_t row) {
  return !has || ((v[row >> 6] >> (row & 63)) & 1);
}

// K7b: DECIMAL physical int16/32/64 -> decimal128 by sign extension, one 16-byte store per row
template <typename IN>
__device__ __forceinline__ void enc_tile_dec128(const mi_col_task& t, int64_t row0, int n) {
  gptr<const IN> src = GC<IN>(t.buf1) + row0;
  gptr<u32x4> out = GM<u32x4>(t.out_data) + row0;
#pragma unroll 4
  for (int r = threadIdx.x; r < n; r += kBlockThreads) {
    const int64_t v = static_cast<int64_t>(__builtin_nontemporal_load(src + r));
    const uint32_t sign = static_cast<uint32_t>(v >> 63);
    u32x4 o;
    o.x = static_cast<uint32_t>(static_cast<uint64_t>(v));
    o.y = static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32);
    o.z = sign;
    o.w = sign;
    __builtin_nontemporal_store(o, out + r);
  }
}

// K7c: byte bool -> bit; data bits start as 1, a valid false clears its bit, NULL rows keep 1
__device__ __forceinline__ void enc_tile_bool(const mi_col_task& t, int64_t row0, int n) {
  gptr<const uint8_t> src = GC<uint8_t>(t.buf1) + row0;
  gptr<const uint64_t> valid = GC<uint64_t>(t.validity);
  const bool has = t.validity != nullptr;
  gptr<uint8_t> out = GM<uint8_t>(t.out_data) + (row0 >> 3);
  const int r = 8 * threadIdx.x;
  if (r >= n) return;
  uint32_t b = 0xFF;
  for (int k = 0; k < 8 && r + k < n; k++) {
    if (enc_row_valid(valid, has, row0 + r + k) && src[r + k] == 0) b &= ~(1u << k);
  }
  out[threadIdx.x] = static_cast<uint8_t>(b);
}

__global__ __launch_bounds__(kBlockThreads) void encode_fixed(const mi_col_task* __restrict__ tasks,
                                                              const uint32_t* __restrict__ tile_begin, const uint32_t* __restrict__ tile_task, int n_tasks,
                                                              uint32_t total_tiles, int64_t* __restrict__ null_counts) {
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE();
    enc_tile_validity(t, row0, n, null_counts);
    switch (t.kind) {
      case MI_K_ENC_COPY: {  // NULL slots copy whatever the source slot holds, like ArrowScalarData::Append
        const int w = static_cast<int>(t.param);
        copy_bytes(GC<uint8_t>(t.buf1) + row0 * w, GM<uint8_t>(t.out_data) + row0 * w, n * w, 2);
        break;
      }
      case MI_K_ENC_DEC128:
        if (t.param == 8) enc_tile_dec128<int64_t>(t, row0, n);
        else if (t.param == 4) enc_tile_dec128<int32_t>(t, row0, n);
        else enc_tile_dec128<int16_t>(t, row0, n);
        break;
      case MI_K_ENC_BOOL: enc_tile_bool(t, row0, n); break;
      default: break;
    }
  }
}

// block-wide exclusive scan of one value per thread (4 waves); returns the exclusive prefix, *total = block sum
__device__ __forceinline__ int64_t block_exclusive_scan(int64_t v, int64_t* total, int64_t* lds4) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int64_t up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (lane == 63) lds4[wave] = incl;
  __syncthreads();
  int64_t base = 0, sum = 0;
  for (int w = 0; w < kBlockThreads / 64; w++) {
    if (w < wave) base += lds4[w];
    sum += lds4[w];
  }
  *total = sum;
  __syncthreads();
  return base + incl - v;
}

// K7d pass 1: payload bytes per tile (valid rows only) -> tile_sums[tile]
__global__ __launch_bounds__(kBlockThreads) void encode_string_tile_sums(const mi_col_task* __restrict__ tasks,
                                                                         const uint32_t* __restrict__ tile_begin, const uint32_t* __restrict__ tile_task,
                                                                         int n_tasks, uint32_t total_tiles,
                                                                         int64_t* __restrict__ tile_sums) {
  __shared__ int64_t lds4[kBlockThreads / 64];
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE();
    // string_t.length (dword 0) or, for MI_K_ENC_LIST32, list_entry_t.length (low dword of the second u64) every 16 B
    gptr<const uint32_t> lens = GC<uint32_t>(t.buf1) + 4 * row0 + (t.kind == MI_K_ENC_LIST32 ? 2 : 0);
    gptr<const uint64_t> valid = GC<uint64_t>(t.validity);
    const bool has = t.validity != nullptr;
    int64_t local = 0;
    for (int r = threadIdx.x; r < n; r += kBlockThreads)
      if (enc_row_valid(valid, has, row0 + r)) local += lens[4 * r];
    int64_t total;
    block_exclusive_scan(local, &total, lds4);
    if (threadIdx.x == 0) tile_sums[tile] = total;
  }
}

// K7d pass 2: per task, exclusive scan of its tiles' sums (in place) + INT32_MAX overflow check.  One workgroup
// per task; a 122880-row batch has 60 tiles, so this is a handful of waves.
__global__ __launch_bounds__(kBlockThreads) void encode_string_scan(const mi_col_task* __restrict__ tasks,
                                                                    const uint32_t* __restrict__ tile_begin,
                                                                    int n_tasks, int64_t* __restrict__ tile_sums,
                                                                    uint32_t* __restrict__ status) {
  __shared__ int64_t lds4[kBlockThreads / 64];
  for (int ti = blockIdx.x; ti < n_tasks; ti += gridDim.x) {
    const uint32_t first = tile_begin[ti], last = tile_begin[ti + 1];
    int64_t carry = 0;
    for (uint32_t base = first; base < last; base += kBlockThreads) {
      const uint32_t i = base + threadIdx.x;
      const int64_t v = i < last ? tile_sums[i] : 0;
      int64_t total;
      const int64_t ex = block_exclusive_scan(v, &total, lds4);
      if (i < last) tile_sums[i] = carry + ex;
      carry += total;
    }
    if (threadIdx.x == 0 && carry > 0x7FFFFFFFll && !(tasks[ti].flags & 1)) atomicOr(status, MI_ST_OFFSET_OVERFLOW);
  }
}

// K7d pass 3: offsets + payload.  The tile is processed as 8 sub-blocks of 256 rows (lane r = row, so the 16-byte
// string_t loads and the 4-byte offset stores are coalesced); per sub-block a wave scan + 4-wave LDS combine gives every
// row its output position, the payload bytes of the sub-block are assembled in LDS (inline bytes come from the string_t
// registers, long strings from the heap behind the pointer) and leave as coalesced 16-byte stores.  A sub-block whose
// payload exceeds the LDS stage falls back to direct byte stores.
constexpr int kEncStage = 16 * 1024;  // bytes of payload staged per 256-row sub-block (16 KB x 8 workgroups per CU)

// One tile of K7d pass 3 with 64-bit positions: sub-block by sub-block (256 rows), byte-wise LDS assembly, a sub-block
// whose payload exceeds the stage falls back to direct byte stores.  The original formulation; now the fallback of
// encode_string_v5 for tiles that hold a string of >= 8 MiB, and variant 0 of the A/B knob.
__device__ __forceinline__ void encode_string_tile_generic(const mi_col_task& t, int64_t row0, int n, int64_t base,
                                                           int64_t* lds4, uint8_t* stage) {
  gptr<const u32x4> str = GC<u32x4>(t.buf1) + row0;
  gptr<const uint64_t> valid = GC<uint64_t>(t.validity);
  const bool has = t.validity != nullptr;
  gptr<const uint8_t> heap = GC<uint8_t>(t.buf2);
  gptr<int32_t> off = GM<int32_t>(t.out_data);
  gptr<int64_t> off64 = GM<int64_t>(t.out_data);
  const bool large = (t.flags & 1) != 0;  // LargeUtf8 / LargeList: int64 offsets (arrow_large_buffer_size)
  gptr<uint8_t> data = GM<uint8_t>(t.out_aux);
  if (row0 == 0 && threadIdx.x == 0) {
    if (large) off64[0] = 0;
    else off[0] = 0;
  }
  for (int k = 0; k < kTileRows / kBlockThreads; k++) {
    const int r = threadIdx.x + k * kBlockThreads;
    if (k * kBlockThreads >= n) break;  // uniform
    u32x4 s = {0u, 0u, 0u, 0u};
    uint32_t len = 0;
    if (r < n) {
      s = str[r];
      len = enc_row_valid(valid, has, row0 + r) ? (t.kind == MI_K_ENC_LIST32 ? s.z : s.x) : 0u;
    }
    int64_t total;
    const int64_t ex = block_exclusive_scan(static_cast<int64_t>(len), &total, lds4);
    const int64_t pos = base + ex;
    if (r < n) {
      if (large) off64[row0 + r + 1] = pos + len;
      else off[row0 + r + 1] = static_cast<int32_t>(pos + len);
    }
    if (t.kind == MI_K_ENC_LIST32) {  // offsets only
      base += total;
      continue;
    }
    // LDS image: byte i of the sub-block's payload lives at stage[shift + i], shift = base mod 16, so that 16-byte
    // aligned global addresses are 16-byte aligned LDS addresses
    const int shift = static_cast<int>(base & 15);
    const bool staged = total + shift <= kEncStage;
    uint8_t* dst_l = stage + shift + static_cast<int>(ex);
    gptr<uint8_t> dst_g = data + pos;
    if (len != 0) {
      if (s.x <= 12) {
        const uint32_t w0 = s.y, w1 = s.z, w2 = s.w;
        for (uint32_t j = 0; j < len; j++) {
          const uint32_t w = j < 4 ? w0 : (j < 8 ? w1 : w2);
          const uint8_t byte = static_cast<uint8_t>(w >> (8 * (j & 3)));
          if (staged) dst_l[j] = byte; else dst_g[j] = byte;
        }
      } else {
        const uint64_t p = static_cast<uint64_t>(s.z) | (static_cast<uint64_t>(s.w) << 32);
        gptr<const uint8_t> src = heap + (p - t.ptr_base);
        const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(src) & 3);
        gptr<const uint32_t> q = (gptr<const uint32_t>)(src - mis);
        const uint32_t ndw = (mis + len + 3) >> 2;
        uint32_t j = 0;
        for (uint32_t d = 0; d < ndw; d++) {
          const uint32_t w = q[d];
          const uint32_t first = d == 0 ? mis : 0;
          for (uint32_t bidx = first; bidx < 4 && j < len; bidx++, j++) {
            const uint8_t byte = static_cast<uint8_t>(w >> (8 * bidx));
            if (staged) dst_l[j] = byte; else dst_g[j] = byte;
          }
        }
      }
    }
    if (staged) {
      __syncthreads();
      // stage[shift .. shift+total) -> data[base .. base+total): unaligned head and tail bytewise, the middle as 16-byte rows
      const int64_t g0 = base, g1 = base + total;
      const int64_t a0 = (g0 + 15) & ~static_cast<int64_t>(15), a1 = g1 & ~static_cast<int64_t>(15);
      if (a0 >= a1) {
        for (int64_t i = g0 + threadIdx.x; i < g1; i += kBlockThreads) data[i] = stage[shift + (i - g0)];
      } else {
        for (int64_t i = g0 + threadIdx.x; i < a0; i += kBlockThreads) data[i] = stage[shift + (i - g0)];
        for (int64_t i = a1 + threadIdx.x; i < g1; i += kBlockThreads) data[i] = stage[shift + (i - g0)];
        const int nvec = static_cast<int>((a1 - a0) >> 4);
        const u32x4* ls = reinterpret_cast<const u32x4*>(stage + shift + (a0 - g0));
        gptr<u32x4> gd = (gptr<u32x4>)(data + a0);
        for (int i = threadIdx.x; i < nvec; i += kBlockThreads) __builtin_nontemporal_store(ls[i], gd + i);
      }
      __syncthreads();
    }
    base += total;
  }
}

// LISTS_ONLY = false: every tile (variant 0 of the A/B knob); true: only the MI_K_ENC_LIST32 tiles, beside encode_string_v5
template <bool LISTS_ONLY>
__global__ __launch_bounds__(kBlockThreads) void encode_string(const mi_col_task* __restrict__ tasks,
                                                               const uint32_t* __restrict__ tile_begin,
                                                               const uint32_t* __restrict__ tile_task, int n_tasks,
                                                               uint32_t total_tiles, const int64_t* __restrict__ tile_sums,
                                                               int64_t* __restrict__ null_counts) {
  __shared__ int64_t lds4[kBlockThreads / 64];
  __shared__ __attribute__((aligned(16))) uint8_t stage[kEncStage + 16];
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE();
    if (LISTS_ONLY && t.kind != MI_K_ENC_LIST32) continue;  // uniform
    enc_tile_validity(t, row0, n, null_counts);
    encode_string_tile_generic(t, row0, n, tile_sums[tile], lds4, stage);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------- fused Q6-style consumer
// sum(a * b) WHERE lo_k <= f_k < hi_k (all k) over decoded fixed-width vectors; NULL in any filter column drops the
// row (SQL comparison semantics), NULL in a or b makes the product NULL, which SUM ignores.  128-bit accumulation
// (DuckDB sums DECIMAL products in a hugeint).  28 bytes per row for TPC-H Q6: HBM / L2 bound.
__device__ __forceinline__ int64_t load_sint(const void* p, int width, int64_t i) {
  switch (width) {
    case 1: return GC<int8_t>(p)[i];
    case 2: return GC<int16_t>(p)[i];
    case 4: return GC<int32_t>(p)[i];
    default: return GC<int64_t>(p)[i];
  }
}

__global__ __launch_bounds__(kBlockThreads) void agg_sum_product(AggSumProductArgs a, unsigned long long* __restrict__ acc) {
  __int128 sum = 0;
  unsigned long long selected = 0;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlockThreads;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlockThreads + threadIdx.x; i < a.nrows; i += stride) {
    bool keep = true;
    for (int k = 0; k < a.n_filters; k++) {
      const bool valid = a.fvalid[k] == nullptr || ((GC<uint64_t>(a.fvalid[k])[i >> 6] >> (i & 63)) & 1);
      const int64_t v = load_sint(a.fcol[k], a.fwidth[k], i);
      keep = keep && valid && v >= a.lo[k] && v < a.hi[k];
    }
    if (keep) {
      selected++;
      const bool va = a.avalid == nullptr || ((GC<uint64_t>(a.avalid)[i >> 6] >> (i & 63)) & 1);
      const bool vb = a.bvalid == nullptr || ((GC<uint64_t>(a.bvalid)[i >> 6] >> (i & 63)) & 1);
      if (va && vb) sum += static_cast<__int128>(load_sint(a.a, a.awidth, i)) * static_cast<__int128>(load_sint(a.b, a.bwidth, i));
    }
  }
  unsigned long long lo = static_cast<unsigned long long>(sum), hi = static_cast<unsigned long long>(sum >> 64);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long olo = __shfl_down(lo, d, 64), ohi = __shfl_down(hi, d, 64);
    const unsigned long long osel = __shfl_down(selected, d, 64);
    const unsigned long long nlo = lo + olo;
    hi += ohi + (nlo < lo ? 1ull : 0ull);
    lo = nlo;
    selected += osel;
  }
  if ((threadIdx.x & 63) == 0) {
    const unsigned long long old = atomicAdd(&acc[0], lo);
    const unsigned long long carry = (old + lo < old) ? 1ull : 0ull;
    if (hi + carry) atomicAdd(&acc[1], hi + carry);
    if (selected) atomicAdd(&acc[2], selected);
  }
}

// ---- dword-granular LDS assembly ------------------------------------------------------------------------------
// W[0..N] hold a source byte stream that starts at byte `sh` (0..3) of W[0] (W[N+1] readable, zero); writes its first
// cnt <= 4N bytes at dst (LDS): <= 3 head bytes up to dst's 4-byte boundary, whole dwords funnel-shifted to the
// destination phase with v_alignbyte_b32, <= 3 tail bytes.  Straight-line code (predicated stores, no loops): the
// kernel is bound by VALU issue, not by memory.  W is consumed (shifted in place).
template <int N>
__device__ __forceinline__ void lds_put_stream(uint8_t* dst, uint32_t (&W)[N + 2], uint32_t sh, uint32_t cnt) {
  const uint32_t dm = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst)) & 3u;
  uint32_t head = (4u - dm) & 3u;
  if (head > cnt) head = cnt;
  const uint32_t first = __builtin_amdgcn_alignbyte(W[1], W[0], sh);
  if (head > 0) dst[0] = static_cast<uint8_t>(first);
  if (head > 1) dst[1] = static_cast<uint8_t>(first >> 8);
  if (head > 2) dst[2] = static_cast<uint8_t>(first >> 16);
  uint32_t tsh = sh + head;  // 0..6: where the dword stream starts inside W
  if (tsh >= 4) {
#pragma unroll
    for (int i = 0; i <= N; i++) W[i] = W[i + 1];
    tsh -= 4;
  }
  const uint32_t nd = (cnt - head) >> 2;
  uint32_t* d4 = reinterpret_cast<uint32_t*>(dst + head);
  uint32_t tailw = 0;
#pragma unroll
  for (int i = 0; i <= N; i++) {
    const uint32_t v = __builtin_amdgcn_alignbyte(W[i + 1], W[i], tsh);
    if (static_cast<uint32_t>(i) < nd) d4[i] = v;
    if (static_cast<uint32_t>(i) == nd) tailw = v;
  }
  const uint32_t tc = (cnt - head) & 3u;
  uint8_t* tp = dst + head + 4 * nd;
  if (tc > 0) tp[0] = static_cast<uint8_t>(tailw);
  if (tc > 1) tp[1] = static_cast<uint8_t>(tailw >> 8);
  if (tc > 2) tp[2] = static_cast<uint8_t>(tailw >> 16);
}

// The first <= 48 bytes of a heap string starting at `src`: 13 aligned dwords cover them at any misalignment.
__device__ __forceinline__ void heap_load13(gptr<const uint8_t> src, uint32_t cnt, uint32_t (&W)[14]) {
  const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(src) & 3);
  gptr<const uint32_t> q = (gptr<const uint32_t>)(src - mis);
  const uint32_t ndw = (mis + (cnt < 48u ? cnt : 48u) + 3) >> 2;
#pragma unroll
  for (int d = 0; d < 13; d++) W[d] = static_cast<uint32_t>(d) < ndw ? __builtin_nontemporal_load(q + d) : 0u;
  W[13] = 0u;
}

// Bytes [c0, c0 + cnt) of one string -> LDS at dst.  W: the string's first 48 heap bytes (heap_load13 of the string
// start) when it is a long string; it is used when c0 == 0 and is scratch otherwise.
__device__ __forceinline__ void string_bytes_to_lds(uint8_t* dst, const u32x4& s, gptr<const uint8_t> heap, uint64_t ptr_base,
                                                    uint32_t c0, uint32_t cnt, uint32_t (&W)[14]) {
  if (s.x <= 12) {
    uint32_t a = s.y, b = s.z, c = s.w;
    if (c0 >= 8) { a = c; b = 0u; c = 0u; }
    else if (c0 >= 4) { a = b; b = c; c = 0u; }
    uint32_t S[5] = {a, b, c, 0u, 0u};
    lds_put_stream<3>(dst, S, c0 & 3u, cnt);
    return;
  }
  const uint64_t p = static_cast<uint64_t>(s.z) | (static_cast<uint64_t>(s.w) << 32);
  gptr<const uint8_t> src = heap + (p - ptr_base) + c0;
  const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(src) & 3);
  uint32_t done = 0;
  if (c0 == 0) {
    lds_put_stream<12>(dst, W, mis, cnt < 48u ? cnt : 48u);
    done = 48;
  }
#pragma clang loop unroll(disable)
  for (; done < cnt; done += 48) {
    heap_load13(src + done, cnt - done, W);
    lds_put_stream<12>(dst + done, W, mis, cnt - done < 48u ? cnt - done : 48u);
  }
}

// Inclusive wave64 scan on the DPP crossbar (no LDS traffic): row_shr 1/2/4/8 inside each 16-lane row, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xf, 0xf, false));
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xf, 0xf, false));
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xf, 0xf, false));
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118, 0xf, 0xf, false));
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xa, 0xf, false));
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xc, 0xf, false));
  return v;
}

// K7d pass 3, the shipped formulation.  Same outputs as encode_string (kept as variant 0 of the A/B knob and as the
// fallback for sub-blocks with huge strings).  rocprofv3 counters showed the first formulations bound by VALU issue
// (1.1 G wave-instructions per SF10 table, 165 per 64 one-byte strings), not by HBM, so everything here is about fewer
// instructions per row and fewer dependent round trips:
//  * the tile's validity words are fetched once (LDS); the NEXT sub-block's string_t is requested (clamped address,
//    so unconditionally) before this one is touched;
//  * a long string's first 48 heap bytes arrive as ONE batch of 13 aligned dword loads (the original walked the string
//    one dependent dword at a time: 7 HBM round trips for a 27-byte l_comment);
//  * the scan is a 6-step DPP wave scan + one LDS exchange of the 4 wave totals, in 32-bit arithmetic (a sub-block
//    holding a string of >= 8 MiB is handed to the 64-bit formulation);
//  * payload is assembled in LDS with dword stores (source stream funnel-shifted with v_alignbyte_b32 to the string
//    start, then to the destination's 4-byte phase; <= 3 head and <= 3 tail byte stores, all predicated straight-line
//    code), in windows of <= 8 KiB so a sub-block of any size is staged; two stage buffers alternate: one barrier per
//    window; all positions inside a sub-block are 32-bit;
//  * the stage leaves as coalesced 16-byte nontemporal stores through a 16-byte aligned uniform base pointer.
constexpr uint32_t kEncBigLen = 1u << 23;
constexpr int kEncStage5 = 8 * 1024;       // bytes per stage buffer
constexpr int kEncStageBuf = kEncStage5 + 64;
static_assert(2 * kEncStageBuf >= kEncStage + 16, "the 64-bit formulation borrows both stage buffers");

template <int OCC>
__global__ __launch_bounds__(kBlockThreads, OCC) void encode_string_v5(const mi_col_task* __restrict__ tasks,
                                                                  const uint32_t* __restrict__ tile_begin,
                                                                  const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                  uint32_t total_tiles, int64_t* __restrict__ tile_sums,
                                                                  int64_t* __restrict__ null_counts) {
  constexpr int kWaves = kBlockThreads / 64;
  static_assert(kWaves == 4, "wave totals travel as one 16-byte LDS row");
  __shared__ int64_t lds4[kWaves];
  __shared__ uint64_t s_valid[kTileRows / 64];
  __shared__ __attribute__((aligned(16))) uint32_t s_tot[2][kWaves];
  __shared__ __attribute__((aligned(16))) uint8_t stage[2 * kEncStageBuf];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE();
    if (t.kind == MI_K_ENC_LIST32) continue;  // uniform: list offsets are the other launch's tiles (encode_string<1>)
    enc_tile_validity(t, row0, n, null_counts, s_valid);
    gptr<const u32x4> str = GC<u32x4>(t.buf1) + row0;
    gptr<const uint8_t> heap = GC<uint8_t>(t.buf2);
    gptr<int32_t> offp = GM<int32_t>(t.out_data) + row0 + 1;
    gptr<int64_t> offp64 = GM<int64_t>(t.out_data) + row0 + 1;
    const bool large = (t.flags & 1) != 0;  // LargeUtf8: int64 offsets (arrow_large_buffer_size)
    gptr<uint8_t> data = GM<uint8_t>(t.out_aux);
    const int64_t tile_base = tile_sums[tile];
    int64_t base = tile_base;
    if (row0 == 0 && threadIdx.x == 0) {
      if (large) offp64[-1] = 0;
      else offp[-1] = 0;
    }
    const int nsub = (n + kBlockThreads - 1) / kBlockThreads;
    u32x4 nxt = __builtin_nontemporal_load(str + (static_cast<int>(threadIdx.x) < n ? static_cast<int>(threadIdx.x) : n - 1));
    __syncthreads();  // s_valid
    uint32_t buf = 0;
#pragma clang loop unroll(disable)
    for (int k = 0; k < nsub; k++) {
      const int r = threadIdx.x + k * kBlockThreads;
      const u32x4 s = nxt;
      {
        const int rn = r + kBlockThreads;
        nxt = __builtin_nontemporal_load(str + (rn < n ? rn : n - 1));
      }
      const bool ok = r < n && ((s_valid[r >> 6] >> (r & 63)) & 1);
      const uint32_t len = ok ? s.x : 0u;
      uint32_t W[14];
      if (len > 12) {
        const uint64_t p = static_cast<uint64_t>(s.z) | (static_cast<uint64_t>(s.w) << 32);
        heap_load13(heap + (p - t.ptr_base), len, W);
      }
      // exclusive scan of the sub-block's lengths
      const uint32_t incl = wave_inclusive_scan_u32(len);
      const bool wave_big = __any(len >= kEncBigLen);
      if (lane == 63) s_tot[k & 1][wave] = wave_big ? 0x80000000u : incl;
      __syncthreads();
      const u32x4 tot = *reinterpret_cast<const u32x4*>(&s_tot[k & 1][0]);
      if ((tot.x | tot.y | tot.z | tot.w) & 0x80000000u) {
        // uniform: a string of >= 8 MiB in this sub-block -- 32-bit sums may wrap.  The tile is handed to the 64-bit
        // formulation (encode_string_redo, launched right after this kernel) by setting the sign bit of its base; the
        // rare path stays out of this kernel's register budget (inlined here it cost 28 VGPRs = two occupancy steps)
        if (threadIdx.x == 0) tile_sums[tile] = tile_base | static_cast<int64_t>(0x8000000000000000ull);
        break;
      }
      const uint32_t before = (wave > 0 ? tot.x : 0u) + (wave > 1 ? tot.y : 0u) + (wave > 2 ? tot.z : 0u);
      const uint32_t total = tot.x + tot.y + tot.z + tot.w;
      const uint32_t ex = before + incl - len;
      if (r < n) {
        if (large) offp64[r] = base + ex + len;
        else offp[r] = static_cast<int32_t>(base + ex + len);
      }
      for (uint32_t w0 = 0; w0 < total;) {  // uniform: stage windows
        const uint32_t shiftw = static_cast<uint32_t>((base + w0) & 15);
        const uint32_t room = static_cast<uint32_t>(kEncStage5) - shiftw;
        const uint32_t w1 = total - w0 < room ? total : w0 + room;
        uint8_t* st = stage + buf * kEncStageBuf;
        const uint32_t lo = ex > w0 ? ex : w0, hi = ex + len < w1 ? ex + len : w1;
        if (lo < hi) string_bytes_to_lds(st + shiftw + (lo - w0), s, heap, t.ptr_base, lo - ex, hi - lo, W);
        __syncthreads();
        // stage bytes [shiftw, end) -> gbase[shiftw, end); gbase is 16-byte aligned
        const uint32_t end = shiftw + (w1 - w0);
        gptr<uint8_t> gbase = data + (base + w0) - shiftw;
        const uint32_t nch = (end + 15) >> 4;
        for (uint32_t c = threadIdx.x; c < nch; c += kBlockThreads) {
          const uint32_t clo = c << 4;
          if (clo >= shiftw && clo + 16 <= end)
            __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(st + clo), (gptr<u32x4>)(gbase + clo));
        }
        if (threadIdx.x < 32) {  // the (at most two) partial 16-byte rows, one byte per lane
          const uint32_t lastlo = (end - 1) & ~15u;
          const uint32_t idx = threadIdx.x < 16 ? threadIdx.x : lastlo + (threadIdx.x - 16);
          const bool first_partial = shiftw != 0 || end < 16;
          const bool last_partial = (end & 15u) != 0 && lastlo != 0;
          const bool mine = threadIdx.x < 16 ? first_partial : last_partial;
          if (mine && idx >= shiftw && idx < end) gbase[idx] = st[idx];
        }
        buf ^= 1u;
        w0 = w1;
      }
      base += total;
    }
    __syncthreads();  // s_valid / stage are rewritten by the next tile
  }
}

// Tiles encode_string_v5 gave up on (sign bit of tile_sums set): 64-bit positions, sub-block by sub-block.  A small
// persistent grid sweeps the flags 256 at a time; validity bitmap and NULL count were already written by v5.
__global__ __launch_bounds__(kBlockThreads) void encode_string_redo(const mi_col_task* __restrict__ tasks,
                                                                    const uint32_t* __restrict__ tile_begin,
                                                                    const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                    uint32_t total_tiles, const int64_t* __restrict__ tile_sums,
                                                                    int64_t* __restrict__ null_counts) {
  __shared__ int64_t lds4[kBlockThreads / 64];
  __shared__ __attribute__((aligned(16))) uint8_t stage[kEncStage + 16];
  __shared__ uint32_t s_todo[kBlockThreads];
  __shared__ uint32_t s_ntodo;
  for (uint32_t first = blockIdx.x * kBlockThreads; first < total_tiles; first += gridDim.x * kBlockThreads) {
    if (threadIdx.x == 0) s_ntodo = 0;
    __syncthreads();
    const uint32_t mine = first + threadIdx.x;
    if (mine < total_tiles && tile_sums[mine] < 0) s_todo[atomicAdd(&s_ntodo, 1u)] = mine;
    __syncthreads();
    const uint32_t ntodo = s_ntodo;
    for (uint32_t j = 0; j < ntodo; j++) {
      const uint32_t tile = s_todo[j];
      MI_TILE_PROLOGUE();
      encode_string_tile_generic(t, row0, n, tile_sums[tile] & 0x7FFFFFFFFFFFFFFFll, lds4, stage);
      __syncthreads();
    }
    __syncthreads();
  }
}

inline uint32_t grid_for(uint32_t total_tiles, int grid_blocks) {
  return total_tiles < static_cast<uint32_t>(grid_blocks) ? total_tiles : static_cast<uint32_t>(grid_blocks);
}

}  // namespace

int TileRowsOfClass(int cls) {
  switch (cls) {
    case kClassCopy: return kCopyTileRows;
    case kClassDec128: return kDecTileRows;
    default: return kTileRows;
  }
}

int ClassOfKind(int32_t kind) {
  switch (kind) {
    case MI_K_COPY: return kClassCopy;
    case MI_K_DEC128: return kClassDec128;
    case MI_K_STR32: case MI_K_STR64: case MI_K_FIXED_BINARY: return kClassString;
    case MI_K_BOOL: case MI_K_DATE64: case MI_K_MUL_I32: case MI_K_MUL_I64: case MI_K_DIV_I64: case MI_K_DURATION:
    case MI_K_DICT: case MI_K_INTERVAL_MONTHS: case MI_K_INTERVAL_MDN: case MI_K_NARROW: case MI_K_HALF_FLOAT:
    case MI_K_NULL: case MI_K_STRVIEW: case MI_K_LIST32: case MI_K_LIST64: case MI_K_STRUCT: return kClassMisc;
    case MI_K_ENC_COPY: case MI_K_ENC_DEC128: case MI_K_ENC_BOOL: case MI_K_ENC_VALIDITY: return kClassEncFixed;
    case MI_K_ENC_STR32: case MI_K_ENC_LIST32: return kClassEncString;
    default: return -1;
  }
}

// Measurement knobs (A/B of kernel variants inside one process; see tools/ab_bench.py).  Defaults are the shipped
// configuration; MI_TUNE_* environment variables or SetTune() override them.
struct Tune {
  int copy_variant = 2;    // 1: 16-byte loads from 8-byte aligned IPC buffers (unaligned-access mode); 2: + nontemporal
  int dec_variant = 2;
  int string_variant = 2;  // 1: 8 rows per lane in flight; 2: + nontemporal stores
  int blocks_per_cu = 0;   // 0: one workgroup per tile (the hardware dispatcher balances the tiles)
  int use_tile_table = 1;
  int misc_light = 1;          // 1: one wave per tile for the common flat kinds (transcode_misc_light)
  int enc_string_variant = 1;  // 1: encode_string_v5 (batched heap loads, dword LDS assembly, windows); 0: encode_string
};
static Tune& TuneRef() {
  static Tune t = [] {
    Tune x;
    auto env = [](const char* n, int d) { const char* v = std::getenv(n); return v ? std::atoi(v) : d; };
    x.copy_variant = env("MI_TUNE_COPY", x.copy_variant);
    x.dec_variant = env("MI_TUNE_DEC128", x.dec_variant);
    x.string_variant = env("MI_TUNE_STRING", x.string_variant);
    x.blocks_per_cu = env("MI_TUNE_GRID", x.blocks_per_cu);
    x.use_tile_table = env("MI_TUNE_TILE_TABLE", x.use_tile_table);
    x.enc_string_variant = env("MI_TUNE_ENC_STRING", x.enc_string_variant);
    x.misc_light = env("MI_TUNE_MISC_LIGHT", x.misc_light);
    return x;
  }();
  return t;
}
bool SetTune(const char* knob, int value) {
  Tune& t = TuneRef();
  const std::string k = knob ? knob : "";
  if (k == "copy") t.copy_variant = value;
  else if (k == "dec128") t.dec_variant = value;
  else if (k == "string") t.string_variant = value;
  else if (k == "grid") t.blocks_per_cu = value;
  else if (k == "tile_table") t.use_tile_table = value;
  else if (k == "enc_string") t.enc_string_variant = value;
  else if (k == "misc_light") t.misc_light = value;
  else return false;
  return true;
}

hipError_t LaunchTranscode(int cls, const mi_col_task* d_tasks, const uint32_t* d_tile_begin, const uint32_t* d_tile_task,
                           int32_t n_tasks, uint32_t total_tiles, uint32_t* d_status, uint32_t misc_groups, int num_cus, hipStream_t stream) {
  if (total_tiles == 0) return hipSuccess;
  const Tune& tune = TuneRef();
  const uint32_t blocks = tune.blocks_per_cu > 0 ? grid_for(total_tiles, num_cus * tune.blocks_per_cu) : total_tiles;
  const dim3 grid(blocks), block(kBlockThreads);
  const uint32_t* tt = tune.use_tile_table ? d_tile_task : nullptr;
#define MI_LAUNCH(KERNEL) hipLaunchKernelGGL(KERNEL, grid, block, 0, stream, d_tasks, d_tile_begin, tt, n_tasks, total_tiles, d_status)
  switch (cls) {
    case kClassCopy:
      if (tune.copy_variant == 2) MI_LAUNCH(transcode_copy<2>);
      else if (tune.copy_variant == 1) MI_LAUNCH(transcode_copy<1>);
      else MI_LAUNCH(transcode_copy<0>);
      break;
    case kClassDec128:
      if (tune.dec_variant == 2) MI_LAUNCH(transcode_dec128<2>);
      else if (tune.dec_variant == 1) MI_LAUNCH(transcode_dec128<1>);
      else MI_LAUNCH(transcode_dec128<0>);
      break;
    case kClassString:
      if (tune.string_variant == 3) MI_LAUNCH(transcode_string<3>);
      else if (tune.string_variant == 2) MI_LAUNCH(transcode_string<2>);
      else if (tune.string_variant == 1) MI_LAUNCH(transcode_string<1>);
      else MI_LAUNCH(transcode_string<0>);
      break;
    case kClassMisc:
      if (misc_groups & 1u) {
        if (tune.misc_light)
          hipLaunchKernelGGL(transcode_misc_light, grid, dim3(kLightThreads), 0, stream, d_tasks, d_tile_begin, tt, n_tasks, total_tiles, d_status);
        else
          MI_LAUNCH(transcode_misc<0>);
      }
      if (misc_groups & 2u) MI_LAUNCH(transcode_misc<1>);
      if (misc_groups & 4u) MI_LAUNCH(transcode_misc<2>);
      break;
    default:
      return hipErrorInvalidValue;
  }
#undef MI_LAUNCH
  return hipGetLastError();
}

hipError_t LaunchAggSumProduct(const AggSumProductArgs& args, unsigned long long* d_acc, int num_cus, hipStream_t stream) {
  if (args.nrows <= 0) return hipSuccess;
  const int64_t want = (args.nrows + kBlockThreads * 8 - 1) / (kBlockThreads * 8);   // ~8 rows per lane
  const uint32_t grid = static_cast<uint32_t>(std::min<int64_t>(want, static_cast<int64_t>(num_cus) * 16));
  hipLaunchKernelGGL(agg_sum_product, dim3(grid ? grid : 1), dim3(kBlockThreads), 0, stream, args, d_acc);
  return hipGetLastError();
}

hipError_t LaunchFilterRange(const void* values, int32_t width, const void* validity, int64_t nrows, int64_t lo,
                             int64_t hi, mi_sel_t* sel_out, uint32_t* count_out, hipStream_t stream) {
  if (nrows <= 0) return hipSuccess;
  const int64_t windows = (nrows + kTileRows - 1) / kTileRows;
  const uint32_t grid = static_cast<uint32_t>((windows + kFilterWindows - 1) / kFilterWindows);
  const uint64_t* v = static_cast<const uint64_t*>(validity);
  switch (width) {
    case 4:
      hipLaunchKernelGGL(filter_range<int32_t>, dim3(grid), dim3(kBlockThreads), 0, stream,
                         static_cast<const int32_t*>(values), v, nrows, lo, hi, sel_out, count_out);
      break;
    case 8:
      hipLaunchKernelGGL(filter_range<int64_t>, dim3(grid), dim3(kBlockThreads), 0, stream,
                         static_cast<const int64_t*>(values), v, nrows, lo, hi, sel_out, count_out);
      break;
    case 2:
      hipLaunchKernelGGL(filter_range<int16_t>, dim3(grid), dim3(kBlockThreads), 0, stream,
                         static_cast<const int16_t*>(values), v, nrows, lo, hi, sel_out, count_out);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// Encode launches follow the decode rule: one workgroup per tile (the dispatcher balances them) and the tile -> task
// table, unless the measurement knobs ask for the persistent grid / binary search.
static uint32_t enc_grid(uint32_t total_tiles, int grid_blocks) {
  return TuneRef().blocks_per_cu > 0 ? grid_for(total_tiles, grid_blocks) : total_tiles;
}

hipError_t LaunchEncodeStringTileSums(const mi_col_task* d_tasks, const uint32_t* d_tile_begin, const uint32_t* d_tile_task,
                                      int32_t n_tasks, uint32_t total_tiles, int64_t* d_tile_sums, int grid_blocks,
                                      hipStream_t stream) {
  if (total_tiles == 0) return hipSuccess;
  const uint32_t* tt = TuneRef().use_tile_table ? d_tile_task : nullptr;
  hipLaunchKernelGGL(encode_string_tile_sums, dim3(enc_grid(total_tiles, grid_blocks)), dim3(kBlockThreads), 0, stream,
                     d_tasks, d_tile_begin, tt, n_tasks, total_tiles, d_tile_sums);
  return hipGetLastError();
}

hipError_t LaunchEncodeStringScan(const mi_col_task* d_tasks, const uint32_t* d_tile_begin, int32_t n_tasks,
                                  int64_t* d_tile_sums, uint32_t* d_status, hipStream_t stream) {
  if (n_tasks == 0) return hipSuccess;
  const uint32_t grid = n_tasks < 2048 ? static_cast<uint32_t>(n_tasks) : 2048u;
  hipLaunchKernelGGL(encode_string_scan, dim3(grid), dim3(kBlockThreads), 0, stream, d_tasks, d_tile_begin, n_tasks,
                     d_tile_sums, d_status);
  return hipGetLastError();
}

hipError_t LaunchEncodeFixed(const mi_col_task* d_tasks, const uint32_t* d_tile_begin, const uint32_t* d_tile_task,
                             int32_t n_tasks, uint32_t total_tiles, int64_t* d_null_counts, int grid_blocks,
                             hipStream_t stream) {
  if (total_tiles == 0) return hipSuccess;
  const uint32_t* tt = TuneRef().use_tile_table ? d_tile_task : nullptr;
  hipLaunchKernelGGL(encode_fixed, dim3(enc_grid(total_tiles, grid_blocks)), dim3(kBlockThreads), 0, stream, d_tasks,
                     d_tile_begin, tt, n_tasks, total_tiles, d_null_counts);
  return hipGetLastError();
}

hipError_t LaunchEncodeString(const mi_col_task* d_tasks, const uint32_t* d_tile_begin, const uint32_t* d_tile_task,
                              int32_t n_tasks, uint32_t total_tiles, const int64_t* d_tile_sums, int64_t* d_null_counts,
                              int grid_blocks, uint32_t groups, hipStream_t stream) {
  if (total_tiles == 0) return hipSuccess;
  const uint32_t* tt = TuneRef().use_tile_table ? d_tile_task : nullptr;
  const int ev = TuneRef().enc_string_variant;
  const dim3 grid(enc_grid(total_tiles, grid_blocks)), block(kBlockThreads);
#define MI_ENC_LAUNCH(KERNEL) \
  hipLaunchKernelGGL(KERNEL, grid, block, 0, stream, d_tasks, d_tile_begin, tt, n_tasks, total_tiles, d_tile_sums, d_null_counts)
  if (ev == 0) {
    MI_ENC_LAUNCH(encode_string<false>);
  } else {
    if (groups & 1u) {  // strings
      int64_t* sums = const_cast<int64_t*>(d_tile_sums);
      if (ev == 2) hipLaunchKernelGGL(encode_string_v5<5>, grid, block, 0, stream, d_tasks, d_tile_begin, tt, n_tasks, total_tiles, sums, d_null_counts);
      else if (ev == 3) hipLaunchKernelGGL(encode_string_v5<4>, grid, block, 0, stream, d_tasks, d_tile_begin, tt, n_tasks, total_tiles, sums, d_null_counts);
      else hipLaunchKernelGGL(encode_string_v5<6>, grid, block, 0, stream, d_tasks, d_tile_begin, tt, n_tasks, total_tiles, sums, d_null_counts);
      const uint32_t sweep = (total_tiles + kBlockThreads - 1) / kBlockThreads;
      hipLaunchKernelGGL(encode_string_redo, dim3(sweep < 512u ? sweep : 512u), block, 0, stream, d_tasks, d_tile_begin, tt, n_tasks,
                         total_tiles, d_tile_sums, d_null_counts);
    }
    if (groups & 2u) MI_ENC_LAUNCH(encode_string<true>);  // list / map offsets
  }
#undef MI_ENC_LAUNCH
  return hipGetLastError();
}

}  // namespace device
}  // namespace miarrow
