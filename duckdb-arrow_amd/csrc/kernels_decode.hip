// kernels_decode.hip -- Arrow IPC buffers -> DuckDB vectors (K1-K5 + nested types), one kernel per class.
//
// Kernel classes: each class is its own kernel so that it gets the register budget of its own inner loop (a single
// switch over all kinds needed 154 VGPRs = 3 waves/SIMD).  A plan groups its tasks by class (engine.cpp).
//
// Semantics restated per kernel from DuckDB's ArrowToDuckDB (call sites in the reference:
// src/scanner/scan_arrow_ipc.cpp:56, src/file_scanner/arrow_file_scan.cpp:68-72); canonical values for slots upstream
// leaves undefined: NULL rows of converted columns = 0, validity pad bits = 1.
#include "device_common.hpp"

#include <algorithm>

namespace miarrow {
namespace device {

namespace {

// ---------------------------------------------------------------------------------------------------- K3a
// Fixed-width direct conversion: a coalesced copy of n*width bytes (copy_bytes, device_common.hpp).
__global__ __launch_bounds__(kBlockThreads) void transcode_copy(const mi_col_task* __restrict__ tasks,
                                                                const uint32_t* __restrict__ tile_begin,
                                                                const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                uint32_t total_tiles, uint32_t* __restrict__ status) {
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE_ROWS(kCopyTileRows);
    tile_validity(t, row0, n);
    const int w = static_cast<int>(t.param);
    copy_bytes(GC<uint8_t>(t.buf1) + (t.row_offset + row0) * w, GM<uint8_t>(t.out_data) + row0 * w, n * w);
  }
}

// ---------------------------------------------------------------------------------------------------- K3b
// decimal128 {u64 lower, i64 upper} -> int16/32/64 for valid rows (Hugeint::TryCast: value fits by precision);
// NULL rows canonical 0.  Each lane reads the whole 16-byte value (the upper half is what proves the range).
template <typename OUT>
__device__ __forceinline__ void tile_dec128(const mi_col_task& t, int64_t row0, int n, uint32_t* status, const uint64_t* s_valid) {
  gptr<const uint8_t> src = GC<uint8_t>(t.buf1) + (t.row_offset + row0) * 16;
  gptr<OUT> out = GM<OUT>(t.out_data) + row0;
  const bool has_nulls = tile_needs_mask(t);
  uint32_t err = 0;
#pragma unroll 4
  for (int r = threadIdx.x; r < n; r += kBlockThreads) {
    const u32x4 v = ld16((gptr<const u32x4_a4>)(src + 16 * static_cast<int64_t>(r)));  // 8-byte aligned source: unaligned-access mode
    const uint64_t lower = static_cast<uint64_t>(v.x) | (static_cast<uint64_t>(v.y) << 32);
    const int64_t upper = static_cast<int64_t>(static_cast<uint64_t>(v.z) | (static_cast<uint64_t>(v.w) << 32));
    OUT o = 0;
    if (row_valid(s_valid, has_nulls, r)) {
      o = static_cast<OUT>(lower);
      const int64_t sext = static_cast<int64_t>(o);
      if (static_cast<uint64_t>(sext) != lower || upper != (sext >> 63)) err = MI_ST_DECIMAL_RANGE;
    }
    __builtin_nontemporal_store(o, out + r);
  }
  raise(status, err);
}

__global__ __launch_bounds__(kBlockThreads) void transcode_dec128(const mi_col_task* __restrict__ tasks,
                                                                  const uint32_t* __restrict__ tile_begin,
                                                                  const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                  uint32_t total_tiles, uint32_t* __restrict__ status) {
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE_ROWS(kDecTileRows);
    __shared__ uint64_t s_valid[kDecTileRows / 64];
    if (tile_needs_mask(t)) __syncthreads();  // the previous tile's lanes are done with the mask
    tile_validity(t, row0, n, s_valid);
    if (t.param == 8) tile_dec128<int64_t>(t, row0, n, status, s_valid);
    else if (t.param == 4) tile_dec128<int32_t>(t, row0, n, status, s_valid);
    else tile_dec128<int16_t>(t, row0, n, status, s_valid);
  }
}

// ---------------------------------------------------------------------------------------------------- K4
// utf8 / binary with int32 or int64 offsets -> string_t, offsets validated like NANOARROW_VALIDATION_LEVEL_FULL.
// Every lane owns the 8 rows {tid + 256k} of the tile and issues all its loads before the first store (8 independent
// offset loads, then up to 8 x 4 payload dwords), so one wave has 8 rows in flight (-8 % time against 2 in flight).
// off[r+1] comes from the neighbouring lane (one permute) except at the wave edge and at the last row.  One 16-byte
// nontemporal store per row (1 KiB per wave store).
template <typename OFF>
__device__ __forceinline__ void tile_string(const mi_col_task& t, int64_t row0, int n, uint32_t* status, const uint64_t* s_valid) {
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
    a[k] = r < n ? off[r] : 0;
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
      __builtin_nontemporal_store(s[k], out + r);
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

__global__ __launch_bounds__(kBlockThreads) void transcode_string(const mi_col_task* __restrict__ tasks,
                                                                  const uint32_t* __restrict__ tile_begin,
                                                                  const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                  uint32_t total_tiles, uint32_t* __restrict__ status) {
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE();
    __shared__ uint64_t s_valid[kTileRows / 64];
    if (tile_needs_mask(t)) __syncthreads();
    tile_validity(t, row0, n, s_valid);
    if (t.kind == MI_K_STR32) tile_string<int32_t>(t, row0, n, status, s_valid);
    else if (t.kind == MI_K_STR64) tile_string<int64_t>(t, row0, n, status, s_valid);
    else tile_fixed_binary(t, row0, n, s_valid);
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

__device__ __forceinline__ int misc_group_of(int32_t kind) {
  switch (kind) {
    case MI_K_BOOL: case MI_K_DICT: case MI_K_DATE64: case MI_K_MUL_I32: case MI_K_MUL_I64: case MI_K_DIV_I64: return 0;
    case MI_K_STRUCT: case MI_K_LIST32: case MI_K_LIST64: case MI_K_STRVIEW: return 1;
    default: return 2;
  }
}

// The common flat kinds (group 0) move 2-16 KB per tile: with 256-thread workgroups a CU holds 8 such
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
    if (misc_group_of(t.kind) != 0) continue;  // uniform: another group's launch owns this tile
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

// The kinds that keep 256-thread workgroups.  GROUP 1: list entries, string views, struct validity; GROUP 2: the rare
// flat kinds (intervals, durations, decimal32/64, half floats, null); the common flat kinds are transcode_misc_light's.
// Each group is its own kernel (own register budget) and a plan launches only the groups its tasks use.
template <int GROUP>
__global__ __launch_bounds__(kBlockThreads) void transcode_misc(const mi_col_task* __restrict__ tasks,
                                                                const uint32_t* __restrict__ tile_begin,
                                                                const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                uint32_t total_tiles, uint32_t* __restrict__ status) {
  static_assert(GROUP == 1 || GROUP == 2, "group 0 is transcode_misc_light");
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE();
    __shared__ uint64_t s_valid[kTileRows / 64];
    if (misc_group_of(t.kind) != GROUP) continue;  // uniform: another group's launch owns this tile
    if (tile_needs_mask(t)) __syncthreads();
    if (t.kind != MI_K_NULL) tile_validity(t, row0, n, s_valid);
    if (GROUP == 1) {
      switch (t.kind) {
        case MI_K_LIST32: tile_list<int32_t>(t, row0, n, status); break;
        case MI_K_LIST64: tile_list<int64_t>(t, row0, n, status); break;
        case MI_K_STRVIEW: tile_strview(t, row0, n, status, s_valid); break;
        default: break;  // MI_K_STRUCT: validity only
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

int ClassOfTask(const mi_col_task& t) {
  if (t.sel != nullptr) return KindCanGather(t.kind) ? kClassGather : -1;
  return ClassOfKind(t.kind);
}

int MiscGroupOfKind(int32_t kind) {
  switch (kind) {
    case MI_K_BOOL: case MI_K_DICT: case MI_K_DATE64: case MI_K_MUL_I32: case MI_K_MUL_I64: case MI_K_DIV_I64: return 0;
    case MI_K_STRUCT: case MI_K_LIST32: case MI_K_LIST64: case MI_K_STRVIEW: return 1;
    default: return 2;
  }
}

hipError_t LaunchTranscode(int cls, const mi_col_task* d_tasks, const uint32_t* d_tile_begin, const uint32_t* d_tile_task,
                           int32_t n_tasks, uint32_t total_tiles, uint32_t* d_status, uint32_t misc_groups, hipStream_t stream) {
  MI_DROP_STALE_ERROR();
  if (total_tiles == 0) return hipSuccess;
  if (d_tile_task == nullptr) return hipErrorInvalidValue;
  const dim3 grid(total_tiles), block(kBlockThreads);  // one workgroup per tile
#define MI_LAUNCH(KERNEL, BLOCK) \
  hipLaunchKernelGGL(KERNEL, grid, BLOCK, 0, stream, d_tasks, d_tile_begin, d_tile_task, n_tasks, total_tiles, d_status)
  switch (cls) {
    case kClassCopy: MI_LAUNCH(transcode_copy, block); break;
    case kClassDec128: MI_LAUNCH(transcode_dec128, block); break;
    case kClassString: MI_LAUNCH(transcode_string, block); break;
    case kClassMisc:
      if (misc_groups & 1u) MI_LAUNCH(transcode_misc_light, dim3(kLightThreads));
      if (misc_groups & 2u) MI_LAUNCH(transcode_misc<1>, block);
      if (misc_groups & 4u) MI_LAUNCH(transcode_misc<2>, block);
      break;
    default:
      return hipErrorInvalidValue;
  }
#undef MI_LAUNCH
  return hipGetLastError();
}

}  // namespace device
}  // namespace miarrow
