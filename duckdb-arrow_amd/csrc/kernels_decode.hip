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
    static_assert(kCopyTileRows / 64 <= kBlockThreads, "one lane per validity word");
    // the lanes that own a validity word ask for it, move their share of the data, and only then shift / pad / store the
    // word: one round trip instead of two (NULL rows keep their source bytes: DirectConversion)
    const LaneValid tv = lane_validity_begin(t, row0, n);
    const int w = static_cast<int>(t.param);
    copy_bytes(GC<uint8_t>(t.buf1) + (t.row_offset + row0) * w, GM<uint8_t>(t.out_data) + row0 * w, n * w);
    lane_validity_end(t, row0, n, tv);
  }
}

// ---------------------------------------------------------------------------------------------------- K3b
// decimal128 {u64 lower, i64 upper} -> int16/32/64 for valid rows (Hugeint::TryCast: value fits by precision);
// NULL rows canonical 0.  Each lane reads the whole 16-byte value (the upper half is what proves the range).
template <typename OUT, bool NULLS>
__device__ __forceinline__ void tile_dec128(const mi_col_task& t, int64_t row0, int n, uint32_t* status, const uint64_t* lds) {
  constexpr int R = kDecTileRows / kBlockThreads, G = 4;   // rows per lane; loads in flight per lane
  static_assert(R % G == 0, "whole groups");
  gptr<const uint8_t> src = GC<uint8_t>(t.buf1) + (t.row_offset + row0) * 16;
  gptr<OUT> out = GM<OUT>(t.out_data) + row0;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int nwords = (n + 63) >> 6;
  uint32_t err = 0;
#pragma clang loop unroll(disable)
  for (int k0 = 0; k0 < R; k0 += G) {
    if (k0 * kBlockThreads >= n) break;  // uniform
    // the wave's rows of step k are rows [64 j, 64 j + 64) of the tile, j = wave + 4 k: their validity word is wave-uniform and
    // is requested (scalar loads) together with the data, not in a round trip of its own in front of it
    u32x4 v[G];
#pragma unroll
    for (int k = 0; k < G; k++) {
      const int r = threadIdx.x + (k0 + k) * kBlockThreads;
      v[k] = u32x4{0u, 0u, 0u, 0u};
      if (r < n) v[k] = ld16((gptr<const u32x4_a4>)(src + 16 * static_cast<int64_t>(r)));  // 8-byte aligned source: unaligned-access mode
    }
    uint32_t okbits = ~0u;   // bit k: this lane's row of step k0 + k is valid (the words leave for the vector as soon as they are in)
    if (NULLS) {
#pragma clang loop unroll(disable)   // one word's SGPRs at a time: the scalar loads hide behind the vector loads in flight anyway
      for (int k = 0; k < G; k++) {
        const int j = wave + (kBlockThreads / 64) * (k0 + k);
        if (j < nwords) {  // uniform
          const uint64_t w = tile_valid_word(t, row0, n, j, lds);
          if (!((w >> lane) & 1ull)) okbits &= ~(1u << k);
          if (lane == 0 && t.out_validity != nullptr && lds == nullptr) GM<uint64_t>(t.out_validity)[(row0 >> 6) + j] = w;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < G; k++) {
      const int r = threadIdx.x + (k0 + k) * kBlockThreads;
      if (r < n) {
        const uint64_t lower = static_cast<uint64_t>(v[k].x) | (static_cast<uint64_t>(v[k].y) << 32);
        const int64_t upper = static_cast<int64_t>(static_cast<uint64_t>(v[k].z) | (static_cast<uint64_t>(v[k].w) << 32));
        OUT o = 0;
        if (!NULLS || ((okbits >> k) & 1u)) {
          o = static_cast<OUT>(lower);
          const int64_t sext = static_cast<int64_t>(o);
          if (static_cast<uint64_t>(sext) != lower || upper != (sext >> 63)) err = MI_ST_DECIMAL_RANGE;
        }
        __builtin_nontemporal_store(o, out + r);
      }
    }
  }
  // a column without NULLs: every word is all ones (pad bits included), one coalesced store behind the data
  if (!NULLS && t.out_validity != nullptr && static_cast<int>(threadIdx.x) < nwords) GM<uint64_t>(t.out_validity)[(row0 >> 6) + threadIdx.x] = ~0ull;
  raise(status, err);
}

// Two instances (see transcode_string): NULLS = false for the tiles of columns without NULLs, NULLS = true for the others.
// (amdgpu_num_sgpr(96): the scalar validity words would otherwise push the kernel over the 96 SGPRs that 8 waves per SIMD
// allow; the few values that do not fit live in VGPR lanes, of which there are plenty)
template <bool NULLS>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_num_sgpr(96))) void transcode_dec128(const mi_col_task* __restrict__ tasks,
                                                                  const uint32_t* __restrict__ tile_begin,
                                                                  const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                  uint32_t total_tiles, uint32_t* __restrict__ status) {
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE_ROWS(kDecTileRows);
    if (tile_needs_mask(t) != NULLS) continue;  // uniform: the other instance owns this tile
    __shared__ uint64_t s_valid[kDecTileRows / 64];
    const uint64_t* lds = nullptr;
    if (NULLS && tile_words_from_lds(t)) {  // child of a fixed-size list: the rare path keeps the LDS mask
      __syncthreads();
      tile_validity(t, row0, n, s_valid);
      lds = s_valid;
    }
    if (t.param == 8) tile_dec128<int64_t, NULLS>(t, row0, n, status, lds);
    else if (t.param == 4) tile_dec128<int32_t, NULLS>(t, row0, n, status, lds);
    else tile_dec128<int16_t, NULLS>(t, row0, n, status, lds);
  }
}

// ---------------------------------------------------------------------------------------------------- K4
// utf8 / binary with int32 or int64 offsets -> string_t, offsets validated like NANOARROW_VALIDATION_LEVEL_FULL.
// Every lane owns the 8 rows {tid + 256k} of the tile and issues all its loads before the first store (8 independent
// offset loads, then up to 8 x 4 payload dwords), so one wave has 8 rows in flight (-8 % time against 2 in flight).
// off[r+1] comes from the neighbouring lane (one permute) except at the wave edge and at the last row.  One 16-byte
// nontemporal store per row (1 KiB per wave store).
template <typename OFF, bool NULLS>
__device__ __forceinline__ void tile_string(const mi_col_task& t, int64_t row0, int n, uint32_t* status, const uint64_t* lds) {
  constexpr int R = kTileRows / kBlockThreads;
  gptr<const OFF> off = GC<OFF>(t.buf1) + t.row_offset + row0;
  gptr<const uint8_t> data = GC<uint8_t>(t.buf2);
  gptr<u32x4> out = GM<u32x4>(t.out_data) + row0;
  const int64_t data_len = t.buf2_len;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int nwords = (n + 63) >> 6;
  OFF a[R], b[R];
  // (the wave-edge loads of off[r+1] leave together with the others: asked for only once the neighbours' values are in --
  // they feed a shuffle -- they were a round trip of their own in front of the payload loads)
#pragma unroll
  for (int k = 0; k < R; k++) {
    const int r = threadIdx.x + k * kBlockThreads;
    const bool edge = lane == 63 || r == n - 1;
    a[k] = r < n ? off[r] : 0;
    if (sizeof(OFF) == 4) b[k] = (r < n && edge) ? off[r + 1] : 0;  // (int64 offsets: 16 more registers at this point, an occupancy step)
  }
  // NULLs (the kernel instance for tiles that have any).  The validity word of the wave's 64 rows of step k (rows [64 j,
  // 64 j + 64), j = wave + 4 k) is wave-uniform: scalar loads (tile_valid_word) issued here, behind the offset loads that are
  // already in flight -- the LDS mask they replace cost every lane of the tile a round trip (bitmap -> LDS -> barrier) in
  // front of its first load.  One bit per row stays in a VGPR; the words leave for the vector at once.
  uint32_t okbits = ~0u;
  if (NULLS) {
#pragma clang loop unroll(disable)   // one word's SGPRs at a time
    for (int k = 0; k < R; k++) {
      const int j = wave + (kBlockThreads / 64) * k;
      if (j < nwords) {  // uniform
        const uint64_t w = tile_valid_word(t, row0, n, j, lds);
        if (!((w >> lane) & 1ull)) okbits &= ~(1u << k);
        if (lane == 0 && t.out_validity != nullptr && lds == nullptr) GM<uint64_t>(t.out_validity)[(row0 >> 6) + j] = w;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < R; k++) {
    const int r = threadIdx.x + k * kBlockThreads;
    const OFF from_neighbour = __shfl_down(a[k], 1, 64);
    const bool edge = lane == 63 || r == n - 1;
    if (sizeof(OFF) == 4) {
      if (!(r < n && edge)) b[k] = from_neighbour;
    } else {
      b[k] = (r < n && edge) ? off[r + 1] : from_neighbour;
    }
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
      } else if (!NULLS || ((okbits >> k) & 1u)) {
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
  // a column without NULLs: every word is all ones (pad bits included), one coalesced store behind the data
  if (!NULLS && t.out_validity != nullptr && static_cast<int>(threadIdx.x) < nwords) GM<uint64_t>(t.out_validity)[(row0 >> 6) + threadIdx.x] = ~0ull;
  raise(status, err);
}

// fixed_size_binary(width) -> string_t
__device__ __forceinline__ void tile_fixed_binary(const mi_col_task& t, int64_t row0, int n, const uint64_t* lds) {
  gptr<const uint8_t> data = GC<uint8_t>(t.buf1);
  gptr<u32x4> out = GM<u32x4>(t.out_data) + row0;
  const bool need_words = tile_needs_mask(t) || t.out_validity != nullptr;   // (the rare kind: words always through the wave path)
  const int64_t width = t.param;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int nwords = (n + 63) >> 6;
#pragma unroll 2
  for (int k = 0; k < kTileRows / kBlockThreads; k++) {
    if (k * kBlockThreads >= n) break;  // uniform
    const int r = threadIdx.x + k * kBlockThreads;
    const int j = wave + (kBlockThreads / 64) * k;
    const uint64_t vw = (need_words && j < nwords) ? tile_valid_word(t, row0, n, j, lds) : ~0ull;
    if (r < n) {
      u32x4 s = {0u, 0u, 0u, 0u};
      if ((vw >> lane) & 1ull) s = make_string_t(data, (t.row_offset + row0 + r) * width, static_cast<uint32_t>(width), t.ptr_base);
      out[r] = s;
    }
    if (lane == 0 && j < nwords && t.out_validity != nullptr && lds == nullptr) GM<uint64_t>(t.out_validity)[(row0 >> 6) + j] = vw;
  }
}

// Two instances, each launched only when the plan has tiles for it: NULLS = false takes the tiles of columns without NULLs
// (no bitmap is read, the validity words are all ones) and keeps the register budget of the plain loop -- the headline
// workload runs this one alone; NULLS = true takes the tiles that need a mask.  A tile of the other instance returns at once.
template <bool NULLS>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_num_sgpr(96))) void transcode_string(const mi_col_task* __restrict__ tasks,
                                                                  const uint32_t* __restrict__ tile_begin,
                                                                  const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                  uint32_t total_tiles, uint32_t* __restrict__ status) {
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE();
    if (tile_needs_mask(t) != NULLS) continue;  // uniform: the other instance owns this tile
    __shared__ uint64_t s_valid[kTileRows / 64];
    const uint64_t* lds = nullptr;
    if (NULLS && tile_words_from_lds(t)) {  // child of a fixed-size list: the rare path keeps the LDS mask
      __syncthreads();
      tile_validity(t, row0, n, s_valid);
      lds = s_valid;
    }
    if (t.kind == MI_K_STR32) tile_string<int32_t, NULLS>(t, row0, n, status, lds);
    else if (t.kind == MI_K_STR64) tile_string<int64_t, NULLS>(t, row0, n, status, lds);
    else tile_fixed_binary(t, row0, n, lds);
  }
}

// ---------------------------------------------------------------------------------------------------- K2, K3c, K5
// bool bit -> byte, date64 -> date32, time / timestamp unit casts and dictionary indices -> sel_t are the common flat kinds:
// they live in transcode_misc_light below (one wave per tile).  The rare kinds follow.
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

__device__ __forceinline__ int misc_group_of(int32_t kind) {
  switch (kind) {
    case MI_K_BOOL: case MI_K_DICT: case MI_K_MUL_I32: return 0;
    case MI_K_DATE64: case MI_K_MUL_I64: case MI_K_DIV_I64: return 3;
    case MI_K_STRUCT: case MI_K_LIST32: case MI_K_LIST64: case MI_K_STRVIEW: return 1;
    default: return 2;
  }
}

// The common flat kinds (group 0) move 2-16 KB per tile: with 256-thread workgroups a CU holds 8 such tiles and each is
// one chain of dependent round trips (task lookup, bitmap, data), which left the kernel latency bound (1.8-2.3 TB/s).
// Here ONE WAVE owns a tile -- 32 independent tiles per CU, no workgroup barriers -- and a lane moves 4 consecutive rows
// per access (16-byte loads and stores), with every load of the tile (or of half of it, for 8-byte inputs) issued before
// the first value is used: a tile is one or two round trips instead of the eight a row-per-lane loop with 4 loads in
// flight needed (3.0 TB/s on config 5's bool + dictionary columns, profiles/r01_final).
constexpr int kLightThreads = 64;

// out[r] = f(r, src[r]) for r < n <= 2048 by one wave; GROUP passes of 64 x V rows are in flight at a time.
// mask(base) = validity bits of rows [base, base + V) (bit k = row base + k), requested with the data loads of its pass.
template <typename IN, typename OUT, int V, int GROUP, typename M, typename F>
__device__ __forceinline__ void light_map(gptr<const IN> src, gptr<OUT> out, int n, M&& mask, F&& f) {
  typedef IN in_vec __attribute__((ext_vector_type(V)));
  typedef in_vec in_vec_a __attribute__((aligned(sizeof(IN))));                 // Arrow buffers: element aligned only
  typedef OUT out_vec __attribute__((ext_vector_type(V)));
  typedef out_vec out_vec_a __attribute__((aligned(V * sizeof(OUT) < 16 ? V * sizeof(OUT) : 16)));  // tiles start 16-byte aligned
  constexpr int PASSES = kTileRows / (kLightThreads * V);
  static_assert(PASSES % GROUP == 0, "whole groups");
  const int lane = threadIdx.x;
#pragma unroll
  for (int g = 0; g < PASSES; g += GROUP) {
    if ((g * kLightThreads) * V >= n) break;  // uniform
    in_vec vin[GROUP];
    uint32_t ok[GROUP];
#pragma unroll
    for (int p = 0; p < GROUP; p++) {
      const int base = ((g + p) * kLightThreads + lane) * V;
      ok[p] = mask(base);   // by every lane, also past the tile's end: the bits come from another lane's register (ds_bpermute)
      if (base + V <= n) {
        vin[p] = __builtin_nontemporal_load((gptr<const in_vec_a>)(src + base));
      } else {
#pragma unroll
        for (int k = 0; k < V; k++) vin[p][k] = base + k < n ? src[base + k] : IN(0);
      }
    }
#pragma unroll
    for (int p = 0; p < GROUP; p++) {
      const int base = ((g + p) * kLightThreads + lane) * V;
      out_vec vout;
#pragma unroll
      for (int k = 0; k < V; k++) vout[k] = f(((ok[p] >> k) & 1u) != 0, vin[p][k]);
      if (base + V <= n) {
        __builtin_nontemporal_store(vout, (gptr<out_vec_a>)(out + base));
      } else {
#pragma unroll
        for (int k = 0; k < V; k++)
          if (base + k < n) out[base + k] = vout[k];
      }
    }
  }
}

// Bit-packed bool -> one byte per row: lane i expands rows [32 i, 32 i + 32) (4 bytes in, two 16-byte stores out)
__device__ __forceinline__ void light_bool(const mi_col_task& t, int64_t row0, int n) {
  gptr<const uint8_t> bits = GC<uint8_t>(t.buf1);
  gptr<uint8_t> out = GM<uint8_t>(t.out_data) + row0;
  const int64_t last_byte = (t.row_offset + t.nrows - 1) >> 3;
  const int r = 32 * static_cast<int>(threadIdx.x);
  if (r >= n) return;
  const int64_t bit = t.row_offset + row0 + r;
  const int64_t byte = bit >> 3;
  const int sh = static_cast<int>(bit & 7);
  uint64_t raw = 0;  // up to 5 source bytes
#pragma unroll
  for (int k = 0; k < 5; k++)
    if (byte + k <= last_byte) raw |= static_cast<uint64_t>(bits[byte + k]) << (8 * k);
  const uint32_t b32 = static_cast<uint32_t>(raw >> sh);
  uint64_t y[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint64_t b = (b32 >> (8 * k)) & 0xFFu;
    uint64_t v = (b * 0x0101010101010101ull) & 0x8040201008040201ull;
    y[k] = ((v + 0x7F7F7F7F7F7F7F7Full) >> 7) & 0x0101010101010101ull;
  }
  if (r + 32 <= n) {
    __builtin_nontemporal_store(u32x4{static_cast<uint32_t>(y[0]), static_cast<uint32_t>(y[0] >> 32), static_cast<uint32_t>(y[1]), static_cast<uint32_t>(y[1] >> 32)},
                                (gptr<u32x4>)(out + r));
    __builtin_nontemporal_store(u32x4{static_cast<uint32_t>(y[2]), static_cast<uint32_t>(y[2] >> 32), static_cast<uint32_t>(y[3]), static_cast<uint32_t>(y[3] >> 32)},
                                (gptr<u32x4>)(out + r + 16));
  } else {
    for (int k = 0; r + k < n; k++) out[r + k] = static_cast<uint8_t>(y[k >> 3] >> (8 * (k & 7)));
  }
}

// WIDE = false: bool, dictionary indices, int32 -> int64 casts; true: the 8-byte inputs (date64, timestamp / time unit
// casts), whose 64-bit multiply-high arithmetic wants a register budget of its own
template <bool WIDE>
__global__ __launch_bounds__(kLightThreads) void transcode_misc_light(const mi_col_task* __restrict__ tasks,
                                                                      const uint32_t* __restrict__ tile_begin,
                                                                      const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                      uint32_t total_tiles, uint32_t* __restrict__ status) {
  __shared__ uint64_t s_valid[kTileRows / 64];
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE();
    if (misc_group_of(t.kind) != (WIDE ? 3 : 0)) continue;  // uniform: another group's launch owns this tile
    // NULLs of the column itself: lane L asks for dword L of the tile's 2048 validity bits BEFORE the data loads leave; a data
    // lane gets the bits of its 4 rows from the lane that holds them (one ds_bpermute, no memory access), and the same dwords
    // are the tile's validity words, stored behind the data -- the tile stays ONE round trip.  Children of structs /
    // fixed-size lists (the parent's NULLs come on top) keep the LDS mask.
    const bool nested = t.out_aux != nullptr;
    const bool has_nulls = tile_needs_mask(t);
    const int lane = threadIdx.x;
    uint32_t vd = ~0u;
    if (nested) {
      __syncthreads();
      tile_validity<kLightThreads>(t, row0, n, s_valid);
    } else if (has_nulls && 32 * lane < n) {
      gptr<const uint32_t> W = GC<uint32_t>(t.validity);
      const int64_t bit = t.row_offset + row0 + 32 * lane;
      const int64_t q = bit >> 5;
      const int sh = static_cast<int>(bit & 31);
      const int64_t last_q = (t.row_offset + t.nrows - 1) >> 5;   // last dword that holds a bit of this column
      const uint32_t lo = W[q];
      const uint32_t hi = (sh != 0 && q + 1 <= last_q) ? W[q + 1] : 0u;
      vd = sh ? __builtin_amdgcn_alignbit(hi, lo, static_cast<uint32_t>(sh)) : lo;
    }
    auto valid4 = [&](int base) -> uint32_t {   // rows base .. base + 3 (base is a multiple of 4: inside one dword)
      if (!has_nulls) return 0xFu;
      if (nested) return static_cast<uint32_t>(s_valid[base >> 6] >> (base & 63)) & 0xFu;
      return (static_cast<uint32_t>(__shfl(static_cast<int>(vd), base >> 5, 64)) >> (base & 31)) & 0xFu;
    };
    uint32_t err = 0;
    switch (WIDE ? t.kind : 0) {
      case MI_K_DATE64:
        light_map<int64_t, int32_t, 4, 4>(GC<int64_t>(t.buf1) + t.row_offset + row0, GM<int32_t>(t.out_data) + row0, n, [](int) { return 0xFu; },
                                          [&](bool, int64_t v) { return static_cast<int32_t>(v / 86400000ll); });
        break;
      case MI_K_MUL_I64: {
        const int64_t factor = t.param;
        light_map<int64_t, int64_t, 4, 4>(GC<int64_t>(t.buf1) + t.row_offset + row0, GM<int64_t>(t.out_data) + row0, n, valid4, [&](bool ok, int64_t v) {
          int64_t o = 0;
          if (ok && __builtin_mul_overflow(v, factor, &o)) {  // TryMultiplyOperator => ConversionException
            o = 0;
            err = MI_ST_MUL_OVERFLOW;
          }
          return o;
        });
        break;
      }
      case MI_K_DIV_I64: {
        // the divisors the type mapping produces are 1000 (ns -> us) and powers of it: a constant division is a
        // multiply-high, the generic 64-bit division (~100 instructions, dozens of registers) stays out of the unrolled code
        const int64_t d = t.param;
        if (d == 1000) {
          light_map<int64_t, int64_t, 4, 4>(GC<int64_t>(t.buf1) + t.row_offset + row0, GM<int64_t>(t.out_data) + row0, n, [](int) { return 0xFu; },
                                            [&](bool, int64_t v) { return v / 1000; });  // all rows, like upstream
        } else {
          gptr<const int64_t> src = GC<int64_t>(t.buf1) + t.row_offset + row0;
          gptr<int64_t> out = GM<int64_t>(t.out_data) + row0;
#pragma unroll 1
          for (int r = threadIdx.x; r < n; r += kLightThreads) out[r] = src[r] / d;
        }
        break;
      }
      default: break;
    }
    switch (WIDE ? 0 : t.kind) {
      case MI_K_BOOL: light_bool(t, row0, n); break;
      case MI_K_MUL_I32: {
        const int64_t factor = t.param;  // int32 * 1e6 cannot overflow int64
        light_map<int32_t, int64_t, 4, 4>(GC<int32_t>(t.buf1) + t.row_offset + row0, GM<int64_t>(t.out_data) + row0, n, valid4,
                                          [&](bool ok, int32_t v) { return ok ? static_cast<int64_t>(v) * factor : int64_t(0); });
        break;
      }
      case MI_K_DICT: {
        // indices -> sel_t; NULL -> dict_len (the extra NULL slot of the decoded dictionary)
        const int iw = static_cast<int>(t.param & 0xFF);
        const bool is_signed = ((t.param >> 8) & 1) != 0;
        const uint32_t dict_len = static_cast<uint32_t>(t.param2);
        auto to_sel = [&](bool ok, uint64_t v) {
          if (!ok) return dict_len;
          if (v > 0xFFFFFFFFull) {  // "DuckDB only supports indices that fit on an uint32"
            err = MI_ST_INDEX_RANGE;
            return dict_len;
          }
          if (v >= dict_len) {  // the selection vector must never point past the dictionary's NULL slot
            err = MI_ST_DICT_INDEX;
            return dict_len;
          }
          return static_cast<uint32_t>(v);
        };
        gptr<uint32_t> out = GM<uint32_t>(t.out_data) + row0;
        gptr<const uint8_t> idx = GC<uint8_t>(t.buf1) + (t.row_offset + row0) * iw;
        if (iw == 4) {
          if (is_signed) light_map<int32_t, uint32_t, 4, 8>((gptr<const int32_t>)idx, out, n, valid4, [&](bool ok, int32_t v) { return to_sel(ok, static_cast<uint64_t>(static_cast<int64_t>(v))); });
          else light_map<uint32_t, uint32_t, 4, 8>((gptr<const uint32_t>)idx, out, n, valid4, [&](bool ok, uint32_t v) { return to_sel(ok, v); });
        } else if (iw == 2) {
          if (is_signed) light_map<int16_t, uint32_t, 4, 8>((gptr<const int16_t>)idx, out, n, valid4, [&](bool ok, int16_t v) { return to_sel(ok, static_cast<uint64_t>(static_cast<int64_t>(v))); });
          else light_map<uint16_t, uint32_t, 4, 8>((gptr<const uint16_t>)idx, out, n, valid4, [&](bool ok, uint16_t v) { return to_sel(ok, v); });
        } else if (iw == 1) {
          if (is_signed) light_map<int8_t, uint32_t, 4, 8>((gptr<const int8_t>)idx, out, n, valid4, [&](bool ok, int8_t v) { return to_sel(ok, static_cast<uint64_t>(static_cast<int64_t>(v))); });
          else light_map<uint8_t, uint32_t, 4, 8>(idx, out, n, valid4, [&](bool ok, uint8_t v) { return to_sel(ok, v); });
        } else {
          light_map<uint64_t, uint32_t, 4, 4>((gptr<const uint64_t>)idx, out, n, valid4, [&](bool ok, uint64_t v) { return to_sel(ok, v); });  // a negative int64 is > UINT32_MAX too
        }
        break;
      }
      default: break;
    }
    if (!nested && t.out_validity != nullptr && 32 * lane < ((n + 63) >> 6) * 64) {
      const int rem = n - 32 * lane;   // canonical pad bits
      const uint32_t d = rem >= 32 ? vd : rem <= 0 ? ~0u : (vd | (~0u << rem));
      GM<uint32_t>(t.out_validity)[(row0 >> 5) + lane] = d;
    }
    raise(status, err);
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
  static_assert(GROUP == 1 || GROUP == 2, "groups 0 and 3 are transcode_misc_light");
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
    case MI_K_BOOL: case MI_K_DICT: case MI_K_MUL_I32: return 0;
    case MI_K_DATE64: case MI_K_MUL_I64: case MI_K_DIV_I64: return 3;
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
    case kClassDec128:   // misc_groups: bit 0 = tiles without NULLs, bit 1 = tiles that need a mask
      if (misc_groups & 1u) MI_LAUNCH(transcode_dec128<false>, block);
      if (misc_groups & 2u) MI_LAUNCH(transcode_dec128<true>, block);
      break;
    case kClassString:
      if (misc_groups & 1u) MI_LAUNCH(transcode_string<false>, block);
      if (misc_groups & 2u) MI_LAUNCH(transcode_string<true>, block);
      break;
    case kClassMisc:
      if (misc_groups & 1u) MI_LAUNCH(transcode_misc_light<false>, dim3(kLightThreads));
      if (misc_groups & 8u) MI_LAUNCH(transcode_misc_light<true>, dim3(kLightThreads));
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
