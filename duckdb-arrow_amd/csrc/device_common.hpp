// device_common.hpp -- shared device-side helpers of the gfx950 kernels (included by every kernels_*.hip).
//
// Everything on this path is HBM-bound integer / byte work ("no MFMA"): the design rules that matter are
// coalesced 16-byte-per-lane accesses, enough bytes in flight per CU, and a handful of launches over a device-resident
// task table for ANY number of record batches (a 122880-row record batch is ~40 MB of traffic = ~7 us at HBM speed, so
// per-column-per-batch launches would be launch-bound).
//
// One TILE = 2048 rows of one column of one record batch = exactly one DuckDB vector (STANDARD_VECTOR_SIZE), so tile t
// of a column writes vector t: data at out_data + t*2048*width, validity words at out_validity + t*32.  Tiles are
// numbered across all tasks of a kernel class and every tile gets its own workgroup (the hardware dispatcher balances
// variable-cost tiles better than a persistent grid did: -12 % time); `tile_task[tile]` names the
// task that owns a tile, so the descriptor is one scalar load away.  Neighbouring tiles share nothing but the cache
// lines at their seams and every byte is streamed exactly once, so blockIdx -> tile stays linear (no XCD remap).
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.hpp"

// hipGetLastError() after a launch reports the thread's LAST error, whoever caused it (another library's probing call, a
// failed call of an earlier request): launchers drop what is pending before they launch, so that what they return is
// about their own kernel.
#define MI_DROP_STALE_ERROR() (void)hipGetLastError()

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

// The owner of a tile is wave-uniform: one scalar load from the tile -> task table, the descriptor lands in SGPRs.
#define MI_TILE_PROLOGUE() MI_TILE_PROLOGUE_ROWS(kTileRows)
#define MI_TILE_PROLOGUE_ROWS(TILE_ROWS)                                                            \
  const int ti = __builtin_amdgcn_readfirstlane(static_cast<int>(tile_task[tile]));                 \
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

// ---- the same words WITHOUT the LDS round trip (the kernels of the common kinds).  tile_validity() above makes every data
// lane of the tile wait for bitmap load -> LDS store -> barrier before its own loads may leave: a second dependent round trip
// per tile on every column that has NULLs.  Below, the words a wave needs are requested together with its data loads.
//
// (a) wave-uniform word j of the tile (rows [row0 + 64 j, row0 + 64 j + 64) = the 64 rows a wave handles in one step of a
//     row-per-lane loop).  The Arrow bitmap is read through the constant address space with a uniform index: scalar loads
//     into SGPRs (the body is never written during a launch), a scalar funnel shift, no VGPRs, no barrier.
typedef const uint64_t __attribute__((address_space(4)))* kptr64;

__device__ __forceinline__ uint64_t bitmap_word(const mi_col_task& t, int64_t row0, int j) {
  kptr64 W = (kptr64)t.validity;
  const int64_t bit = t.row_offset + row0 + 64 * static_cast<int64_t>(j);
  const int64_t q = bit >> 6;
  const int sh = static_cast<int>(bit & 63);
  const int64_t last_q = (t.row_offset + t.nrows - 1) >> 6;  // last 8-byte word that holds a bit of this column
  const uint64_t lo = W[q];
  const uint64_t hi = (sh != 0 && q + 1 <= last_q) ? W[q + 1] : 0ull;
  return sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
}

// Word j (< ceil(n / 64), wave-uniform) of the tile's combined validity, pad bits canonical.  `lds` (non-NULL only for
// children of fixed-size lists, whose parent rows are r / list_size: tile_validity() has built the tile's words in LDS and
// stored them) overrides; a struct child's parent word is the same word of the parent's validity, written by an earlier
// launch of the same plan (tasks run in depth order).
__device__ __forceinline__ uint64_t tile_valid_word(const mi_col_task& t, int64_t row0, int n, int j, const uint64_t* lds) {
  if (lds != nullptr) return lds[j];
  uint64_t w = ~0ull;
  if (t.validity != nullptr && t.null_count != 0) w = bitmap_word(t, row0, j);
  if (t.out_aux != nullptr) w &= GC<uint64_t>(t.out_aux)[(row0 >> 6) + j];
  const int rem = n - 64 * j;
  if (rem < 64) w |= ~0ull << rem;
  return w;
}
// the tiles that need the LDS words (uniform)
__device__ __forceinline__ bool tile_words_from_lds(const mi_col_task& t) { return t.out_aux != nullptr && t.flags > 1; }

// (b) one lane per output word (lane < ceil(n / 64)), for kernels whose data lanes do not own whole 64-row groups: the loads
//     leave in lane_validity_begin -- before the tile's data loads -- and are only looked at in lane_validity_end, after the
//     data has been moved.
struct LaneValid {
  uint64_t lo, hi, parent;
};
__device__ __forceinline__ LaneValid lane_validity_begin(const mi_col_task& t, int64_t row0, int n) {
  LaneValid v{~0ull, 0ull, ~0ull};
  const int lane = threadIdx.x;
  if (t.out_validity == nullptr || lane >= ((n + 63) >> 6)) return v;
  if (t.validity != nullptr && t.null_count != 0) {
    gptr<const uint64_t> W = GC<uint64_t>(t.validity);
    const int64_t bit = t.row_offset + row0 + 64 * lane;
    const int64_t q = bit >> 6;
    const int64_t last_q = (t.row_offset + t.nrows - 1) >> 6;
    v.lo = W[q];
    v.hi = ((bit & 63) != 0 && q + 1 <= last_q) ? W[q + 1] : 0ull;
  }
  if (t.out_aux != nullptr && t.flags <= 1) v.parent = GC<uint64_t>(t.out_aux)[(row0 >> 6) + lane];
  return v;
}
__device__ __forceinline__ void lane_validity_end(const mi_col_task& t, int64_t row0, int n, const LaneValid& v) {
  const int lane = threadIdx.x;
  if (t.out_validity == nullptr || lane >= ((n + 63) >> 6)) return;
  uint64_t w = ~0ull;
  if (t.validity != nullptr && t.null_count != 0) {
    const int sh = static_cast<int>((t.row_offset + row0 + 64 * lane) & 63);
    w = sh ? ((v.lo >> sh) | (v.hi << (64 - sh))) : v.lo;
  }
  w &= v.parent;
  if (t.out_aux != nullptr && t.flags > 1) {
    gptr<const uint64_t> P = GC<uint64_t>(t.out_aux);
    const int64_t r = row0 + 64 * lane, div = t.flags;
    uint64_t pw = 0;
    for (int i = 0; i < 64; i++) {
      const int64_t pr = (r + i) / div;
      pw |= ((P[pr >> 6] >> (pr & 63)) & 1ull) << i;
    }
    w &= pw;
  }
  const int rem = n - 64 * lane;
  if (rem < 64) w |= ~0ull << rem;  // canonical pad bits
  GM<uint64_t>(t.out_validity)[(row0 >> 6) + lane] = w;
}

// Row validity for the data lanes: the combined tile mask in LDS (null_count == 0 and no parent => every row valid,
// the bitmap is not even read: GetValidityMask).
__device__ __forceinline__ bool row_valid(const uint64_t* s_valid, bool need_mask, int r) {
  return !need_mask || ((s_valid[r >> 6] >> (r & 63)) & 1);
}

// 16-byte vector whose loads may sit on any 4-byte boundary: gfx950 runs global memory in unaligned-access mode, so a
// global_load_dwordx4 from an 8-mod-16 address is legal; IPC buffers are only 8-byte aligned (after an odd-length
// offsets buffer every following buffer of the body is 8 mod 16), the destination vectors are always 16-byte aligned.
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef u32x4 u32x4_a1 __attribute__((aligned(1)));  // ... or on any byte (string payloads)

__device__ __forceinline__ u32x4 ld16(gptr<const u32x4_a4> p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st16(gptr<u32x4> p, u32x4 v) { __builtin_nontemporal_store(v, p); }

// ---------------------------------------------------------------------------------------------------- copies
// Coalesced copy of `bytes` bytes by one 256-thread workgroup.  The destination of every tile is 16-byte aligned (tiles
// start at multiples of 2048 rows of a 16-byte aligned vector); the source is an IPC buffer, only 8-byte aligned (after
// an odd-length offsets buffer every following buffer of the body is 8 mod 16), so it is read with dwordx4 loads in
// unaligned-access mode: 16 bytes per lane, 4 independent loads in flight before the first store, nontemporal both ways
// (streamed once).  Sources that are not even 4-byte aligned (narrow types at an odd Arrow array offset, only reachable
// through the kernel-level ABI) take the plain loops.
template <typename V>
__device__ __forceinline__ void copy_plain(gptr<const uint8_t> src, gptr<uint8_t> dst, int bytes) {
  const int nvec = bytes / static_cast<int>(sizeof(V));
  gptr<const V> s = (gptr<const V>)src;
  gptr<V> d = (gptr<V>)dst;
#pragma clang loop unroll(disable)
  for (int i = threadIdx.x; i < nvec; i += kBlockThreads) d[i] = s[i];
#pragma clang loop unroll(disable) vectorize(disable)
  for (int j = nvec * static_cast<int>(sizeof(V)) + threadIdx.x; j < bytes; j += kBlockThreads) dst[j] = src[j];
}

__device__ __forceinline__ void copy_bytes(gptr<const uint8_t> src, gptr<uint8_t> dst, int bytes) {
  const uintptr_t sa = reinterpret_cast<uintptr_t>(src), da = reinterpret_cast<uintptr_t>(dst);
  if ((da & 15) != 0 || (sa & 3) != 0) {  // wave-uniform
    if (((sa | da) & 3) == 0) copy_plain<uint32_t>(src, dst, bytes);
    else copy_plain<uint8_t>(src, dst, bytes);
    return;
  }
  const int nvec = bytes / 16;
  gptr<const u32x4_a4> s = (gptr<const u32x4_a4>)src;
  gptr<u32x4> d = (gptr<u32x4>)dst;
  int i = threadIdx.x;
#pragma clang loop unroll(disable)
  for (; i + 3 * kBlockThreads < nvec; i += 4 * kBlockThreads) {
    const u32x4 a = ld16(s + i), b = ld16(s + i + kBlockThreads), c = ld16(s + i + 2 * kBlockThreads), e = ld16(s + i + 3 * kBlockThreads);
    st16(d + i, a);
    st16(d + i + kBlockThreads, b);
    st16(d + i + 2 * kBlockThreads, c);
    st16(d + i + 3 * kBlockThreads, e);
  }
#pragma clang loop unroll(disable)
  for (; i < nvec; i += kBlockThreads) st16(d + i, ld16(s + i));
#pragma clang loop unroll(disable) vectorize(disable)
  for (int j = nvec * 16 + threadIdx.x; j < bytes; j += kBlockThreads) dst[j] = src[j];
}

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


}  // namespace
}  // namespace device
}  // namespace miarrow
