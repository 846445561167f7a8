// kernels_encode.hip -- K7: DuckDB vectors -> Arrow buffers (COPY TO / to_arrow_ipc), ArrowAppender semantics
// (call site in the reference: src/writer/column_data_collection_serializer.cpp:85).
#include "device_common.hpp"

#include <algorithm>

namespace miarrow {
namespace device {

namespace {

// ==================================================================================================== K7 (encode)
// DuckDB vectors -> Arrow buffers, ArrowAppender semantics.  Task fields for encode kinds:
//   validity  = DuckDB validity words of the whole column (NULL = all valid)   buf1 = vector data
//   buf2      = string heap base (long string_t pointers are ptr - ptr_base into it)
//   out_validity = Arrow bitmap (ceil(n/8) bytes, always emitted, pad bits 1)  out_data = Arrow buffer 1
//   out_aux   = Arrow buffer 2 (string data)          param2 = index of this task's null counter

// K7a: DuckDB validity words have Arrow's bit order and polarity, so the bitmap is a byte copy of the words with
// the pad bits of the last byte forced to 1 (ResizeValidity fills with 0xFF) and NULLs counted on the way.
// In two halves, so that a kernel can put its own loads between the request for the validity word and its use.
struct EncValidity {
  uint64_t w;
  bool active;
};
__device__ __forceinline__ EncValidity enc_tile_validity_load(const mi_col_task& t, int64_t row0, int n, const uint64_t* s_valid = nullptr) {
  EncValidity v{~0ull, false};
  if (threadIdx.x >= 64 || (t.out_validity == nullptr && s_valid == nullptr)) return v;  // wave 0, uniform
  const int lane = threadIdx.x;
  v.active = lane < ((n + 63) >> 6);
  if (v.active && t.validity != nullptr) v.w = GC<uint64_t>(t.validity)[(row0 >> 6) + lane];
  return v;
}
__device__ __forceinline__ void enc_tile_validity_finish(const mi_col_task& t, int64_t row0, int n, int64_t* null_counts, EncValidity v,
                                                         uint64_t* s_valid = nullptr) {
  if (threadIdx.x >= 64 || (t.out_validity == nullptr && s_valid == nullptr)) return;  // wave 0, uniform
  const int lane = threadIdx.x;
  const bool active = v.active;
  uint64_t w = v.w;
  const int rem = n - 64 * lane;
  if (active) {
    if (rem < 64) w |= ~0ull << rem;
    if (s_valid) s_valid[lane] = w;
  }
  if (t.out_validity == nullptr) return;
  // one counter update per tile: the counter of a column is ONE address for all of its tiles
  int nulls = active ? 64 - __builtin_popcountll(w) : 0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) nulls += __shfl_down(nulls, d, 64);
  if (lane == 0 && nulls) atomicAdd(reinterpret_cast<unsigned long long*>(null_counts + t.param2), static_cast<unsigned long long>(nulls));
  if (!active) return;
  gptr<uint8_t> out = GM<uint8_t>(t.out_validity) + (row0 >> 3) + 8 * lane;
  const int nbytes = rem >= 64 ? 8 : (rem + 7) >> 3;
  if (nbytes == 8 && (reinterpret_cast<uintptr_t>(out) & 7) == 0) {
    *(gptr<uint64_t>)out = w;
  } else {
    for (int k = 0; k < nbytes; k++) out[k] = static_cast<uint8_t>(w >> (8 * k));
  }
}
__device__ __forceinline__ void enc_tile_validity(const mi_col_task& t, int64_t row0, int n, int64_t* null_counts,
                                                  uint64_t* s_valid = nullptr) {
  enc_tile_validity_finish(t, row0, n, null_counts, enc_tile_validity_load(t, row0, n, s_valid), s_valid);
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
    // wave 0 asks for the tile's validity words, moves its share of the data, and only then turns the words into bitmap
    // bytes and a NULL count: one round trip for the wave instead of two
    const EncValidity tv = enc_tile_validity_load(t, row0, n);
    switch (t.kind) {
      case MI_K_ENC_COPY: {  // NULL slots copy whatever the source slot holds, like ArrowScalarData::Append
        const int w = static_cast<int>(t.param);
        copy_bytes(GC<uint8_t>(t.buf1) + row0 * w, GM<uint8_t>(t.out_data) + row0 * w, n * w);
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
    enc_tile_validity_finish(t, row0, n, null_counts, tv);
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

constexpr int kEncStage = 16 * 1024;  // bytes of payload staged per 256-row sub-block (16 KB x 8 workgroups per CU)

// One tile with 64-bit positions: sub-block by sub-block (256 rows), byte-wise LDS assembly, a sub-block whose payload
// exceeds the stage falls back to direct byte stores.  The slow, exact path of encode_string_1p for tiles that hold a string
// of >= 8 MiB.
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


// cnt >= 1 bytes of a heap string, src -> dst (LDS), both at any byte alignment: gfx950 takes unaligned 16- / 8- / 4-byte
// accesses in global memory and in LDS alike, so a string travels in pieces of 16 bytes whose last piece ends where the
// string ends and overlaps the one before it (the same bytes written twice) -- no funnel shifts, no head or tail bytes, and
// never a byte read outside [src, src + cnt).  Up to 64 bytes are four loads in flight, those a shorter string does not
// need predicated off; longer strings go on 16 bytes at a time.  (The formulation before this one covered 48 bytes with 13
// aligned dword loads, 13 v_alignbyte_b32 and as many predicated dword stores per row: the kernel is bound by instruction
// issue, and on DuckDB's own string heaps -- long strings back to back, no gaps for the inline ones, so no wave-wide copy --
// every long row took it.)
typedef u32x2 u32x2_a1 __attribute__((aligned(1)));
typedef uint32_t u32_a1 __attribute__((aligned(1)));
__device__ __forceinline__ void heap_bytes_to_lds(uint8_t* dst, gptr<const uint8_t> src, uint32_t cnt) {
  if (cnt >= 16) {
    const uint32_t last = cnt - 16;
    const bool hb = last > 16, hc = last > 32;
    const u32x4 a = __builtin_nontemporal_load((gptr<const u32x4_a1>)src);
    const u32x4 e = __builtin_nontemporal_load((gptr<const u32x4_a1>)(src + last));
    u32x4 b = a, c = a;
    if (hb) b = __builtin_nontemporal_load((gptr<const u32x4_a1>)(src + 16));
    if (hc) c = __builtin_nontemporal_load((gptr<const u32x4_a1>)(src + 32));
    *reinterpret_cast<u32x4_a1*>(dst) = a;
    if (hb) *reinterpret_cast<u32x4_a1*>(dst + 16) = b;
    if (hc) *reinterpret_cast<u32x4_a1*>(dst + 32) = c;
    *reinterpret_cast<u32x4_a1*>(dst + last) = e;
#pragma clang loop unroll(disable)
    for (uint32_t done = 48; done < last; done += 16)
      *reinterpret_cast<u32x4_a1*>(dst + done) = __builtin_nontemporal_load((gptr<const u32x4_a1>)(src + done));
  } else if (cnt >= 8) {
    const u32x2 a = *(gptr<const u32x2_a1>)src, e = *(gptr<const u32x2_a1>)(src + (cnt - 8));
    *reinterpret_cast<u32x2_a1*>(dst) = a;
    *reinterpret_cast<u32x2_a1*>(dst + (cnt - 8)) = e;
  } else if (cnt >= 4) {
    const uint32_t a = *(gptr<const u32_a1>)src, e = *(gptr<const u32_a1>)(src + (cnt - 4));
    *reinterpret_cast<u32_a1*>(dst) = a;
    *reinterpret_cast<u32_a1*>(dst + (cnt - 4)) = e;
  } else {
    const uint8_t a = src[0], m = src[cnt >> 1], e = src[cnt - 1];   // 1: a a a; 2: a e e; 3: a m e
    dst[0] = a;
    dst[cnt >> 1] = m;
    dst[cnt - 1] = e;
  }
}

// Bytes [c0, c0 + cnt) of one string -> LDS at dst: an inline string from its registers, a long one from the heap.
__device__ __forceinline__ void string_bytes_to_lds(uint8_t* dst, const u32x4& s, gptr<const uint8_t> heap, uint64_t ptr_base,
                                                    uint32_t c0, uint32_t cnt) {
  if (s.x <= 12) {
    uint32_t a = s.y, b = s.z, c = s.w;
    if (c0 >= 8) { a = c; b = 0u; c = 0u; }
    else if (c0 >= 4) { a = b; b = c; c = 0u; }
    uint32_t S[5] = {a, b, c, 0u, 0u};
    lds_put_stream<3>(dst, S, c0 & 3u, cnt);
    return;
  }
  const uint64_t p = static_cast<uint64_t>(s.z) | (static_cast<uint64_t>(s.w) << 32);
  heap_bytes_to_lds(dst, heap + (p - ptr_base) + c0, cnt);
}

// Stage bytes [shiftw, end) -> gbase[shiftw, end) by the whole workgroup (end > shiftw); gbase is 16-byte aligned:
// coalesced 16-byte nontemporal stores, the (at most two) partial 16-byte rows one byte per lane.
__device__ __forceinline__ void stage_to_data(const uint8_t* st, uint32_t shiftw, uint32_t end, gptr<uint8_t> gbase) {
  const uint32_t nch = (end + 15) >> 4;
  for (uint32_t c = threadIdx.x; c < nch; c += kBlockThreads) {
    const uint32_t clo = c << 4;
    if (clo >= shiftw && clo + 16 <= end)
      __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(st + clo), (gptr<u32x4>)(gbase + clo));
  }
  if (threadIdx.x < 32) {
    const uint32_t lastlo = (end - 1) & ~15u;
    const uint32_t idx = threadIdx.x < 16 ? threadIdx.x : lastlo + (threadIdx.x - 16);
    const bool first_partial = shiftw != 0 || end < 16;
    const bool last_partial = (end & 15u) != 0 && lastlo != 0;
    const bool mine = threadIdx.x < 16 ? first_partial : last_partial;
    if (mine && idx >= shiftw && idx < end) gbase[idx] = st[idx];
  }
}

// ---------------------------------------------------------------------------------------------------- K7d, single pass
// string_t rows (or list_entry_t rows: offsets only) -> Arrow offsets + data in ONE pass over HBM: every tile first adds up
// its own lengths (a pass over the length fields, whose cache lines the main loop then finds in L2), publishes the sum and
// picks up the sum of the tiles before it in the same column (decoupled look-back, Merrill & Garland), so the string_t rows
// come from HBM exactly once and a column costs one launch.  The first formulation ran three launches (per-tile sums, a scan
// over them, the encode) and read every string_t twice: at 21..47 B/row of algorithmic traffic those 16 B/row were why its
// 4.2 TB/s of real traffic showed as 3.2 TB/s algorithmic.
//
// The encode of a tile is the loop the counters shaped (rocprofv3 SQ_INSTS_* in profiles/r01_encode): 256-row sub-blocks,
// lane = row (coalesced 16-byte string_t loads, coalesced offset stores); the NEXT sub-block's string_t is requested before
// this one is touched; a 6-step DPP wave scan + the group totals of the length pass place every row; the payload is
// assembled in LDS in windows of <= 8 KiB on two alternating stage buffers (one barrier per window) and leaves as coalesced
// 16-byte nontemporal stores.  How the bytes reach LDS is decided per wave: when its long strings lie in the heap as they
// will lie in the data buffer (vectors decoded from Arrow buffers, staged heaps), the span from the first to the last of
// them is ONE coalesced copy and only the inline strings place themselves; otherwise every long row brings its own bytes
// in unaligned 16-byte pieces (heap_bytes_to_lds).  The kernel keeps to 64 VGPRs and 17 KiB of LDS: 8 workgroups per CU.
//
// Look-back words: tile_state[tile] bits 62..63 = 0 nothing yet, 1 = sum of this tile, 2 = sum of every tile of the column up
// to and including this one.  They are read and written with RELAXED agent-scope atomics: the word is the whole message, and
// an acquire / release pair would make every tile write back and invalidate its XCD's L2 (measured: 12x slower).
// Tile = workgroup id.  A tile waits only for tiles with a lower id, and the dispatcher hands workgroups out in id order
// (on MI355X round-robin over the 8 XCDs, each XCD starting its share in order): the lowest unfinished tile is therefore
// always running, so the walk cannot deadlock.  (A ticket counter -- tiles numbered in the order they start -- bought the
// same without that argument but cost one same-address far atomic per tile: 3.95 ms instead of 3.64 ms for SF10.)  The spin
// is bounded all the same: a logic error ends in MI_ST_INTERNAL, not in a hung device.  tile_state[total_tiles] is unused.
// tile_state[total_tiles + 1 + tile] = 0, or (1 << 63 | first output byte) of a tile left to encode_string_slow: list offsets
// (no payload) and tiles holding a string of >= 8 MiB (32-bit positions inside a sub-block could wrap).
constexpr uint32_t kEncBigLen = 1u << 23;
constexpr int kEncStage5 = 8 * 1024;       // bytes per stage buffer: a 256-row sub-block of lineitem comments (6.8 KB) is one window
constexpr int kEncStageBuf = kEncStage5 + 64;
constexpr uint64_t kStateMask = (1ull << 62) - 1ull;
constexpr int kLookBack = 4;   // predecessors inspected per look-back step

__global__ __launch_bounds__(kBlockThreads, 8) void encode_string_1p(const mi_col_task* __restrict__ tasks,
                                                                     const uint32_t* __restrict__ tile_begin,
                                                                     const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                     uint32_t total_tiles, unsigned long long* __restrict__ tile_state,
                                                                     int64_t* __restrict__ null_counts, uint32_t* __restrict__ status) {
  constexpr int kWaves = kBlockThreads / 64;
  static_assert(kWaves == 4, "wave totals travel as one 16-byte LDS row");
  __shared__ uint64_t s_valid[kTileRows / 64];
  __shared__ uint32_t s_wtot[kTileRows / 64];         // bytes of every (sub-block, wave) group: 32 of them
  __shared__ uint32_t s_wbase[kTileRows / 64 + 1];     // ... and their exclusive prefix (+ the tile total)
  __shared__ unsigned long long s_sum[kWaves];
  __shared__ uint32_t s_tiny[kWaves];
  __shared__ int64_t s_prefix;
  __shared__ __attribute__((aligned(16))) uint8_t stage[2 * kEncStageBuf];
  (void)n_tasks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t tile = blockIdx.x;
  if (tile >= total_tiles) return;
  MI_TILE_PROLOGUE();
  const bool is_list = t.kind == MI_K_ENC_LIST32;
  const bool large = (t.flags & 1) != 0;  // LargeUtf8 / LargeList: int64 offsets (arrow_large_buffer_size)
  // The length pass reads two dwords per row (the length field: dword 0 of a string_t, the low dword of a list_entry_t's
  // length; and the dword behind it: a string_t's first four bytes, all there is to a string of <= 4 bytes -- a tile of
  // flags or codes is encoded below from these registers alone).  The loads leave before the validity words are even
  // asked for: the length of a NULL row is read and then ignored.
  uint32_t pre_len[kTileRows / kBlockThreads], pre_bytes[kTileRows / kBlockThreads];
  {
    gptr<const u32x2> lens = (gptr<const u32x2>)(GC<uint32_t>(t.buf1) + 4 * row0 + (is_list ? 2 : 0));
#pragma unroll
    for (int k = 0; k < kTileRows / kBlockThreads; k++) {
      const int r = static_cast<int>(threadIdx.x) + k * kBlockThreads;
      const u32x2 ly = lens[2 * (r < n ? r : n - 1)];  // not nontemporal: the encode loop finds the rows of a tile with longer strings in L2
      pre_len[k] = ly.x;
      pre_bytes[k] = ly.y;
    }
  }
  enc_tile_validity(t, row0, n, null_counts, s_valid);
  gptr<const u32x4> str = GC<u32x4>(t.buf1) + row0;
  // the first sub-block's rows are on their way while the tile adds up its lengths
  u32x4 nxt = __builtin_nontemporal_load(str + (static_cast<int>(threadIdx.x) < n ? static_cast<int>(threadIdx.x) : n - 1));
  __syncthreads();  // s_valid
  // ---- the tile's payload size: the length field of every valid row, 64 bits so that it is exact whatever the strings hold
  {
    unsigned long long local = 0;
    uint32_t longest = 0;
    // ... and, in the same pass, the bytes of every (sub-block, wave) group of 64 rows, so that the encode loop below does
    // not wait for the other waves' totals (32-bit: tiles that could wrap take the slow path)
#pragma unroll
    for (int k = 0; k < kTileRows / kBlockThreads; k++) {
      const int r = static_cast<int>(threadIdx.x) + k * kBlockThreads;
      const uint32_t l = (r < n && ((s_valid[r >> 6] >> (r & 63)) & 1)) ? pre_len[k] : 0u;
      local += l;
      longest = longest > l ? longest : l;
      const uint32_t incl = wave_inclusive_scan_u32(l);
      if (lane == 63) s_wtot[k * kWaves + wave] = incl;
      pre_len[k] = l;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) local += __shfl_down(local, d, 64);
    const bool wave_big = __any(longest >= kEncBigLen), wave_tiny = !__any(longest > 4u);
    if (lane == 0) {
      s_sum[wave] = local | (wave_big ? (1ull << 63) : 0ull);
      s_tiny[wave] = wave_tiny ? 1u : 0u;
    }
  }
  __syncthreads();
  const unsigned long long packed = s_sum[0] | s_sum[1] | s_sum[2] | s_sum[3];
  const int64_t tile_total = static_cast<int64_t>((s_sum[0] & kStateMask) + (s_sum[1] & kStateMask) + (s_sum[2] & kStateMask) + (s_sum[3] & kStateMask));
  const bool slow = is_list || (packed >> 63) != 0 || tile_total >= (int64_t(1) << 31);  // uniform
  // every string of the tile is at most 4 bytes (flags, codes): its payload is the low bytes of string_t dword 1
  const bool tiny = (s_tiny[0] & s_tiny[1] & s_tiny[2] & s_tiny[3]) != 0;  // uniform
  // ---- decoupled look-back over the tiles of this column (wave 0; one window = the 64 tiles before the current point)
  const uint32_t first_tile = tile_begin[ti];
  if (wave == 0) {
    {  // exclusive prefix of the 32 group totals
      const uint32_t v = lane < 32 ? s_wtot[lane] : 0u;
      const uint32_t inc = wave_inclusive_scan_u32(v);
      if (lane < 32) s_wbase[lane] = inc - v;
      if (lane == 31) s_wbase[32] = inc;
    }
    if (lane == 0) {
      const unsigned long long mine = (tile == first_tile ? (2ull << 62) : (1ull << 62)) | (static_cast<unsigned long long>(tile_total) & kStateMask);
      __hip_atomic_store(&tile_state[tile], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int64_t prefix = 0;
    int64_t hi = static_cast<int64_t>(tile) - 1;  // nearest predecessor not yet accounted for
    while (hi >= static_cast<int64_t>(first_tile)) {
      const int64_t j = hi - lane;
      unsigned long long st = 2ull << 62;  // lanes past the column's first tile: a finished, empty prefix
      if (lane >= kLookBack) st = 1ull << 62;   // lanes outside the step: an empty sum that ends nothing
      else if (j >= static_cast<int64_t>(first_tile)) {
        st = __hip_atomic_load(&tile_state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the predecessor is dispatched before this tile and publishes after one load + one scan; the bound only keeps a
        // logic error from hanging the device (~1 s), it is never reached
        for (int spins = 0; (st >> 62) == 0; spins++) {
          if (spins > (1 << 22)) {
            atomicOr(status, MI_ST_INTERNAL);
            st = 2ull << 62;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          st = __hip_atomic_load(&tile_state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // the nearest tile (smallest lane) that already knows its inclusive prefix ends the walk
      const uint64_t done_mask = __ballot((st >> 62) == 2ull);
      const int stop = done_mask ? __builtin_ctzll(done_mask) : 64;
      int64_t v = lane <= stop ? static_cast<int64_t>(st & kStateMask) : 0;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
      prefix += __shfl(v, 0, 64);
      if (done_mask) break;
      hi -= kLookBack;
    }
    if (lane == 0) {
      if (tile != first_tile)
        __hip_atomic_store(&tile_state[tile], (2ull << 62) | (static_cast<unsigned long long>(prefix + tile_total) & kStateMask), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      s_prefix = prefix;
      // the column's last tile knows the size of its data buffer: int32 offsets must fit (ArrowAppender's check)
      if (tile + 1 == tile_begin[ti + 1] && !large && prefix + tile_total > 0x7FFFFFFFll) atomicOr(status, MI_ST_OFFSET_OVERFLOW);
      if (slow) tile_state[total_tiles + 1 + tile] = static_cast<unsigned long long>(prefix) | (1ull << 63);
    }
  }
  if (slow) return;  // uniform: encode_string_slow (launched right behind) takes the tile with 64-bit positions
  __syncthreads();
  const int64_t tile_base = s_prefix;
  gptr<const uint8_t> heap = GC<uint8_t>(t.buf2);
  gptr<int32_t> offp = GM<int32_t>(t.out_data) + row0 + 1;
  gptr<int64_t> offp64 = GM<int64_t>(t.out_data) + row0 + 1;
  gptr<uint8_t> data = GM<uint8_t>(t.out_aux);
  int64_t base = tile_base;
  if (row0 == 0 && threadIdx.x == 0) {
    if (large) offp64[-1] = 0;
    else offp[-1] = 0;
  }
  if (tiny) {  // uniform.  <= 8 KiB for the tile: one window over both stage buffers, one barrier, nothing more to load
    const uint32_t shiftw = static_cast<uint32_t>(tile_base & 15);
#pragma unroll
    for (int k = 0; k < kTileRows / kBlockThreads; k++) {
      const int r = static_cast<int>(threadIdx.x) + k * kBlockThreads;
      const uint32_t len = pre_len[k], ex = s_wbase[k * kWaves + wave] + (wave_inclusive_scan_u32(pre_len[k]) - pre_len[k]), y = pre_bytes[k];
      if (r < n) {
        if (large) offp64[r] = tile_base + ex + len;
        else offp[r] = static_cast<int32_t>(tile_base + ex + len);
      }
      uint8_t* d = stage + shiftw + ex;
      if (len > 0) d[0] = static_cast<uint8_t>(y);
      if (len > 1) d[1] = static_cast<uint8_t>(y >> 8);
      if (len > 2) d[2] = static_cast<uint8_t>(y >> 16);
      if (len > 3) d[3] = static_cast<uint8_t>(y >> 24);
    }
    __syncthreads();
    if (tile_total > 0) stage_to_data(stage, shiftw, shiftw + static_cast<uint32_t>(tile_total), data + tile_base - shiftw);
    return;
  }
  const int nsub = (n + kBlockThreads - 1) / kBlockThreads;
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
    // the row's place inside the sub-block, and the sub-block's size, from the tile-wide scan above
    const uint32_t sub0 = s_wbase[k * kWaves];
    // (the scan inside the wave is done again here: 12 VALU instructions per 64 rows, against 8 KiB of LDS -- one occupancy
    // step -- to carry its result over from the length pass)
    const uint32_t ex = s_wbase[k * kWaves + wave] - sub0 + (wave_inclusive_scan_u32(len) - len);
    const uint32_t total = s_wbase[k * kWaves + kWaves] - sub0;
    // Do the long strings of this wave's 64 rows lie in the heap the way they will lie in the data buffer (every one of them
    // at the same distance from its place in the sub-block)?  Vectors decoded from Arrow buffers do, and so does a staged
    // heap.  Then the bytes between the first and the last long string are one coalesced copy (16 bytes per lane) instead
    // of two to four loads per row, and only the short rows, whose bytes are inline, place them
    // themselves -- after the copy, over whatever the heap holds where they go (LDS accesses of one wave execute in program
    // order).  The bytes between two long strings are < 64 * 12 bytes apart, so they share a 4 KiB page with the end of one
    // or the start of the other: reading them cannot fault.  Anything else takes the per-row path below.
    const uint64_t long_mask = __ballot(len > 12);
    bool contig = false;      // wave-uniform
    uint64_t cbase = 0;       // heap offset of the sub-block's byte 0, were it all in the heap
    uint32_t blo = 0, bhi = 0;  // sub-block bytes [blo, bhi): first long string .. end of the last one
    if (long_mask != 0) {
      const uint64_t cdelta = (static_cast<uint64_t>(s.z) | (static_cast<uint64_t>(s.w) << 32)) - t.ptr_base - ex;
      const int fl = __builtin_ctzll(long_mask), ll = 63 - __builtin_clzll(long_mask);
      cbase = static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(cdelta)), fl))) |
              (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(cdelta >> 32)), fl))) << 32);
      contig = __ballot(len > 12 && cdelta != cbase) == 0;
      blo = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ex), fl));
      bhi = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ex + len), ll));
    }
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
      {
        if (contig) {  // wave-uniform: sub-block bytes [a, b) of this window come straight from the heap
          const uint32_t a = blo > w0 ? blo : w0, b = bhi < w1 ? bhi : w1;
          if (a < b) {
            const uint32_t sa = shiftw + (a - w0), sb = shiftw + (b - w0);        // ... as stage bytes [sa, sb)
            gptr<const uint8_t> src0 = heap + cbase + w0 - shiftw;                 // src0[i] is what stage byte i holds
            const uint32_t ca = (sa + 15u) & ~15u, cb = sb & ~15u;                 // whole 16-byte stage rows [ca, cb)
            for (uint32_t c = ca + 16u * lane; c < cb; c += 16u * 64u)
              *reinterpret_cast<u32x4*>(st + c) = __builtin_nontemporal_load((gptr<const u32x4_a1>)(src0 + c));
            // < 16 bytes in front of them and < 16 behind, one byte per lane
            const uint32_t head_end = ca < sb ? ca : sb, tail0 = ca > cb ? ca : cb;
            const uint32_t idx = lane < 16 ? sa + lane : tail0 + (lane - 16);
            const bool mine = lane < 16 ? idx < head_end : (lane < 32 && idx < sb);
            if (mine) st[idx] = src0[idx];
          }
        }
        if (lo < hi && !(contig && len > 12))
          string_bytes_to_lds(st + shiftw + (lo - w0), s, heap, t.ptr_base, lo - ex, hi - lo);
      }
      __syncthreads();
      stage_to_data(st, shiftw, shiftw + (w1 - w0), data + (base + w0) - shiftw);
      buf ^= 1u;
      w0 = w1;
    }
    base += total;
  }
}

// Tiles encode_string_1p left alone (list offsets; a string of >= 8 MiB): 64-bit positions, sub-block by sub-block.  A small
// persistent grid sweeps the flags 256 at a time; validity bitmap and NULL count were already written by encode_string_1p.
__global__ __launch_bounds__(kBlockThreads) void encode_string_slow(const mi_col_task* __restrict__ tasks,
                                                                   const uint32_t* __restrict__ tile_begin,
                                                                   const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                   uint32_t total_tiles, const unsigned long long* __restrict__ tile_state,
                                                                   int per_tile) {
  __shared__ int64_t lds4[kBlockThreads / 64];
  __shared__ __attribute__((aligned(16))) uint8_t stage[kEncStage + 16];
  __shared__ uint32_t s_todo[kBlockThreads];
  __shared__ uint32_t s_ntodo;
  (void)n_tasks;
  const unsigned long long* flags = tile_state + total_tiles + 1;
  if (per_tile) {  // plans with list columns: many tiles are flagged, one workgroup each
    const uint32_t tile = blockIdx.x;
    if (tile >= total_tiles || !(flags[tile] >> 63)) return;  // uniform
    MI_TILE_PROLOGUE();
    encode_string_tile_generic(t, row0, n, static_cast<int64_t>(flags[tile] & 0x7FFFFFFFFFFFFFFFull), lds4, stage);
    return;
  }
  for (uint32_t first = blockIdx.x * kBlockThreads; first < total_tiles; first += gridDim.x * kBlockThreads) {
    if (threadIdx.x == 0) s_ntodo = 0;
    __syncthreads();
    const uint32_t mine = first + threadIdx.x;
    if (mine < total_tiles && (flags[mine] >> 63)) s_todo[atomicAdd(&s_ntodo, 1u)] = mine;
    __syncthreads();
    const uint32_t ntodo = s_ntodo;
    for (uint32_t j = 0; j < ntodo; j++) {
      const uint32_t tile = s_todo[j];
      MI_TILE_PROLOGUE();
      encode_string_tile_generic(t, row0, n, static_cast<int64_t>(flags[tile] & 0x7FFFFFFFFFFFFFFFull), lds4, stage);
      __syncthreads();
    }
    __syncthreads();
  }
}

}  // namespace

// Encode launches follow the decode rule: one workgroup per tile (the dispatcher balances them), owner of a tile from
// the tile -> task table.
hipError_t LaunchEncodeFixed(const mi_col_task* d_tasks, const uint32_t* d_tile_begin, const uint32_t* d_tile_task,
                             int32_t n_tasks, uint32_t total_tiles, int64_t* d_null_counts, hipStream_t stream) {
  MI_DROP_STALE_ERROR();
  if (total_tiles == 0) return hipSuccess;
  hipLaunchKernelGGL(encode_fixed, dim3(total_tiles), dim3(kBlockThreads), 0, stream, d_tasks, d_tile_begin, d_tile_task, n_tasks,
                     total_tiles, d_null_counts);
  return hipGetLastError();
}

// d_tile_state: 2 * total_tiles + 1 words (look-back states, the ticket counter, the big-tile flags), zeroed here
hipError_t LaunchEncodeString(const mi_col_task* d_tasks, const uint32_t* d_tile_begin, const uint32_t* d_tile_task,
                              int32_t n_tasks, uint32_t total_tiles, int64_t* d_tile_state, int64_t* d_null_counts,
                              uint32_t* d_status, bool has_lists, hipStream_t stream) {
  MI_DROP_STALE_ERROR();
  if (total_tiles == 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(d_tile_state, 0, (2 * static_cast<size_t>(total_tiles) + 1) * sizeof(int64_t), stream);
  if (e != hipSuccess) return e;
  unsigned long long* state = reinterpret_cast<unsigned long long*>(d_tile_state);
  hipLaunchKernelGGL(encode_string_1p, dim3(total_tiles), dim3(kBlockThreads), 0, stream, d_tasks, d_tile_begin, d_tile_task, n_tasks,
                     total_tiles, state, d_null_counts, d_status);
  const uint32_t sweep = (total_tiles + kBlockThreads - 1) / kBlockThreads;
  hipLaunchKernelGGL(encode_string_slow, dim3(has_lists ? total_tiles : (sweep < 512u ? sweep : 512u)), dim3(kBlockThreads), 0, stream,
                     d_tasks, d_tile_begin, d_tile_task, n_tasks, total_tiles, state, has_lists ? 1 : 0);
  return hipGetLastError();
}

}  // namespace device
}  // namespace miarrow
