// kernels_filter.hip -- K6: pushed-down predicates -> selection vectors, and the fused Q6-style consumer.
#include "device_common.hpp"

#include <algorithm>
#include <cstring>
#include <type_traits>

namespace miarrow {
namespace device {

namespace {

// ---------------------------------------------------------------------------------------------------- K6
// Pushed-down predicate -> ascending window-relative row indices, one selection vector per 2048-row window (the
// reference sets filter_pushdown = false, src/scanner/read_arrow.cpp:47-48: these are the rows DuckDB's own filter above
// the scan would keep; a comparison with NULL is false).  The predicate arrives in conjunctive normal form as a kernel
// argument (FilterProgram, scalar loads): leaves are inclusive ranges on the stored integers (optionally negated),
// IN-lists and IS [NOT] NULL over decoded fixed-width vectors of any projected column.
// A workgroup takes kFilterWindows consecutive windows: lane i owns rows [8i, 8i+8) of each (8 to 64 contiguous bytes,
// loaded as one vector), the loads of all windows of a leaf are issued before anything depends on them (a single 8 KB
// window per workgroup left the kernel bound by the latency of that one round trip: 2.1 TB/s), then per window a DPP wave
// scan + one LDS exchange of the 4 wave totals gives every lane its output position: the selection vector comes out
// sorted without a second pass.
constexpr int kFilterWindows = 4;

template <typename T>
__device__ __forceinline__ void leaf_compare(const FilterLeafDev& L, int64_t first_window, int64_t nrows, int r,
                                             uint32_t (&m)[kFilterWindows]) {
  typedef T vec8 __attribute__((ext_vector_type(8)));
  typedef vec8 vec8_a4 __attribute__((aligned(4)));
  gptr<const T> values = GC<T>(L.data);
  const bool is_unsigned = (L.flags & kLeafUnsigned) != 0;
  const int64_t bias = (L.flags & kLeafBias) ? static_cast<int64_t>(0x8000000000000000ull) : 0;  // uint64: order-preserving map to int64
  int64_t x[kFilterWindows][8];
#pragma unroll
  for (int w = 0; w < kFilterWindows; w++) {
    const int64_t row0 = (first_window + w) * kTileRows;
    const int64_t left = nrows - row0;
    const int n = left < kTileRows ? static_cast<int>(left < 0 ? 0 : left) : kTileRows;
    if (r + 8 <= n) {
      const vec8 v = __builtin_nontemporal_load((gptr<const vec8_a4>)(values + row0 + r));
#pragma unroll
      for (int k = 0; k < 8; k++) {
        typedef typename std::make_unsigned<T>::type UT;
        x[w][k] = (is_unsigned ? static_cast<int64_t>(static_cast<uint64_t>(static_cast<UT>(v[k]))) : static_cast<int64_t>(v[k])) ^ bias;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        typedef typename std::make_unsigned<T>::type UT;
        const T v = r + k < n ? values[row0 + r + k] : T(0);
        x[w][k] = (is_unsigned ? static_cast<int64_t>(static_cast<uint64_t>(static_cast<UT>(v))) : static_cast<int64_t>(v)) ^ bias;
      }
    }
  }
  if (L.op == kLeafRange) {
#pragma unroll
    for (int w = 0; w < kFilterWindows; w++) {
      uint32_t mm = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) mm |= (x[w][k] >= L.lo && x[w][k] <= L.hi) ? (1u << k) : 0u;
      m[w] = mm;
    }
  } else {  // kLeafIn: the list is small and uniform (scalar loads)
#pragma unroll
    for (int w = 0; w < kFilterWindows; w++) m[w] = 0;
    for (int j = 0; j < L.n_in; j++) {
      const int64_t c = L.in_values[j];
#pragma unroll
      for (int w = 0; w < kFilterWindows; w++)
#pragma unroll
        for (int k = 0; k < 8; k++) m[w] |= (x[w][k] == c) ? (1u << k) : 0u;
    }
  }
}

// kLeafStrIn: the column is a decoded string_t vector (16 bytes per row; strings of <= 12 bytes sit inline, zero padded, longer
// ones hold a 4-byte prefix and a pointer into the Arrow data buffer).  A row equals a constant of <= 12 bytes exactly when
// the four dwords are equal; a longer constant needs equal length and prefix, then the bytes behind the pointer (L.lo = the
// device copy of the data buffer, L.hi = the pointer value of its byte 0).  One window at a time: 8 rows are 32 registers.
__device__ __forceinline__ void leaf_strin(const FilterLeafDev& L, int64_t first_window, int64_t nrows, int r, uint32_t (&m)[kFilterWindows]) {
  gptr<const u32x4> rows = GC<u32x4>(L.data);
  gptr<const uint8_t> heap = GC<uint8_t>(reinterpret_cast<const void*>(static_cast<uintptr_t>(L.lo)));
  const uint64_t ptr_base = static_cast<uint64_t>(L.hi);
#pragma clang loop unroll(disable)
  for (int w = 0; w < kFilterWindows; w++) {
    const int64_t row0 = (first_window + w) * kTileRows;
    const int64_t left = nrows - row0;
    const int n = left < kTileRows ? static_cast<int>(left < 0 ? 0 : left) : kTileRows;
    u32x4 s[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const u32x4 zero = {0u, 0u, 0u, 0u};
      s[k] = r + k < n ? __builtin_nontemporal_load(rows + row0 + r + k) : zero;
    }
    uint32_t mm = 0;
    for (int j = 0; j < L.n_in; j++) {   // uniform: the constants arrive through scalar loads
      const uint64_t c0 = static_cast<uint64_t>(L.in_values[3 * j]), c1 = static_cast<uint64_t>(L.in_values[3 * j + 1]);
      const uint32_t clen = static_cast<uint32_t>(c0), cw1 = static_cast<uint32_t>(c0 >> 32), cw2 = static_cast<uint32_t>(c1),
                     cw3 = static_cast<uint32_t>(c1 >> 32);
      gptr<const uint8_t> cbytes = GC<uint8_t>(reinterpret_cast<const void*>(static_cast<uintptr_t>(L.in_values[3 * j + 2])));
#pragma unroll
      for (int k = 0; k < 8; k++) {
        bool eq = s[k].x == clen && s[k].y == cw1;
        if (clen <= 12) {
          eq = eq && s[k].z == cw2 && s[k].w == cw3;
        } else if (eq) {
          const uint64_t p = static_cast<uint64_t>(s[k].z) | (static_cast<uint64_t>(s[k].w) << 32);
          gptr<const uint8_t> a = heap + (p - ptr_base);
          typedef uint32_t u32_any __attribute__((aligned(1)));
          uint32_t i = 4;
          for (; i + 4 <= clen && eq; i += 4) eq = *(gptr<const u32_any>)(a + i) == *(gptr<const u32_any>)(cbytes + i);
          for (; i < clen && eq; i++) eq = a[i] == cbytes[i];
        }
        mm |= eq ? (1u << k) : 0u;
      }
    }
    m[w] = mm;
  }
}

// kLeafStrRange: lower <(=) row <(=) upper in byte-wise order (memcmp over the common length, then the shorter string first:
// DuckDB's string_t comparison).  The first four bytes decide most rows: a string_t keeps them in dword 1, zero padded, so the
// byte-swapped dwords compare like the strings do unless they are equal; then the bytes from 4 on are walked -- inline
// strings from the row's own registers, longer ones from the heap -- and at last the lengths.
__device__ __forceinline__ int strrow_compare(const u32x4& s, gptr<const uint8_t> heap, uint64_t ptr_base, uint32_t clen, uint32_t cw1,
                                              gptr<const uint8_t> cbytes) {
  const uint32_t pr = __builtin_bswap32(s.y), pc = __builtin_bswap32(cw1);
  if (pr != pc) return pr < pc ? -1 : 1;
  const uint32_t rlen = s.x, m = rlen < clen ? rlen : clen;
  if (m > 4) {
    gptr<const uint8_t> a = nullptr;
    if (rlen > 12) {
      const uint64_t p = static_cast<uint64_t>(s.z) | (static_cast<uint64_t>(s.w) << 32);
      a = heap + (p - ptr_base);
    }
    for (uint32_t i = 4; i < m; i++) {
      const uint32_t rb = rlen > 12 ? a[i] : ((i < 8 ? s.z : s.w) >> (8 * (i & 3))) & 0xFFu;
      const uint32_t cb = cbytes[i];
      if (rb != cb) return rb < cb ? -1 : 1;
    }
  }
  return rlen < clen ? -1 : rlen > clen ? 1 : 0;
}

__device__ __forceinline__ void leaf_strrange(const FilterLeafDev& L, int64_t first_window, int64_t nrows, int r, uint32_t (&m)[kFilterWindows]) {
  gptr<const u32x4> rows = GC<u32x4>(L.data);
  gptr<const uint8_t> heap = GC<uint8_t>(reinterpret_cast<const void*>(static_cast<uintptr_t>(L.lo)));
  const uint64_t ptr_base = static_cast<uint64_t>(L.hi);
  const bool has_lo = (L.n_in & 1) != 0, lo_incl = (L.n_in & 2) != 0, has_hi = (L.n_in & 4) != 0, hi_incl = (L.n_in & 8) != 0;
  const uint64_t l0 = static_cast<uint64_t>(L.in_values[0]), h0 = static_cast<uint64_t>(L.in_values[3]);
  gptr<const uint8_t> lbytes = GC<uint8_t>(reinterpret_cast<const void*>(static_cast<uintptr_t>(L.in_values[2])));
  gptr<const uint8_t> hbytes = GC<uint8_t>(reinterpret_cast<const void*>(static_cast<uintptr_t>(L.in_values[5])));
#pragma clang loop unroll(disable)
  for (int w = 0; w < kFilterWindows; w++) {
    const int64_t row0 = (first_window + w) * kTileRows;
    const int64_t left = nrows - row0;
    const int n = left < kTileRows ? static_cast<int>(left < 0 ? 0 : left) : kTileRows;
    uint32_t mm = 0;
#pragma clang loop unroll(disable)
    for (int k = 0; k < 8; k++) {
      if (r + k >= n) break;
      const u32x4 s = __builtin_nontemporal_load(rows + row0 + r + k);
      bool ok = true;
      if (has_lo) {
        const int c = strrow_compare(s, heap, ptr_base, static_cast<uint32_t>(l0), static_cast<uint32_t>(l0 >> 32), lbytes);
        ok = c > 0 || (c == 0 && lo_incl);
      }
      if (ok && has_hi) {
        const int c = strrow_compare(s, heap, ptr_base, static_cast<uint32_t>(h0), static_cast<uint32_t>(h0 >> 32), hbytes);
        ok = c < 0 || (c == 0 && hi_incl);
      }
      mm |= ok ? (1u << k) : 0u;
    }
    m[w] = mm;
  }
}

// kLeafDictMap: a string predicate on a dictionary-encoded column was evaluated once per dictionary (on the host: one byte per
// entry: 0 no, 1 yes, 2 the entry is NULL); a row passes by looking its index up -- 4 bytes per row instead of a string_t and
// its heap bytes.  Rows without a value point at the extra NULL entry (index dict_len), so the row validity is not needed, and
// IS [NOT] NULL on a dictionary column -- a row is NULL when its index OR its dictionary entry is -- is the same look-up.
__device__ __forceinline__ void leaf_dictmap(const FilterLeafDev& L, int64_t first_window, int64_t nrows, int r, uint32_t (&m)[kFilterWindows]) {
  typedef uint32_t vec8 __attribute__((ext_vector_type(8)));
  typedef vec8 vec8_a4 __attribute__((aligned(4)));
  gptr<const uint32_t> idx = GC<uint32_t>(L.data);
  gptr<const uint8_t> map = GC<uint8_t>(L.in_values);
  const uint32_t n_map = static_cast<uint32_t>(L.n_in);
  const int mode = static_cast<int>(L.lo);   // 0: matches, 1: does not match (NULL fails both), 2: IS NULL, 3: IS NOT NULL
  auto pass = [&](uint32_t code) -> bool {
    return mode == 0 ? code == 1u : mode == 1 ? code == 0u : mode == 2 ? code == 2u : code != 2u;
  };
#pragma unroll
  for (int w = 0; w < kFilterWindows; w++) {
    const int64_t row0 = (first_window + w) * kTileRows;
    const int64_t left = nrows - row0;
    const int n = left < kTileRows ? static_cast<int>(left < 0 ? 0 : left) : kTileRows;
    uint32_t mm = 0;
    if (r + 8 <= n) {
      const vec8 v = __builtin_nontemporal_load((gptr<const vec8_a4>)(idx + row0 + r));
#pragma unroll
      for (int k = 0; k < 8; k++) mm |= pass(v[k] < n_map ? map[v[k]] : 2u) ? (1u << k) : 0u;
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (r + k < n) {
          const uint32_t i = idx[row0 + r + k];
          mm |= pass(i < n_map ? map[i] : 2u) ? (1u << k) : 0u;
        }
      }
    }
    m[w] = mm;
  }
}

__global__ __launch_bounds__(kBlockThreads) void filter_program(const FilterProgram prog, int64_t nrows,
                                                                mi_sel_t* __restrict__ sel_out_p,
                                                                uint32_t* __restrict__ count_out_p) {
  __shared__ uint32_t wave_total[kFilterWindows][kBlockThreads / 64];
  gptr<mi_sel_t> sel_out = GM<mi_sel_t>(sel_out_p);
  gptr<uint32_t> count_out = GM<uint32_t>(count_out_p);
  const int64_t first_window = static_cast<int64_t>(blockIdx.x) * kFilterWindows;
  const int r = 8 * threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t acc[kFilterWindows], clause[kFilterWindows];
#pragma unroll
  for (int w = 0; w < kFilterWindows; w++) {
    const int64_t left = nrows - (first_window + w) * kTileRows - r;  // rows of this lane's group that exist
    acc[w] = left >= 8 ? 0xFFu : (left <= 0 ? 0u : ((1u << left) - 1u));
    clause[w] = 0;
  }
#pragma clang loop unroll(disable)
  for (int l = 0; l < prog.n_leaves; l++) {
    const FilterLeafDev& L = prog.leaves[l];
    uint32_t vb[kFilterWindows];
#pragma unroll
    for (int w = 0; w < kFilterWindows; w++) {
      const int64_t row = (first_window + w) * kTileRows + r;  // multiple of 8: the 8 bits sit in one word
      vb[w] = (L.validity != nullptr && row < nrows) ? static_cast<uint32_t>((GC<uint64_t>(L.validity)[row >> 6] >> (row & 63)) & 0xFF) : 0xFFu;
    }
    uint32_t m[kFilterWindows];
    if (L.op == kLeafIsNull || L.op == kLeafIsNotNull) {
#pragma unroll
      for (int w = 0; w < kFilterWindows; w++) m[w] = L.op == kLeafIsNull ? (~vb[w] & 0xFFu) : vb[w];
    } else {
      if (L.op == kLeafStrIn) leaf_strin(L, first_window, nrows, r, m);
      else if (L.op == kLeafStrRange) leaf_strrange(L, first_window, nrows, r, m);
      else if (L.op == kLeafDictMap) leaf_dictmap(L, first_window, nrows, r, m);
      else switch (L.width) {
        case 1: leaf_compare<int8_t>(L, first_window, nrows, r, m); break;
        case 2: leaf_compare<int16_t>(L, first_window, nrows, r, m); break;
        case 4: leaf_compare<int32_t>(L, first_window, nrows, r, m); break;
        default: leaf_compare<int64_t>(L, first_window, nrows, r, m); break;
      }
#pragma unroll
      for (int w = 0; w < kFilterWindows; w++) {
        if (L.flags & kLeafNegate) m[w] = ~m[w] & 0xFFu;
        if (L.op != kLeafDictMap) m[w] &= vb[w];  // a comparison with NULL is not true (the dictionary map knows its NULLs)
      }
    }
#pragma unroll
    for (int w = 0; w < kFilterWindows; w++) {
      clause[w] |= m[w];
      if (L.flags & kLeafEndsClause) {
        acc[w] &= clause[w];
        clause[w] = 0;
      }
    }
  }
  uint32_t incl[kFilterWindows];
#pragma unroll
  for (int w = 0; w < kFilterWindows; w++) {
    incl[w] = wave_inclusive_scan_u32(__builtin_popcount(acc[w]));
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
    uint32_t pos = base + incl[w] - __builtin_popcount(acc[w]);
    gptr<mi_sel_t> out = sel_out + window * kTileRows;
    uint32_t mm = acc[w];
    while (mm) {  // ascending set bits
      const int k = __builtin_ctz(mm);
      mm &= mm - 1;
      out[pos++] = static_cast<mi_sel_t>(r + k);
    }
    if (threadIdx.x == 0) count_out[window] = total;
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

}  // namespace

hipError_t LaunchAggSumProduct(const AggSumProductArgs& args, unsigned long long* d_acc, int num_cus, hipStream_t stream) {
  MI_DROP_STALE_ERROR();
  if (args.nrows <= 0) return hipSuccess;
  const int64_t want = (args.nrows + kBlockThreads * 8 - 1) / (kBlockThreads * 8);   // ~8 rows per lane
  const uint32_t grid = static_cast<uint32_t>(std::min<int64_t>(want, static_cast<int64_t>(num_cus) * 16));
  hipLaunchKernelGGL(agg_sum_product, dim3(grid ? grid : 1), dim3(kBlockThreads), 0, stream, args, d_acc);
  return hipGetLastError();
}

hipError_t LaunchFilterProgram(const FilterProgram& prog, int64_t nrows, mi_sel_t* sel_out, uint32_t* count_out, hipStream_t stream) {
  MI_DROP_STALE_ERROR();
  if (nrows <= 0) return hipSuccess;
  if (prog.n_leaves < 0 || prog.n_leaves > kMaxFilterLeaves) return hipErrorInvalidValue;
  const int64_t windows = (nrows + kTileRows - 1) / kTileRows;
  const uint32_t grid = static_cast<uint32_t>((windows + kFilterWindows - 1) / kFilterWindows);
  hipLaunchKernelGGL(filter_program, dim3(grid), dim3(kBlockThreads), 0, stream, prog, nrows, sel_out, count_out);
  return hipGetLastError();
}

hipError_t LaunchFilterRange(const void* values, int32_t width, const void* validity, int64_t nrows, int64_t lo,
                             int64_t hi, mi_sel_t* sel_out, uint32_t* count_out, hipStream_t stream) {
  MI_DROP_STALE_ERROR();
  if (width != 1 && width != 2 && width != 4 && width != 8) return hipErrorInvalidValue;
  FilterProgram prog;
  memset(&prog, 0, sizeof(prog));
  prog.n_leaves = 1;
  FilterLeafDev& L = prog.leaves[0];
  L.data = values;
  L.validity = static_cast<const uint64_t*>(validity);
  L.op = kLeafRange;
  L.width = width;
  L.flags = kLeafEndsClause;
  L.lo = lo;
  L.hi = hi == INT64_MIN ? hi : hi - 1;  // lo <= v < hi as an inclusive range
  if (hi == INT64_MIN) { L.lo = 1; L.hi = 0; }  // nothing is < INT64_MIN
  return LaunchFilterProgram(prog, nrows, sel_out, count_out, stream);
}

}  // namespace device
}  // namespace miarrow
