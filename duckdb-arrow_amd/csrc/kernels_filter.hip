// kernels_filter.hip -- K6: pushed-down predicates -> selection vectors, and the fused Q6-style consumer.
#include "device_common.hpp"

#include <algorithm>

namespace miarrow {
namespace device {

namespace {

// ---------------------------------------------------------------------------------------------------- K6
// Range filter lo <= v < hi AND valid -> ascending window-relative indices, one selection vector per 2048-row window.
// A workgroup takes kFilterWindows consecutive windows: lane i owns rows [8i, 8i+8) of each (16 to 64 contiguous bytes,
// loaded as one vector), all windows' loads are issued before anything depends on them (a single 8 KB window per
// workgroup left the kernel bound by the latency of that one round trip: 2.1 TB/s), then per window a DPP wave scan +
// one LDS exchange of the 4 wave totals gives every lane its output position: the selection vector comes out sorted
// without a second pass.
constexpr int kFilterWindows = 4;

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


}  // namespace device
}  // namespace miarrow
