// kernels_gather.hip -- late materialisation through the selection vector (K6 + K1-K5).
//
// When a pushed-down filter has produced a selection vector per 2048-row window, the projected columns are decoded for the
// SELECTED rows only.  They land densely packed, in row order, at the start of the column's output array: output row of
// window w, selected row i = (sum of the counts of the windows before w) + i.  The chunks of the batch are then ordinary
// flat 2048-row vectors of the surviving rows, without a selection vector.
//
// DuckDB reaches the same vectors by slicing the scan's output with the filter's selection: the reference declares
// filter_pushdown = false (src/scanner/read_arrow.cpp:47-48), so the filter runs above the scan.  Here the rows that fail
// the predicate are never written.  lineitem at 16 % selectivity: 158 B/row of vectors shrink to 25 B/row, and so does
// the D2H that follows for a host consumer.
//
// One workgroup per window (tile).  Lane r handles selected row r: source row = sel[r] (window relative, ascending), so
// loads are monotone gathers inside a 2048-row window (every 64-byte line is touched at most once per wave pass) and
// stores are coalesced.  Validity: the host presets the output words to all ones; a column with NULLs clears bits with one
// wave ballot + at most two atomicAnd per 64 rows (a window's output range is not word aligned).
#include "device_common.hpp"

namespace miarrow {
namespace device {

namespace {

__device__ __forceinline__ bool src_row_valid(const mi_col_task& t, bool has_nulls, int64_t row) {
  if (!has_nulls) return true;
  const int64_t bit = t.row_offset + row;
  return (GC<uint64_t>(t.validity)[bit >> 6] >> (bit & 63)) & 1;
}

// BODY(out_row, src_row, ok): stores the value of one selected row (canonical 0 when !ok).  `row0` = first source row of
// the window, `out0` = first output row of the window.
template <typename BODY>
__device__ __forceinline__ void gather_rows(const mi_col_task& t, int64_t row0, int64_t out0, gptr<const uint32_t> sel, int cnt, BODY&& body) {
  const bool has_nulls = t.validity != nullptr && t.null_count != 0;
  const int lane = threadIdx.x & 63;
  for (int base = 0; base < cnt; base += kBlockThreads) {  // uniform
    const int r = base + static_cast<int>(threadIdx.x);
    const bool active = r < cnt;
    const int64_t src = row0 + (active ? static_cast<int64_t>(sel[r]) : 0);
    const bool ok = active && src_row_valid(t, has_nulls, src);
    if (active) body(out0 + r, src, ok);
    if (has_nulls && t.out_validity != nullptr) {  // uniform
      const uint64_t word = __ballot(ok || !active);
      if (lane == 0 && r < cnt && word != ~0ull) {
        const int64_t pos = out0 + r;  // output row of this wave's lane 0
        const int sh = static_cast<int>(pos & 63);
        unsigned long long* W = reinterpret_cast<unsigned long long*>(t.out_validity) + (pos >> 6);
        atomicAnd(W, (word << sh) | ((1ull << sh) - 1ull));
        if (sh) atomicAnd(W + 1, (word >> (64 - sh)) | (~0ull << sh));
      }
    }
  }
}

template <typename T>
__device__ __forceinline__ void gather_copy(const mi_col_task& t, int64_t row0, int64_t out0, gptr<const uint32_t> sel, int cnt) {
  gptr<const T> src = GC<T>(t.buf1) + t.row_offset;
  gptr<T> out = GM<T>(t.out_data);
  gather_rows(t, row0, out0, sel, cnt, [&](int64_t r, int64_t s, bool) { out[r] = src[s]; });  // NULL slots keep the source bytes
}

template <typename OUT>
__device__ __forceinline__ void gather_dec128(const mi_col_task& t, int64_t row0, int64_t out0, gptr<const uint32_t> sel, int cnt, uint32_t* status) {
  gptr<const uint8_t> src = GC<uint8_t>(t.buf1) + t.row_offset * 16;
  gptr<OUT> out = GM<OUT>(t.out_data);
  uint32_t err = 0;
  gather_rows(t, row0, out0, sel, cnt, [&](int64_t r, int64_t s, bool ok) {
    const u32x4 v = *(gptr<const u32x4_a4>)(src + 16 * s);
    const uint64_t lower = static_cast<uint64_t>(v.x) | (static_cast<uint64_t>(v.y) << 32);
    const int64_t upper = static_cast<int64_t>(static_cast<uint64_t>(v.z) | (static_cast<uint64_t>(v.w) << 32));
    OUT o = 0;
    if (ok) {
      o = static_cast<OUT>(lower);
      const int64_t sext = static_cast<int64_t>(o);
      if (static_cast<uint64_t>(sext) != lower || upper != (sext >> 63)) err = MI_ST_DECIMAL_RANGE;
    }
    out[r] = o;
  });
  raise(status, err);
}

template <typename OFF>
__device__ __forceinline__ void gather_string(const mi_col_task& t, int64_t row0, int64_t out0, gptr<const uint32_t> sel, int cnt, uint32_t* status) {
  gptr<const OFF> off = GC<OFF>(t.buf1) + t.row_offset;
  gptr<const uint8_t> data = GC<uint8_t>(t.buf2);
  gptr<u32x4> out = GM<u32x4>(t.out_data);
  const int64_t data_len = t.buf2_len;
  uint32_t err = 0;
  gather_rows(t, row0, out0, sel, cnt, [&](int64_t r, int64_t s, bool ok) {
    const int64_t a = static_cast<int64_t>(off[s]), b = static_cast<int64_t>(off[s + 1]);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (!(a >= 0 && b >= a && b <= data_len)) err |= MI_ST_BAD_OFFSETS;  // only the selected rows are read, so only they are checked
    else if (sizeof(OFF) == 8 && b > 0xFFFFFFFFll) err |= MI_ST_STRING_TOO_LARGE;
    else if (ok) v = make_string_t(data, a, static_cast<uint32_t>(b - a), t.ptr_base);
    __builtin_nontemporal_store(v, out + r);
  });
  raise(status, err);
}

// First output row of every window = selected rows of the windows before it in the same column: one workgroup per task
// scans the task's counts (a record batch has a few dozen windows; a 100 M row column handed over at kernel level has 50 k).
__global__ __launch_bounds__(kBlockThreads) void gather_window_bases(const mi_col_task* __restrict__ tasks,
                                                                     const uint32_t* __restrict__ tile_begin, int n_tasks,
                                                                     int64_t* __restrict__ window_base) {
  __shared__ int64_t s_wave[kBlockThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int ti = blockIdx.x; ti < n_tasks; ti += gridDim.x) {
    const uint32_t first = tile_begin[ti], last = tile_begin[ti + 1];
    gptr<const uint32_t> counts = GC<uint32_t>(tasks[ti].sel_count);
    int64_t carry = 0;
    constexpr uint32_t kPer = 8;   // consecutive windows per thread and step
    for (uint32_t base = first; base < last; base += kBlockThreads * kPer) {  // uniform
      const uint32_t i0 = base + threadIdx.x * kPer;
      uint32_t c[kPer];
      int64_t v = 0;
#pragma unroll
      for (uint32_t k = 0; k < kPer; k++) {
        c[k] = i0 + k < last ? counts[i0 + k - first] : 0u;
        v += c[k];
      }
      int64_t incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int64_t up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
      }
      __syncthreads();  // the previous round's readers are done
      if (lane == 63) s_wave[wave] = incl;
      __syncthreads();
      int64_t before = 0, total = 0;
#pragma unroll
      for (int w = 0; w < kBlockThreads / 64; w++) {
        if (w < wave) before += s_wave[w];
        total += s_wave[w];
      }
      int64_t at = carry + before + incl - v;
#pragma unroll
      for (uint32_t k = 0; k < kPer; k++) {
        if (i0 + k < last) window_base[i0 + k] = at;
        at += c[k];
      }
      carry += total;
    }
  }
}

__global__ __launch_bounds__(kBlockThreads) void transcode_gather(const mi_col_task* __restrict__ tasks,
                                                                  const uint32_t* __restrict__ tile_begin,
                                                                  const uint32_t* __restrict__ tile_task, int n_tasks,
                                                                  uint32_t total_tiles, const int64_t* __restrict__ window_base,
                                                                  uint32_t* __restrict__ status) {
  for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    MI_TILE_PROLOGUE();
    (void)n;
    const int64_t window = row0 / kTileRows;
    gptr<const uint32_t> sel = GC<uint32_t>(t.sel) + row0;
    const int cnt = static_cast<int>(GC<uint32_t>(t.sel_count)[window]);
    if (cnt <= 0) continue;  // uniform
    const int64_t out0 = window_base[tile];  // first output row of this window (gather_window_bases)
    switch (t.kind) {
      case MI_K_COPY:
        switch (t.param) {
          case 1: gather_copy<uint8_t>(t, row0, out0, sel, cnt); break;
          case 2: gather_copy<uint16_t>(t, row0, out0, sel, cnt); break;
          case 4: gather_copy<uint32_t>(t, row0, out0, sel, cnt); break;
          case 8: gather_copy<uint64_t>(t, row0, out0, sel, cnt); break;
          default: {
            gptr<const uint8_t> src = GC<uint8_t>(t.buf1) + t.row_offset * 16;
            gptr<u32x4> out = GM<u32x4>(t.out_data);
            gather_rows(t, row0, out0, sel, cnt, [&](int64_t r, int64_t s, bool) { out[r] = *(gptr<const u32x4_a4>)(src + 16 * s); });
            break;
          }
        }
        break;
      case MI_K_DEC128:
        if (t.param == 8) gather_dec128<int64_t>(t, row0, out0, sel, cnt, status);
        else if (t.param == 4) gather_dec128<int32_t>(t, row0, out0, sel, cnt, status);
        else gather_dec128<int16_t>(t, row0, out0, sel, cnt, status);
        break;
      case MI_K_STR32: gather_string<int32_t>(t, row0, out0, sel, cnt, status); break;
      case MI_K_STR64: gather_string<int64_t>(t, row0, out0, sel, cnt, status); break;
      case MI_K_FIXED_BINARY: {
        gptr<const uint8_t> data = GC<uint8_t>(t.buf1);
        gptr<u32x4> out = GM<u32x4>(t.out_data);
        const int64_t width = t.param;
        gather_rows(t, row0, out0, sel, cnt, [&](int64_t r, int64_t s, bool ok) {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (ok) v = make_string_t(data, (t.row_offset + s) * width, static_cast<uint32_t>(width), t.ptr_base);
          out[r] = v;
        });
        break;
      }
      case MI_K_BOOL: {
        gptr<const uint8_t> bits = GC<uint8_t>(t.buf1);
        gptr<uint8_t> out = GM<uint8_t>(t.out_data);
        gather_rows(t, row0, out0, sel, cnt, [&](int64_t r, int64_t s, bool) {
          const int64_t bit = t.row_offset + s;
          out[r] = (bits[bit >> 3] >> (bit & 7)) & 1;
        });
        break;
      }
      case MI_K_DATE64: {
        gptr<const int64_t> src = GC<int64_t>(t.buf1) + t.row_offset;
        gptr<int32_t> out = GM<int32_t>(t.out_data);
        gather_rows(t, row0, out0, sel, cnt, [&](int64_t r, int64_t s, bool) { out[r] = static_cast<int32_t>(src[s] / 86400000ll); });
        break;
      }
      case MI_K_MUL_I32: {
        gptr<const int32_t> src = GC<int32_t>(t.buf1) + t.row_offset;
        gptr<int64_t> out = GM<int64_t>(t.out_data);
        gather_rows(t, row0, out0, sel, cnt, [&](int64_t r, int64_t s, bool ok) { out[r] = ok ? static_cast<int64_t>(src[s]) * t.param : 0; });
        break;
      }
      case MI_K_MUL_I64: {
        gptr<const int64_t> src = GC<int64_t>(t.buf1) + t.row_offset;
        gptr<int64_t> out = GM<int64_t>(t.out_data);
        uint32_t err = 0;
        gather_rows(t, row0, out0, sel, cnt, [&](int64_t r, int64_t s, bool ok) {
          int64_t v = 0;
          if (ok && __builtin_mul_overflow(src[s], t.param, &v)) {
            v = 0;
            err = MI_ST_MUL_OVERFLOW;
          }
          out[r] = v;
        });
        raise(status, err);
        break;
      }
      case MI_K_DIV_I64: {
        gptr<const int64_t> src = GC<int64_t>(t.buf1) + t.row_offset;
        gptr<int64_t> out = GM<int64_t>(t.out_data);
        const int64_t d = t.param;
        gather_rows(t, row0, out0, sel, cnt, [&](int64_t r, int64_t s, bool) { out[r] = d == 1000 ? src[s] / 1000 : src[s] / d; });
        break;
      }
      case MI_K_DICT: {
        const int iw = static_cast<int>(t.param & 0xFF);
        const bool is_signed = ((t.param >> 8) & 1) != 0;
        gptr<const uint8_t> idx = GC<uint8_t>(t.buf1) + t.row_offset * iw;
        gptr<uint32_t> out = GM<uint32_t>(t.out_data);
        const uint32_t dict_len = static_cast<uint32_t>(t.param2);
        uint32_t err = 0;
        gather_rows(t, row0, out0, sel, cnt, [&](int64_t r, int64_t s, bool ok) {
          uint32_t o = dict_len;
          if (ok) {
            uint64_t v;
            switch (iw) {
              case 1: v = is_signed ? static_cast<uint64_t>(static_cast<int64_t>(((gptr<const int8_t>)idx)[s])) : idx[s]; break;
              case 2: v = is_signed ? static_cast<uint64_t>(static_cast<int64_t>(((gptr<const int16_t>)idx)[s])) : ((gptr<const uint16_t>)idx)[s]; break;
              case 4: v = is_signed ? static_cast<uint64_t>(static_cast<int64_t>(((gptr<const int32_t>)idx)[s])) : ((gptr<const uint32_t>)idx)[s]; break;
              default: v = ((gptr<const uint64_t>)idx)[s]; break;
            }
            if (v > 0xFFFFFFFFull) { err = MI_ST_INDEX_RANGE; v = dict_len; }
            else if (v >= dict_len) { err = MI_ST_DICT_INDEX; v = dict_len; }
            o = static_cast<uint32_t>(v);
          }
          out[r] = o;
        });
        raise(status, err);
        break;
      }
      default: break;
    }
  }
}

}  // namespace

bool KindCanGather(int32_t kind) {
  switch (kind) {
    case MI_K_COPY: case MI_K_DEC128: case MI_K_STR32: case MI_K_STR64: case MI_K_FIXED_BINARY: case MI_K_BOOL: case MI_K_DATE64:
    case MI_K_MUL_I32: case MI_K_MUL_I64: case MI_K_DIV_I64: case MI_K_DICT:
      return true;
    default:
      return false;
  }
}

hipError_t LaunchGather(const mi_col_task* d_tasks, const uint32_t* d_tile_begin, const uint32_t* d_tile_task, int32_t n_tasks,
                        uint32_t total_tiles, int64_t* d_window_base, uint32_t* d_status, hipStream_t stream) {
  MI_DROP_STALE_ERROR();
  if (total_tiles == 0) return hipSuccess;
  hipLaunchKernelGGL(gather_window_bases, dim3(static_cast<uint32_t>(n_tasks < 4096 ? n_tasks : 4096)), dim3(kBlockThreads), 0, stream, d_tasks,
                     d_tile_begin, n_tasks, d_window_base);
  hipLaunchKernelGGL(transcode_gather, dim3(total_tiles), dim3(kBlockThreads), 0, stream, d_tasks, d_tile_begin, d_tile_task, n_tasks,
                     total_tiles, d_window_base, d_status);
  return hipGetLastError();
}

}  // namespace device
}  // namespace miarrow
