// kernels_lz4.hip -- K8: compressed record-batch bodies decompressed in HBM: LZ4_FRAME here, the ZSTD entropy
// stage in kernels_zstd.inl (included below), the copy stages shared.
//
// The reference decompresses every buffer of a compressed record batch on the CPU before it slices the body
// (DuckDBDecompressZstd, src/ipc/stream_reader/base_stream_reader.cpp:11-32; LZ4_FRAME is the other codec of Message.fbs
// BodyCompression and the default of Feather V2 / pyarrow).  Here the COMPRESSED body crosses PCIe and is expanded where the
// transcode kernels read it.
//
// An LZ4 block is a chain of sequences (token, literals, 2-byte match offset): finding the sequences is serial inside a
// block, and inside a frame with linked blocks -- what LZ4F_compressFrame writes -- a match may reach back into the
// previous block, so "one wave per block, copy as you parse" serialises a whole buffer.  The work is therefore cut the
// other way round, into steps that are each data parallel and never wait for another workgroup:
//   1. lz4_parse    one workgroup (256 lanes) per block walks the tokens only (no data is copied): one descriptor per
//                   sequence {output position, literal source, literal length, match length} + offset, and the block's
//                   decompressed size.  The lanes walk 256 segments of the block speculatively, exchange where they leave
//                   them and repeat until their start positions agree (see the kernel).  Every block of every buffer at once.
//                   (ZSTD: zstd_entropy produces the same descriptors from the Huffman / FSE streams.)
//   2. lz4_layout   one lane per buffer: first output byte of each of its blocks (running sum), and the check the
//                   reference makes after decompressing: the sizes must add up to the declared uncompressed length.
//                   (ZSTD: zstd_layout, which also settles the repeat offsets.)
//   3. lz4_expand   one workgroup per block, output-centric: a thread owns 256 consecutive output bytes, finds its first
//                   sequence by binary search and walks on from there.  Every decompressed byte gets a 32-bit LINK word -- a
//                   literal's word holds the byte itself, a match byte's word the position it copies from (always an earlier
//                   byte of the buffer; an overlapping match links straight into its first period).
//   4. lz4_resolve_local / lz4_collect / lz4_resolve_skeleton
//                   pointer jumping over the links, in place: first inside 8 KiB tiles in LDS until nothing moves, then over
//                   the SKELETON only -- the still-open words that other tiles' open words point at -- four hops per round,
//                   <= log5(tiles of the longest buffer) + 1 rounds launched blindly (a round that finds nothing left makes
//                   the later ones return at once).
//   5. lz4_emit     the last hop of everything outside the skeleton, then the bytes leave the link words for the
//                   decompressed body; a word still open here is an error, never silent data.
// HBM traffic per decompressed byte: 4 B memset + 4 B expand + 8 B local pass + 5 B emit (+ the skeleton's few words).  The
// token walk of step 1 is the latency-bound part -- hence the speculative lanes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "device_common.hpp"
#include "kernels.hpp"
#include "zstd_format.hpp"

namespace miarrow {
namespace zstd {
// zstd_format.hpp's memory accesses through global-address-space pointers (see Mem there)
template <typename T>
struct Mem<T __attribute__((address_space(1)))*> {
  using P = T __attribute__((address_space(1)))*;
  static __device__ __forceinline__ uint32_t Load32(P aligned) { return *(const uint32_t __attribute__((address_space(1)))*)aligned; }
  static __device__ __forceinline__ void Store32(P aligned, uint32_t v) { *(uint32_t __attribute__((address_space(1)))*)aligned = v; }
  static __device__ __forceinline__ uintptr_t Address(P p) { return (uintptr_t)p; }
};
template <typename T>
struct Mem<T __attribute__((address_space(3)))*> {
  using P = T __attribute__((address_space(3)))*;
  static __device__ __forceinline__ uint32_t Load32(P aligned) { return *(const uint32_t __attribute__((address_space(3)))*)aligned; }
  static __device__ __forceinline__ void Store32(P aligned, uint32_t v) { *(uint32_t __attribute__((address_space(3)))*)aligned = v; }
  static __device__ __forceinline__ uintptr_t Address(P p) { return (uintptr_t)p; }
};
}  // namespace zstd
namespace device {
namespace {

// A link word: bit 31 set = the byte is known and sits in the low 8 bits (kLinkUntouched: nothing was decompressed here);
// bit 31 clear = the position this byte copies from.  One word is one atomic message: no second array to keep in step.
constexpr uint32_t kLinkKnown = 0x80000000u;
constexpr uint32_t kLinkUntouched = 0xFFFFFFFFu;
constexpr uint32_t kSkelCount = 36;            // round_left[36]: entries of the skeleton list
constexpr int kSkelHops = 4;                   // links followed per skeleton round
constexpr uint32_t kLocalTileQuads = 2048;     // lz4_resolve_local: 8 KiB of output = 32 KiB of LDS per workgroup

__device__ __forceinline__ void lz4_fail(uint32_t* status) { atomicOr(status, MI_ST_DECOMPRESS); }

// Token walk of one lane: from `ip` (a token position, true or guessed) until the first token position >= stop (or the
// end of the block).  Descriptors go to seq[0..cap): output positions are relative to the lane's first sequence.
struct Lz4Walk {
  uint32_t exit;    // where the next lane's first token is (== block end when the block ends here)
  uint32_t nseq;
  uint32_t olen;    // bytes these sequences produce
  uint32_t first8;  // bit g: the walk has a token at mark0 + g (g < 8)
  bool ok;
};
using lptr = const uint8_t __attribute__((address_space(3)))*;

template <bool STORE, typename BYTES>   // BYTES: the compressed bytes in global memory, or the workgroup's LDS copy of the block
__device__ __forceinline__ Lz4Walk lz4_walk(BYTES in, uint32_t ip, uint32_t stop, uint32_t end, uint32_t block_max,
                                             gptr<u32x4> seq, gptr<uint32_t> seq_off, uint32_t cap, uint32_t lit_bias, uint32_t mark0 = 0) {
  Lz4Walk w;
  w.nseq = 0;
  w.olen = 0;
  w.first8 = 0;
  w.ok = true;
  bool zero_offset = false;
  while (ip < stop) {
    if (ip - mark0 < 8u) w.first8 |= 1u << (ip - mark0);
    const uint32_t token = in[ip++];
    uint32_t ll = token >> 4;
    if (ll == 15) {
      uint32_t x;
      do {
        if (ip >= end) { w.ok = false; break; }
        x = in[ip++];
        ll += x;
      } while (x == 255 && ll < (1u << 24));
      if (!w.ok || ll >= (1u << 24)) { w.ok = false; break; }
    }
    const uint32_t lit_src = ip;
    if (ll > end - ip) { w.ok = false; break; }
    ip += ll;
    uint32_t ml = 0, offset = 0;
    if (ip < end) {  // the last sequence of a block is literals only
      if (end - ip < 2) { w.ok = false; break; }
      offset = static_cast<uint32_t>(in[ip]) | (static_cast<uint32_t>(in[ip + 1]) << 8);
      ip += 2;
      ml = token & 15u;
      if (ml == 15) {
        uint32_t x;
        do {
          if (ip >= end) { w.ok = false; break; }
          x = in[ip++];
          ml += x;
        } while (x == 255 && ml < (1u << 24));
        if (!w.ok || ml >= (1u << 24)) { w.ok = false; break; }
      }
      ml += 4;
      if (offset == 0) zero_offset = true;   // an error of the true chain; a guessed walk keeps going until it falls in step
    }
    if (w.nseq >= cap || ll + ml > block_max - w.olen) { w.ok = false; break; }
    u32x4 d;
    d.x = w.olen;
    d.y = lit_src + lit_bias;
    d.z = ll;
    d.w = ml;
    if (STORE) {
      seq[w.nseq] = d;
      seq_off[w.nseq] = offset;
    }
    w.nseq++;
    w.olen += ll + ml;
  }
  // a walk that ran into nonsense knows nothing about where the next segment's chain begins: leave the next lane its own guess
  w.exit = w.ok ? ip : stop;
  if (zero_offset) w.ok = false;
  return w;
}

// The walk that stores nothing and only wants to know where it leaves [ip, stop): ONE dependent read per sequence (the token;
// the literal length and the two offset bytes are skipped, not read), more only for the 255-runs of long lengths.  The bytes
// are not validated here -- a walk that runs into nonsense reports `stop` (no information), the storing walk validates.
template <typename BYTES>
__device__ __forceinline__ uint32_t lz4_probe(BYTES in, uint32_t ip, uint32_t stop, uint32_t end, uint32_t mark0, uint32_t* first8) {
  uint32_t seen = 0;
  while (ip < stop) {
    if (ip - mark0 < 8u) seen |= 1u << (ip - mark0);
    const uint32_t token = in[ip++];
    uint32_t ll = token >> 4;
    if (ll == 15) {
      uint32_t x;
      do {
        if (ip >= end) { *first8 = seen; return stop; }
        x = in[ip++];
        ll += x;
      } while (x == 255 && ll < (1u << 24));
    }
    if (ll > end - ip) { *first8 = seen; return stop; }
    ip += ll;
    if (ip < end) {  // the last sequence of a block is literals only
      if (end - ip < 2) { *first8 = seen; return stop; }
      ip += 2;
      if ((token & 15u) == 15u) {
        uint32_t x, ml = 0;
        do {
          if (ip >= end) { *first8 = seen; return stop; }
          x = in[ip++];
          ml += x;
        } while (x == 255 && ml < (1u << 24));
      }
    }
  }
  *first8 = seen;
  return ip;
}

// One workgroup per block.  The token chain of a block is serial, but LZ4 streams re-synchronise: a walk that starts at a wrong
// position lands on a true token position after a few sequences and stays on the chain from there.  So the block is cut into
// 256 segments; every lane walks its segment from a GUESSED start (the segment boundary), then from the position the lane
// before it left its own segment at, and again while that position keeps changing.  Lane 0 starts at the true position 0, so
// lane k is final after at most k + 1 rounds whatever the bytes are (the worst case is the serial walk); with
// re-synchronisation everything is final after two or three rounds of ~1/256 of the block each.  Walk errors of a round that
// gets repeated mean nothing; the errors of the last round are the block's.
typedef u32x4 u32x4_any __attribute__((aligned(1)));

constexpr uint32_t kParseLanes = 256;   // segments (= threads) per block: 4 waves (512: slower, the exchange rounds cost more than the shorter walks save)

template <bool IN_LDS>
__global__ __launch_bounds__(kParseLanes) void lz4_parse(Lz4Args a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_block[];
  __shared__ uint32_t s_wave[kParseLanes / 64];     // per-wave maximum / sum of the exchange in progress
  __shared__ uint32_t s_wave2[kParseLanes / 64];
  __shared__ uint16_t s_known[kParseLanes][8];      // every lane's look-up table, as distances from its segment boundary
  __shared__ uint32_t s_true[kParseLanes];          // chain follow: where the chain enters each segment (or `end`)
  const uint32_t bi = blockIdx.x;
  const uint32_t L = threadIdx.x, lane = L & 63u, wave = L >> 6;
  const Lz4BlockDev b = a.blocks[bi];
  const uint32_t block_max = a.buffers[b.buffer].block_max;
  if (b.stored) {  // the block holds its bytes as they are (uniform)
    if (L == 0) {
      a.block_out_size[bi] = b.comp_size <= block_max ? b.comp_size : 0u;
      a.block_nseq[bi] = 0;
      if (b.comp_size > block_max) lz4_fail(a.status);
    }
    return;
  }
  gptr<const uint8_t> in = GC<uint8_t>(a.comp);
  if (IN_LDS) {
    // the walk is a chain of dependent byte loads: from LDS they cost a fraction of an L2 round trip.  16 bytes per lane and
    // step, whatever the alignment (the compressed body is followed by >= 64 readable bytes)
    for (uint32_t i = L * 16; i < b.comp_size; i += kParseLanes * 16)
      *reinterpret_cast<u32x4*>(s_block + i) = *(gptr<const u32x4_any>)(in + b.comp_off + i);
    __syncthreads();
  }
  const uint32_t seg = (b.comp_size + kParseLanes - 1) / kParseLanes;
  const uint32_t cap = seg / 3 + 2;               // a sequence that is not the block's last takes >= 3 bytes
  gptr<u32x4> seq = GM<u32x4>(a.seq) + b.seq_base + L * cap;
  gptr<uint32_t> seq_off = GM<uint32_t>(a.seq_off) + b.seq_base + L * cap;
  // positions are block-relative when the bytes come from LDS, body-relative otherwise
  const uint32_t origin = IN_LDS ? 0u : b.comp_off;
  const uint32_t end = origin + b.comp_size;
  const uint32_t seg_end = origin + (L + 1) * seg < end ? origin + (L + 1) * seg : end;
  uint32_t start = origin + L * seg < end ? origin + L * seg : end;
  const uint32_t seg_start = start;
  // a walk that stores nothing: where it leaves the segment, and which of the first 8 positions it has tokens at
  auto probe = [&](uint32_t from, uint32_t* first8) -> uint32_t {
    uint32_t seen = 0;
    const uint32_t e = IN_LDS ? lz4_probe((lptr)s_block, from, seg_end, end, seg_start, &seen) : lz4_probe(in, from, seg_end, end, seg_start, &seen);
    if (first8) *first8 = seen;
    return e;
  };
  uint32_t exit_at = start < seg_end ? probe(start, nullptr) : start;   // round 0: every lane from its boundary guess
  bool have_tables = false;   // uniform
  uint32_t known[8];
#pragma unroll
  for (int g = 0; g < 8; g++) known[g] = 0;
  bool need = false;
  uint32_t rounds = 0;
  for (uint32_t round = 0; round < kParseLanes + 2; round++) {   // lane k is final after <= k + 1 rounds
    rounds++;
    if (round % 6 == 5) {
      // Still not settled.  Text falls in step within a few sequences, so this is regular numeric data: 3-6-byte sequences
      // in a fixed rhythm, a guessed walk stays out of step for the whole block and the truth advances one lane per round
      // (an exchange with two barriers each).  The chain enters a segment at its first token at or after the boundary, i.e.
      // within one sequence length of it -- so every lane now tabulates where a walk leaves its segment when it starts at
      // the boundary, 1 byte later, ... 7 bytes later.  Walks from different starts share their tail: a start that an
      // earlier walk has a token at gets that walk's answer (a rhythm of p bytes costs p walks, not 8).
      if (!have_tables) {
        uint32_t done = 0;
#pragma unroll
        for (uint32_t g = 0; g < 8; g++) {
          const uint32_t from = seg_start + g;
          if (from >= seg_end) {
            known[g] = from;
          } else if (!((done >> g) & 1u)) {
            uint32_t m = 0;
            const uint32_t e = probe(from, &m);
            m &= ~((1u << g) - 1u);   // tokens at or after this start
            done |= m;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++)
              if ((m >> j) & 1u) known[j] = e;
          }
        }
#pragma unroll
        for (uint32_t g = 0; g < 8; g++) {   // exits are < 64 KiB past the boundary (the block is), 0xFFFF = not representable
          const uint32_t d = known[g] - seg_start;
          s_known[L][g] = static_cast<uint16_t>(d < 0xFFFFu ? d : 0xFFFFu);
        }
        have_tables = true;
      }
      // ONE lane follows the chain through the tables, a hop per segment and no barrier, as far as they reach
      s_true[L] = end;
      __syncthreads();
      if (L == 0) {
        uint32_t pos = origin;
        while (pos < end) {
          const uint32_t k = (pos - origin) / seg;
          s_true[k] = pos;
          const uint32_t k0 = origin + k * seg, delta = pos - k0;
          if (delta >= 8) break;                       // a real walk is needed from here: the rounds take over again
          const uint32_t d = s_known[k][delta];
          if (d == 0xFFFFu) break;
          pos = k0 + d;
        }
      }
      __syncthreads();
      const uint32_t t = s_true[L];
      if (t != end && t != start) {
        start = t;
        need = true;
      }
    }
    if (need) {
      const uint32_t delta = start - seg_start;
      if (start >= seg_end) {
        exit_at = start;   // the chain jumps over this segment
      } else if (have_tables && start >= seg_start && delta < 8) {
        exit_at = delta == 0 ? known[0] : delta == 1 ? known[1] : delta == 2 ? known[2] : delta == 3 ? known[3] : delta == 4 ? known[4]
                  : delta == 5 ? known[5] : delta == 6 ? known[6] : known[7];
      } else {
        exit_at = probe(start, nullptr);
      }
    }
    // the chain enters this lane's segment where the lanes before it left theirs: the furthest exit so far (a sequence
    // that spans many segments leaves the lanes in between with nothing; taking the running maximum tells all of them in
    // one round instead of one lane per round).  Inclusive maximum inside the wave, the waves before it through LDS.
    uint32_t reach = exit_at;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(reach, d, 64);
      if (lane >= static_cast<uint32_t>(d) && o > reach) reach = o;
    }
    if (lane == 63) s_wave[wave] = reach;
    __syncthreads();
    uint32_t before = origin;
    for (uint32_t v = 0; v < wave; v++) before = s_wave[v] > before ? s_wave[v] : before;
    uint32_t from = __shfl_up(reach, 1, 64);
    if (lane == 0 || before > from) from = before;
    need = from != start;
    start = from;
    if (!__syncthreads_or(need ? 1 : 0)) break;   // also: s_wave may be written again
  }
  // every start is final: the one walk that stores its descriptors (positions relative to the lane's first output byte)
  Lz4Walk w;
  w.nseq = 0;
  w.olen = 0;
  w.ok = true;
  if (start < seg_end) {
    if (IN_LDS) w = lz4_walk<true>((lptr)s_block, start, seg_end, end, block_max, seq, seq_off, cap, b.comp_off);
    else w = lz4_walk<true>(in, start, seg_end, end, block_max, seq, seq_off, cap, 0u);
  }
  if (!w.ok) lz4_fail(a.status);
  if (L == 0) {   // how well the speculation worked (mi_scan_stats)
    atomicMax(&a.round_left[38], rounds);
    atomicAdd(&a.round_left[39], rounds);
    atomicAdd(&a.round_left[37], 1u);
  }
  // where each lane's sequences and output bytes begin inside the block: lz4_expand reads the lanes' slices as they are
  uint32_t seq_before = w.nseq, out_before = w.olen;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t s1 = __shfl_up(seq_before, d, 64), s2 = __shfl_up(out_before, d, 64);
    if (lane >= static_cast<uint32_t>(d)) {
      seq_before += s1;
      out_before += s2;
    }
  }
  if (lane == 63) {
    s_wave[wave] = seq_before;
    s_wave2[wave] = out_before;
  }
  const bool any_bad = __syncthreads_or(w.ok ? 0 : 1) != 0;
  uint32_t total_seq = 0, total_out = 0;
  for (uint32_t v = 0; v < kParseLanes / 64; v++) {
    if (v < wave) {
      seq_before += s_wave[v];
      out_before += s_wave2[v];
    }
    total_seq += s_wave[v];
    total_out += s_wave2[v];
  }
  const bool all_ok = !any_bad && total_out <= block_max;
  out_before -= w.olen;
  a.lane_out[static_cast<size_t>(bi) * kParseLanes + L] = all_ok ? out_before : 0u;
  a.lane_nseq[static_cast<size_t>(bi) * kParseLanes + L] = all_ok ? w.nseq : 0u;
  if (!all_ok && total_out > block_max && L == 0) lz4_fail(a.status);
  if (L == 0) {
    a.block_out_size[bi] = all_ok ? total_out : 0u;
    a.block_nseq[bi] = all_ok ? total_seq : 0u;
  }
}

#include "kernels_zstd.inl"

__global__ __launch_bounds__(64) void lz4_layout(Lz4Args a) {
  const uint32_t u = blockIdx.x * 64 + threadIdx.x;
  if (u >= a.n_buffers) return;
  const Lz4BufferDev f = a.buffers[u];
  uint64_t at = f.out_off;
  for (uint32_t k = 0; k < f.n_blocks; k++) {
    a.block_out_base[f.first_block + k] = at;
    at += a.block_out_size[f.first_block + k];
  }
  const bool ok = at - f.out_off == f.out_len;
  a.buffer_ok[u] = ok ? 1u : 0u;
  if (!ok) lz4_fail(a.status);  // "Expected decompressed size of N bytes but got M bytes" (base_stream_reader.cpp:24-29)
}

typedef uint64_t u64_any __attribute__((aligned(1)));

// Output-centric: a thread produces 4 consecutive link words (one 16-byte store, consecutive lanes = consecutive
// addresses).  The block's sequences lie in the 256 slices lz4_parse's lanes wrote, with the output position each slice
// begins at: the thread finds its slice in that table (LDS), its sequence inside the slice by binary search (<= 7 probes)
// and walks on from there, slice to slice.  A thread per SEQUENCE was 3x slower: its stores were scattered 4-byte words and
// every wave ran as long as its longest match.
__global__ __launch_bounds__(kBlockThreads) void lz4_expand(Lz4Args a) {
  static_assert(kBlockThreads == static_cast<int>(kParseLanes), "one table entry per thread");
  __shared__ uint32_t s_lane_out[kParseLanes + 1];
  __shared__ uint32_t s_lane_n[kParseLanes];
  const uint32_t bi = blockIdx.x;
  const Lz4BlockDev b = a.blocks[bi];
  if (!a.buffer_ok[b.buffer]) return;  // uniform
  const uint64_t base = a.block_out_base[bi];      // offset in the decompressed body
  const uint64_t buffer_lo = a.buffers[b.buffer].out_off;
  const uint32_t n_out = a.block_out_size[bi];
  gptr<const uint8_t> in = GC<uint8_t>(a.comp);
  gptr<uint32_t> link = GM<uint32_t>(a.link);
  if (b.stored) {
    for (uint32_t i = threadIdx.x; i < b.comp_size; i += kBlockThreads) link[base + i] = kLinkKnown | in[b.comp_off + i];
    return;
  }
  if (a.block_nseq[bi] == 0) return;
  s_lane_out[threadIdx.x] = a.lane_out[static_cast<size_t>(bi) * kParseLanes + threadIdx.x];
  s_lane_n[threadIdx.x] = a.lane_nseq[static_cast<size_t>(bi) * kParseLanes + threadIdx.x];
  if (threadIdx.x == 0) s_lane_out[kParseLanes] = n_out;
  __syncthreads();
  const uint32_t cap = b.seq_cap / kParseLanes;   // slice stride, as lz4_parse / zstd_entropy wrote them
  gptr<const u32x4> seq0 = GC<u32x4>(a.seq) + b.seq_base;
  gptr<const uint32_t> off0 = GC<uint32_t>(a.seq_off) + b.seq_base;
  const bool aligned = (base & 3u) == 0;
  bool bad = false;
  // a thread owns kChunk consecutive output bytes: ONE search, then it walks the sequences forward (each descriptor is loaded
  // once); its 16-byte stores fill whole cache lines over the chunk
  constexpr uint32_t kChunk = 256;
  for (uint32_t c0 = threadIdx.x * kChunk; c0 < n_out; c0 += kBlockThreads * kChunk) {
    // the last slice that begins at or before c0: empty slices begin where the next one does, so this one is not empty
    uint32_t lo = 0, hi = kParseLanes;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (s_lane_out[mid] <= c0) lo = mid; else hi = mid;
    }
    uint32_t k = lo, lane_base = s_lane_out[k], nk = s_lane_n[k];
    gptr<const u32x4> seq = seq0 + k * cap;
    gptr<const uint32_t> seq_off = off0 + k * cap;
    // ... and the last sequence of it that begins at or before c0
    uint32_t si = 0;
    {
      uint32_t slo = 0, shi = nk;
      const uint32_t rel = c0 - lane_base;
      while (shi - slo > 1) {
        const uint32_t mid = (slo + shi) >> 1;
        if (seq[mid].x <= rel) slo = mid; else shi = mid;
      }
      si = slo;
    }
    // ZSTD: an offset may still name the repeat-offset history its slice started from (zstd_layout wrote it per slice)
    gptr<const u32x4> rep = GC<u32x4>(a.rep_state) + static_cast<size_t>(bi) * kParseLanes;
    auto offset_of = [&](uint32_t raw, uint32_t slice) -> uint32_t {
      if (!(raw >> 31)) return raw;
      const u32x4 h = rep[slice];
      return zstd::RepResolve(raw, h.x, h.y, h.z);
    };
    u32x4 d = seq[si];
    uint32_t offset = offset_of(seq_off[si], k);
    const uint32_t c1 = c0 + kChunk < n_out ? c0 + kChunk : n_out;
#pragma clang loop unroll(disable)
    for (uint32_t p0 = c0; p0 < c1; p0 += 4) {
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t p = p0 + q;
      w[q] = kLinkUntouched;
      if (p >= n_out) continue;
      while (p - lane_base >= d.x + d.z + d.w) {   // on to the next sequence (p < n_out: there is one)
        si++;
        if (si >= nk) {                            // ... of the next slice that has any
          do { k++; } while (k + 1 < kParseLanes && s_lane_n[k] == 0);
          lane_base = s_lane_out[k];
          nk = s_lane_n[k];
          seq = seq0 + k * cap;
          seq_off = off0 + k * cap;
          si = 0;
          if (nk == 0) break;                      // cannot happen for p < n_out; never spin
        }
        d = seq[si];
        offset = offset_of(seq_off[si], k);
      }
      const uint32_t r = p - lane_base - d.x;
      if (r < d.z) {
        w[q] = kLinkKnown | in[d.y + r];
      } else {
        const uint64_t m_at = base + lane_base + d.x + d.z;
        const uint32_t i = r - d.z;
        if (i >= d.w || offset == 0 || offset > m_at - buffer_lo) {   // reaches in front of the buffer: not a frame an encoder writes
          bad = true;
          w[q] = kLinkKnown;
        } else {
          // an overlapping match (offset < length: a run) repeats its first `offset` bytes: every byte links straight into
          // that period instead of to the byte `offset` before it, or a run of n bytes would be a chain n / offset deep
          w[q] = static_cast<uint32_t>(m_at - offset) + (i < offset ? i : i % offset);
        }
      }
    }
    if (aligned && p0 + 4 <= n_out) {
      u32x4 v;
      v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
      *(gptr<u32x4>)(link + base + p0) = v;
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (p0 + q < n_out) link[base + p0 + q] = w[q];
    }
    }  // quads of the chunk
  }
  if (bad) lz4_fail(a.status);
}

// Before the global rounds: every tile (8 KiB of output = 32 KiB of link words in a workgroup's LDS) follows the links
// that stay INSIDE the tile, in LDS, until nothing moves.  The deep chains of columnar data are local -- a value copies its
// high bytes from the value before it, thousands of times in a row -- and a hop in LDS costs a fraction of a hop through
// L2 (three hops per global round instead of one made the rounds slower, not fewer: the gathers are what a round costs).
// Afterwards a chain crosses at least one tile boundary per hop, so the global rounds see depths of tiles, not of bytes.
__global__ __launch_bounds__(kBlockThreads) void lz4_resolve_local(Lz4Args a) {
  __shared__ uint32_t s_link[4 * kLocalTileQuads];
  gptr<uint32_t> link = GM<uint32_t>(a.link);
  const uint64_t nquads = (a.out_size + 3) / 4;
  const uint32_t ntiles = static_cast<uint32_t>((nquads + kLocalTileQuads - 1) / kLocalTileQuads);
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint64_t q0 = static_cast<uint64_t>(tile) * kLocalTileQuads;
    const uint32_t nq = static_cast<uint32_t>(q0 + kLocalTileQuads < nquads ? kLocalTileQuads : nquads - q0);
    const uint32_t lo = static_cast<uint32_t>(4 * q0);          // first byte position of the tile
    bool open = false;
    for (uint32_t q = threadIdx.x; q < nq; q += kBlockThreads) {
      const u32x4 v = *(gptr<const u32x4>)(link + 4 * (q0 + q));
      *reinterpret_cast<u32x4*>(&s_link[4 * q]) = v;
      open |= !((v.x & v.y & v.z & v.w) >> 31);
    }
    if (!__syncthreads_or(open ? 1 : 0)) continue;   // nothing but known bytes (or untouched words): uniform
    for (int round = 0; round < 20; round++) {        // chain depth inside a tile < 8192: 14 rounds at most
      bool moved = false;
      for (uint32_t i = threadIdx.x; i < 4 * nq; i += kBlockThreads) {
        const uint32_t s = s_link[i];
        if ((s >> 31) || s < lo) continue;            // known, or the source lies in an earlier tile
        const uint32_t u = s_link[s - lo];             // s < position of i: inside this tile
        if (u != s) {
          s_link[i] = u;
          moved = true;
        }
      }
      if (!__syncthreads_or(moved ? 1 : 0)) break;
    }
    gptr<uint8_t> mark = GM<uint8_t>(a.mark);
    for (uint32_t q = threadIdx.x; q < nq; q += kBlockThreads) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(&s_link[4 * q]);
      *(gptr<u32x4>)(link + 4 * (q0 + q)) = v;
      // what is still open points into an earlier tile: those target words are the SKELETON the global rounds work on
      if (!(v.x >> 31)) mark[v.x] = 1;
      if (!(v.y >> 31)) mark[v.y] = 1;
      if (!(v.z >> 31)) mark[v.z] = 1;
      if (!(v.w >> 31)) mark[v.w] = 1;
    }
    __syncthreads();
  }
}

// The marked words that are themselves still open, as a list.  A marked word's own target was marked by it (it is an open
// word of its tile), so the list is closed under "follow the link": pointer jumping over the list alone resolves it.
__global__ __launch_bounds__(kBlockThreads) void lz4_collect(Lz4Args a) {
  // a workgroup gathers the entries of one 8 KiB tile in LDS and claims its slice of the list with ONE atomic: the counter is
  // one address for the whole launch, and same-address atomics are served one after the other (~20 ns each)
  __shared__ uint32_t s_pos[4 * kLocalTileQuads];
  __shared__ uint32_t s_n, s_base;
  gptr<const uint32_t> link = GC<uint32_t>(a.link);
  gptr<const uint32_t> mark4 = GC<uint32_t>(a.mark);   // four marks per word
  gptr<uint32_t> skel = GM<uint32_t>(a.skel);
  const uint64_t nquads = (a.out_size + 3) / 4;
  const uint32_t ntiles = static_cast<uint32_t>((nquads + kLocalTileQuads - 1) / kLocalTileQuads);
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint64_t q0 = static_cast<uint64_t>(tile) * kLocalTileQuads;
    const uint32_t nq = static_cast<uint32_t>(q0 + kLocalTileQuads < nquads ? kLocalTileQuads : nquads - q0);
    for (uint32_t q = threadIdx.x; q < nq; q += kBlockThreads) {
      const uint32_t m = mark4[q0 + q];
      if (!m) continue;
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (((m >> (8 * k)) & 0xFFu) && !(link[4 * (q0 + q) + k] >> 31)) s_pos[atomicAdd(&s_n, 1u)] = static_cast<uint32_t>(4 * (q0 + q) + k);
    }
    __syncthreads();
    const uint32_t n = s_n;
    if (n) {   // uniform
      if (threadIdx.x == 0) s_base = atomicAdd(&a.round_left[kSkelCount], n);
      __syncthreads();
      const uint32_t base = s_base;
      for (uint32_t i = threadIdx.x; i < n; i += kBlockThreads) skel[base + i] = s_pos[i];
    }
    __syncthreads();
  }
}

// One round of pointer jumping over the skeleton.
__global__ __launch_bounds__(kBlockThreads) void lz4_resolve_skeleton(Lz4Args a, int round) {
  if (round > 0 && a.round_left[round - 1] == 0) return;  // the previous round left nothing (round_left[round] stays 0)
  gptr<uint32_t> link = GM<uint32_t>(a.link);
  gptr<const uint32_t> skel = GC<uint32_t>(a.skel);
  const uint32_t n = a.round_left[kSkelCount];
  bool left = false;
  for (uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x; i < n; i += gridDim.x * kBlockThreads) {
    const uint32_t j = skel[i];
    const uint32_t s = link[j];
    if (s >> 31) continue;
    // a skeleton round is short and latency-bound (a launch, one dependent gather, a store): following kSkelHops links in it
    // costs little more and divides the chain depth by kSkelHops + 1 instead of 2 -- fewer rounds.  (Over ALL words, where
    // the gathers are what a round costs, more hops per round were slower.)  A value read before this launch or during it
    // is on the chain either way.
    uint32_t u = __builtin_nontemporal_load(link + s);
#pragma unroll
    for (int h = 1; h < kSkelHops; h++)
      if (!(u >> 31)) u = __builtin_nontemporal_load(link + u);
    link[j] = u;
    if (!(u >> 31)) left = true;
  }
  if (__syncthreads_or(left ? 1 : 0) && threadIdx.x == 0) atomicAdd(&a.round_left[round], 1u);
}

// The known bytes leave the link words for the decompressed body; words still holding kLinkUntouched belong to bytes the
// K8 kernels did not produce (raw buffers were copied, padding was zeroed).
__global__ __launch_bounds__(kBlockThreads) void lz4_emit(Lz4Args a) {
  gptr<const uint32_t> link = GC<uint32_t>(a.link);
  gptr<uint8_t> out = GM<uint8_t>(a.out);
  const uint64_t nquads = (a.out_size + 3) / 4;
  bool bad = false;
  for (uint64_t q = static_cast<uint64_t>(blockIdx.x) * kBlockThreads + threadIdx.x; q < nquads; q += static_cast<uint64_t>(gridDim.x) * kBlockThreads) {
    u32x4 v = *(gptr<const u32x4>)(link + 4 * q);
    // a word that is still open names a skeleton word, and those are all known by now: the last hop happens here
    if (!(v.x >> 31)) v.x = link[v.x];
    if (!(v.y >> 31)) v.y = link[v.y];
    if (!(v.z >> 31)) v.z = link[v.z];
    if (!(v.w >> 31)) v.w = link[v.w];
    bad |= !((v.x & v.y & v.z & v.w) >> 31);
    const bool t0 = v.x != kLinkUntouched, t1 = v.y != kLinkUntouched, t2 = v.z != kLinkUntouched, t3 = v.w != kLinkUntouched;
    if (t0 && t1 && t2 && t3 && 4 * q + 4 <= a.out_size) {
      *(gptr<uint32_t>)(out + 4 * q) = (v.x & 0xFFu) | ((v.y & 0xFFu) << 8) | ((v.z & 0xFFu) << 16) | ((v.w & 0xFFu) << 24);
    } else {
      if (t0 && 4 * q + 0 < a.out_size) out[4 * q + 0] = static_cast<uint8_t>(v.x);
      if (t1 && 4 * q + 1 < a.out_size) out[4 * q + 1] = static_cast<uint8_t>(v.y);
      if (t2 && 4 * q + 2 < a.out_size) out[4 * q + 2] = static_cast<uint8_t>(v.z);
      if (t3 && 4 * q + 3 < a.out_size) out[4 * q + 3] = static_cast<uint8_t>(v.w);
    }
  }
  if (bad) lz4_fail(a.status);   // never silently: a chain the rounds did not finish is an internal error, not data
}

}  // namespace

// a.link must hold kLinkUntouched in every word (0xFF bytes), a.round_left zeros, a.out zeros where padding is expected.
hipError_t LaunchLz4Decompress(const Lz4Args& a, int num_cus, hipStream_t stream) {
  MI_DROP_STALE_ERROR();
  if (a.n_blocks == 0) return hipSuccess;
  static const int dbg_skip = std::getenv("MI_K8_DIAG_SKIP") ? std::atoi(std::getenv("MI_K8_DIAG_SKIP")) : 0;   // DIAG-TEMP
  if (dbg_skip & 1) return hipSuccess;   // DIAG-TEMP
  if (a.zblocks) {
    hipLaunchKernelGGL(zstd_entropy, dim3(a.n_blocks), dim3(kZstdThreads), 0, stream, a);
    hipLaunchKernelGGL(zstd_layout, dim3(a.n_buffers), dim3(64), 0, stream, a);
  } else {
    // compressed blocks below 64 KiB (64 KiB is the default block size of every writer) are walked from an LDS copy
    static const bool force_global = std::getenv("MI_LZ4_PARSE_GLOBAL") != nullptr;   // tests: the variant for blocks too large for LDS
    if (!force_global && a.max_block_comp + 32u + 6144u <= (64u << 10))   // 64 KiB of LDS per workgroup without opting in to more (the kernel's own tables: 5 KiB)
      hipLaunchKernelGGL(lz4_parse<true>, dim3(a.n_blocks), dim3(kParseLanes), ((a.max_block_comp + 15u) & ~15u) + 16u, stream, a);
    else
      hipLaunchKernelGGL(lz4_parse<false>, dim3(a.n_blocks), dim3(kParseLanes), 0, stream, a);
    hipLaunchKernelGGL(lz4_layout, dim3((a.n_buffers + 63) / 64), dim3(64), 0, stream, a);
  }
  hipLaunchKernelGGL(lz4_expand, dim3(a.n_blocks), dim3(kBlockThreads), 0, stream, a);
  // chains only run backwards inside one buffer, and after lz4_resolve_local every hop that is left crosses a boundary of
  // its 8 KiB tiles: depth <= tiles the longest buffer touches, rounds <= log5(depth) + 1 (5 for a 3 MB buffer)
  const uint64_t depth = a.max_buffer_len / (4 * kLocalTileQuads) + 2;
  int rounds = 2;   // a round divides the depth by kSkelHops + 1
  for (uint64_t reach = kSkelHops + 1; rounds < 33 && reach < depth; reach *= kSkelHops + 1) rounds++;
  const uint64_t want = ((a.out_size + 3) / 4 + kBlockThreads - 1) / kBlockThreads;
  const uint32_t grid = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(want, static_cast<uint64_t>(num_cus) * 16)));
  const uint64_t nlocal = ((a.out_size + 3) / 4 + kLocalTileQuads - 1) / kLocalTileQuads;
  hipLaunchKernelGGL(lz4_resolve_local, dim3(static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(nlocal, static_cast<uint64_t>(num_cus) * 16)))),
                     dim3(kBlockThreads), 0, stream, a);
  hipLaunchKernelGGL(lz4_collect, dim3(static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(nlocal, static_cast<uint64_t>(num_cus) * 16)))),
                     dim3(kBlockThreads), 0, stream, a);
  for (int r = 0; r < rounds; r++) hipLaunchKernelGGL(lz4_resolve_skeleton, dim3(static_cast<uint32_t>(num_cus) * 4), dim3(kBlockThreads), 0, stream, a, r);
  hipLaunchKernelGGL(lz4_emit, dim3(grid), dim3(kBlockThreads), 0, stream, a);
  return hipGetLastError();
}

}  // namespace device
}  // namespace miarrow
