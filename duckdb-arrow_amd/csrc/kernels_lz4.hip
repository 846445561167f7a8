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
//   1. lz4_parse_dp one workgroup (256 lanes) per block, tokens only (no data is copied): one descriptor per sequence
//                   {output position, literal source, literal length, match length} + offset in the lane's slice, and the
//                   block's decompressed size.  Every lane tabulates for every byte of its segment where a walk with a token
//                   there leaves the segment (one backward pass in LDS), one lane follows the chain through the tables, then
//                   every lane walks once from its true entry and stores.  Blocks too large for the tables keep lz4_parse,
//                   the speculative formulation (guessed entries, repeated until they agree).
//                   (ZSTD: zstd_entropy produces the same descriptors from the Huffman / FSE streams.)
//   2. lz4_layout   one lane per buffer: first output byte of each of its blocks (running sum), and the check the
//                   reference makes after decompressing: the sizes must add up to the declared uncompressed length.
//                   (ZSTD: zstd_layout, which also settles the repeat offsets.)
//   3. k8_chunk_map numbers the CHUNKS (<= 8 KiB of one block's output) of the launch set: the later kernels run one
//                   workgroup per chunk.
//   4. k8_expand_local
//                   a chunk's sequences are dealt to the threads one each; every decompressed byte gets a 32-bit LINK word in
//                   LDS -- a literal's word holds the byte itself, a match byte's word the position it copies from (always an
//                   earlier byte of the buffer; an overlapping match links straight into its first period) -- and the links
//                   that stay inside the chunk are followed right there until nothing moves.  The words reach HBM once;
//                   targets in earlier chunks are marked.
//   5. lz4_collect / lz4_resolve_skeleton
//                   pointer jumping over the SKELETON only -- the still-open words that other chunks' open words point at --
//                   24 hops per round, 3 rounds launched blindly (a round that finds nothing left makes the later ones return
//                   at once).
//   6. k8_emit      the last hop of everything outside the skeleton, then the bytes leave the link words for the
//                   decompressed body; a word still open here is an error, never silent data.
// HBM traffic per decompressed byte: 4 B of link words written + 1 B of marks cleared by the batch's one memset + 4 B read and
// 1 B written by k8_emit (+ the descriptors, 20 B per sequence, and the skeleton's few words).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

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

// A link word: bit 31 set = the byte is known and sits in the low 8 bits;
// bit 31 clear = the position this byte copies from.  One word is one atomic message: no second array to keep in step.
constexpr uint32_t kLinkKnown = 0x80000000u;
constexpr uint32_t kSkelCount = 36;            // round_left[36]: entries of the skeleton list
constexpr int kSkelHops = 24;                  // links followed per skeleton round (a lane stops at the first known word): 3 rounds for a 3 MB buffer, 5 with 4 hops
constexpr uint32_t kLocalTileQuads = 2048;     // lz4_collect: 8 KiB of output per workgroup

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

constexpr uint32_t kDpMaxComp = 50u << 10;   // lz4_parse_dp's blocks: 3 x 50 KiB + the kernel's own tables < 160 KiB of LDS
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
  if (a._pad != 0 && b.comp_size <= kDpMaxComp) return;   // uniform: lz4_parse_dp owns this block (set by the launcher)
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
  // where each lane's sequences and output bytes begin inside the block: k8_expand_local reads the lanes' slices as they are
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


// ---------------------------------------------------------------------------------------------------------------------
// lz4_parse_dp: the same result as lz4_parse (the lanes' slices of sequence descriptors, their output positions, the block's
// size) without speculation.  lz4_parse guesses where the token chain enters a lane's segment and repeats until the guesses
// agree; regular numeric data never falls in step, its 8-entry tables cover only entries within 8 bytes of the boundary, and
// a block with one longer sequence per segment fell back to one lane per round: 249 rounds for the worst block of an SF10
// lineitem scan, and the launch lasts as long as its slowest block (255 us against ~6 rounds on average).
// Here every lane tabulates, for EVERY byte position p of its segment, where a walk that has a token at p leaves the segment:
// one backward pass -- exit[p] = next(p) if the sequence at p ends past the segment, else exit[next(p)], already known -- of
// one sequence header per position, no dependence between lanes.  Then ONE lane follows the chain through the tables, a hop
// per segment it touches (<= 256 dependent LDS reads), and every lane makes its one storing walk from its true entry.
// Cost is a function of the block's size alone.  LDS: the block's bytes + 2 bytes per byte for the tables, so blocks whose
// compressed size exceeds kDpMaxComp (and frames with > 64 KiB blocks) keep lz4_parse.
constexpr uint32_t kDpInvalid = 0xFFFFu;

// where the sequence whose token is at `ip` ends (= the next token), or kDpInvalid when the bytes at ip are no sequence
__device__ __forceinline__ uint32_t lz4_next_token(lptr in, uint32_t ip, uint32_t token, uint32_t end) {
  ip++;
  uint32_t ll = token >> 4;
  if (ll == 15) {
    uint32_t x;
    do {
      if (ip >= end) return kDpInvalid;
      x = in[ip++];
      ll += x;
    } while (x == 255 && ll < (1u << 24));
  }
  if (ll > end - ip) return kDpInvalid;
  ip += ll;
  if (ip == end) return end;       // the block's last sequence: literals only
  if (end - ip < 2) return kDpInvalid;
  ip += 2;
  if ((token & 15u) == 15u) {
    uint32_t x, ml = 0;
    do {
      if (ip >= end) return kDpInvalid;
      x = in[ip++];
      ml += x;
    } while (x == 255 && ml < (1u << 24));
  }
  return ip <= end ? ip : kDpInvalid;
}

__global__ __launch_bounds__(kParseLanes) void lz4_parse_dp(Lz4Args a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];   // [block bytes, padded to 16][uint16 exit per byte]
  __shared__ uint32_t s_wave[kParseLanes / 64];
  __shared__ uint32_t s_wave2[kParseLanes / 64];
  __shared__ uint32_t s_true[kParseLanes];
  __shared__ uint32_t s_chain_ok;
  const uint32_t bi = blockIdx.x;
  const uint32_t L = threadIdx.x, lane = L & 63u, wave = L >> 6;
  const Lz4BlockDev b = a.blocks[bi];
  const uint32_t block_max = a.buffers[b.buffer].block_max;
  if (b.stored) {
    if (L == 0) {
      a.block_out_size[bi] = b.comp_size <= block_max ? b.comp_size : 0u;
      a.block_nseq[bi] = 0;
      if (b.comp_size > block_max) lz4_fail(a.status);
    }
    return;
  }
  if (b.comp_size > kDpMaxComp) return;   // uniform: lz4_parse owns this block
  gptr<const uint8_t> in = GC<uint8_t>(a.comp);
  const uint32_t comp_pad = (b.comp_size + 15u) & ~15u;
  uint8_t* s_block = s_dyn;
  uint16_t* s_exit = reinterpret_cast<uint16_t*>(s_dyn + comp_pad + 16);
  for (uint32_t i = L * 16; i < b.comp_size; i += kParseLanes * 16)
    *reinterpret_cast<u32x4*>(s_block + i) = *(gptr<const u32x4_any>)(in + b.comp_off + i);
  __syncthreads();
  const uint32_t end = b.comp_size;
  const uint32_t seg = (b.comp_size + kParseLanes - 1) / kParseLanes;
  const uint32_t cap = seg / 3 + 2;
  const uint32_t seg_start = L * seg < end ? L * seg : end;
  const uint32_t seg_end = (L + 1) * seg < end ? (L + 1) * seg : end;
  // backward pass over the lane's own segment, four positions at a time: their tokens are independent LDS reads (the one
  // latency a position costs), their table entries are settled in order (an entry may name one of the same group)
  for (uint32_t p = seg_end; p > seg_start;) {
    const uint32_t cnt = p - seg_start < 4u ? p - seg_start : 4u;
    uint32_t tok[4], nx[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) tok[j] = j < cnt ? ((lptr)s_block)[p - 1 - j] : 0u;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) nx[j] = j < cnt ? lz4_next_token((lptr)s_block, p - 1 - j, tok[j], end) : kDpInvalid;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
      if (j < cnt) {
        uint32_t e = nx[j];
        if (e != kDpInvalid && e < seg_end) e = s_exit[e];   // e > this position: written earlier in this pass
        s_exit[p - 1 - j] = static_cast<uint16_t>(e);
      }
    }
    p -= cnt;
  }
  s_true[L] = end;
  if (L == 0) s_chain_ok = 1;
  __syncthreads();
  if (L == 0) {   // the chain: a hop per segment it has a token in
    uint32_t pos = 0, k = 0;
    while (pos < end) {
      while ((k + 1) * seg <= pos) k++;
      s_true[k] = pos;
      const uint32_t e = s_exit[pos];
      if (e == kDpInvalid || e <= pos) { s_chain_ok = 0; break; }
      pos = e;
    }
  }
  __syncthreads();
  const uint32_t start = s_true[L];
  gptr<u32x4> seq = GM<u32x4>(a.seq) + b.seq_base + L * cap;
  gptr<uint32_t> seq_off = GM<uint32_t>(a.seq_off) + b.seq_base + L * cap;
  Lz4Walk w;
  w.nseq = 0;
  w.olen = 0;
  w.ok = s_chain_ok != 0;
  if (w.ok && start < seg_end) w = lz4_walk<true>((lptr)s_block, start, seg_end, end, block_max, seq, seq_off, cap, b.comp_off);
  if (!w.ok) lz4_fail(a.status);
  if (L == 0) {
    atomicMax(&a.round_left[38], 1u);
    atomicAdd(&a.round_left[39], 1u);
    atomicAdd(&a.round_left[37], 1u);
  }
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

// ---------------------------------------------------------------------------------------------------------------------
// Chunks.  The copy stages work on CHUNKS: up to kChunkBytes consecutive output bytes of ONE block (the last chunk of a block
// is shorter).  k8_chunk_map numbers them (a prefix sum over the blocks, whose sizes only the device knows), every later
// kernel is launched with an upper bound of workgroups (one per chunk) and finds its block by binary search.
constexpr uint32_t kChunkBytes = 8192;
constexpr uint32_t kChunkCount = 35;   // round_left[35]: chunks of this launch set

__global__ __launch_bounds__(kBlockThreads) void k8_chunk_map(Lz4Args a) {
  __shared__ uint32_t s_wave[kBlockThreads / 64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < a.n_blocks; b0 += kBlockThreads) {
    const uint32_t i = b0 + threadIdx.x;
    uint32_t c = 0;
    if (i < a.n_blocks && a.buffer_ok[a.blocks[i].buffer]) c = (a.block_out_size[i] + kChunkBytes - 1) / kChunkBytes;
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if (lane >= static_cast<uint32_t>(d)) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t v = 0; v < kBlockThreads / 64; v++) {
      if (v < wave) before += s_wave[v];
      total += s_wave[v];
    }
    if (i < a.n_blocks) a.chunk_base[i] = carry + before + incl - c;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    a.chunk_base[a.n_blocks] = carry;
    a.round_left[kChunkCount] = carry;
  }
}

// the block that owns chunk w: the last one whose first chunk is <= w (blocks without chunks share their successor's base)
__device__ __forceinline__ uint32_t chunk_block(const Lz4Args& a, uint32_t w) {
  uint32_t lo = 0, hi = a.n_blocks;   // chunk_base[lo] <= w < chunk_base[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.chunk_base[mid] <= w) lo = mid; else hi = mid;
  }
  return lo;
}

// Expand + local resolve, one workgroup per chunk.  Every decompressed byte gets a 32-bit LINK word -- a literal's word holds
// the byte itself, a match byte's word the position it copies from (always an earlier byte of the buffer; an overlapping match
// links straight into its first period) -- built in LDS: the chunk's sequences (the slices lz4_parse's / zstd_entropy's lanes
// wrote, with the output position each slice begins at) are dealt to the threads one sequence each, coalesced descriptor
// loads, a sequence writes its own words; sequences longer than kLongSeq bytes (runs) are left to the whole workgroup.  Then
// the links that stay INSIDE the chunk are followed in LDS until nothing moves: the deep chains of columnar data are local (a
// value copies its high bytes from the value before it, thousands of times in a row) and a hop in LDS costs a fraction of a
// hop through L2.  What is still open afterwards points into an earlier chunk: those target words are marked (the SKELETON
// the global rounds work on).  The words leave for HBM once, final or open; nothing is preset or re-read in between.
// (Before: lz4_expand, one workgroup per BLOCK with a serial walk per thread through descriptors in HBM, 4 B per byte written;
// a 4 B per byte memset in front of it; lz4_resolve_local reading and writing the same words again.)
constexpr uint32_t kLongSeq = 128;
constexpr uint32_t kMaxLong = 80;      // > kChunkBytes / (kLongSeq + 1) + 2

__global__ __launch_bounds__(kBlockThreads) void k8_expand_local(Lz4Args a) {
  static_assert(kBlockThreads == static_cast<int>(kParseLanes), "one table entry per thread");
  __shared__ uint32_t s_link[kChunkBytes];
  __shared__ uint32_t s_lane_out[kParseLanes + 1];
  __shared__ uint32_t s_lane_n[kParseLanes];
  __shared__ uint32_t s_pref[kParseLanes + 1];
  __shared__ uint32_t s_wave[kBlockThreads / 64];
  __shared__ uint32_t s_long[kMaxLong];
  __shared__ uint32_t s_nlong;
  const uint32_t w = blockIdx.x;
  if (w >= a.round_left[kChunkCount]) return;   // uniform
  const uint32_t bi = chunk_block(a, w);
  const Lz4BlockDev b = a.blocks[bi];
  const uint64_t base = a.block_out_base[bi];       // offset of the block in the decompressed body
  const uint64_t buffer_lo = a.buffers[b.buffer].out_off;
  const uint32_t n_out = a.block_out_size[bi];
  const uint32_t c0 = (w - a.chunk_base[bi]) * kChunkBytes;
  const uint32_t c1 = c0 + kChunkBytes < n_out ? c0 + kChunkBytes : n_out;
  const uint32_t n = c1 - c0;
  const uint32_t lo = static_cast<uint32_t>(base) + c0;   // position of the chunk's first byte (31-bit positions)
  gptr<const uint8_t> in = GC<uint8_t>(a.comp);
  gptr<uint32_t> link = GM<uint32_t>(a.link);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (b.stored) {   // the block holds its bytes as they are
    for (uint32_t i = tid; i < n; i += kBlockThreads) link[lo + i] = kLinkKnown | in[b.comp_off + c0 + i];
    return;
  }
  s_lane_out[tid] = a.lane_out[static_cast<size_t>(bi) * kParseLanes + tid];
  s_lane_n[tid] = a.lane_nseq[static_cast<size_t>(bi) * kParseLanes + tid];
  if (tid == 0) {
    s_lane_out[kParseLanes] = n_out;
    s_nlong = 0;
  }
  for (uint32_t i = tid; i < kChunkBytes; i += kBlockThreads) s_link[i] = kLinkKnown;
  __syncthreads();
  // the slices that hold the chunk's sequences: from the last one that begins at or before c0 to the last one that begins
  // before c1 (empty slices begin where the next one does, so neither end is an empty slice unless the block is)
  uint32_t kf, kl;
  {
    uint32_t x = 0, y = kParseLanes;
    while (y - x > 1) {
      const uint32_t mid = (x + y) >> 1;
      if (s_lane_out[mid] <= c0) x = mid; else y = mid;
    }
    kf = x;
    x = kf;
    y = kParseLanes;
    while (y - x > 1) {
      const uint32_t mid = (x + y) >> 1;
      if (s_lane_out[mid] < c1) x = mid; else y = mid;
    }
    kl = x;
  }
  const uint32_t nsl = kl - kf + 1;
  // exclusive prefix of the slices' sequence counts
  {
    const uint32_t c = tid < nsl ? s_lane_n[kf + tid] : 0u;
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if (lane >= static_cast<uint32_t>(d)) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t v = 0; v < wave; v++) before += s_wave[v];
    s_pref[tid] = before + incl - c;
    if (tid == kBlockThreads - 1) s_pref[kParseLanes] = before + incl;
    __syncthreads();
  }
  const uint32_t nseq = s_pref[kParseLanes];
  const uint32_t cap = b.seq_cap / kParseLanes;   // slice stride, as lz4_parse / zstd_entropy wrote them
  gptr<const u32x4> seq0 = GC<u32x4>(a.seq) + b.seq_base;
  gptr<const uint32_t> off0 = GC<uint32_t>(a.seq_off) + b.seq_base;
  gptr<const u32x4> rep = GC<u32x4>(a.rep_state) + static_cast<size_t>(bi) * kParseLanes;
  bool bad = false;
  // the words of output bytes [from, to) of sequence d (its first byte is output byte o0 of the block), step `stride`
  auto write_words = [&](const u32x4& d, uint32_t offset, uint32_t o0, uint32_t from, uint32_t to, uint32_t stride) {
    const uint64_t m_at = base + o0 + d.z;   // position of the match's first byte
    const bool bad_offset = offset == 0 || offset > m_at - buffer_lo;   // reaches in front of the buffer: not a frame an encoder writes
    for (uint32_t p = from; p < to; p += stride) {
      const uint32_t r = p - o0;
      uint32_t word;
      if (r < d.z) {
        word = kLinkKnown | in[d.y + r];
      } else if (bad_offset) {
        bad = true;
        word = kLinkKnown;
      } else {
        // an overlapping match (offset < length: a run) repeats its first `offset` bytes: every byte links straight into
        // that period instead of to the byte `offset` before it, or a run of n bytes would be a chain n / offset deep
        const uint32_t i = r - d.z;
        word = static_cast<uint32_t>(m_at - offset) + (i < offset ? i : i % offset);
      }
      s_link[p - c0] = word;
    }
  };
  auto fetch = [&](uint32_t g, u32x4* d, uint32_t* offset, uint32_t* o0) {
    uint32_t x = 0, y = nsl;   // the slice of sequence g: the last one whose prefix is <= g (empty slices share their successor's)
    while (y - x > 1) {
      const uint32_t mid = (x + y) >> 1;
      if (s_pref[mid] <= g) x = mid; else y = mid;
    }
    const uint32_t k = kf + x, idx = g - s_pref[x];
    *d = seq0[k * cap + idx];
    const uint32_t raw = off0[k * cap + idx];
    // ZSTD: an offset may still name the repeat-offset history its slice started from (zstd_layout wrote it per slice)
    if (raw >> 31) {
      const u32x4 h = rep[k];
      *offset = zstd::RepResolve(raw, h.x, h.y, h.z);
    } else {
      *offset = raw;
    }
    *o0 = s_lane_out[k] + d->x;
  };
  for (uint32_t g = tid; g < nseq; g += kBlockThreads) {
    u32x4 d;
    uint32_t offset, o0;
    fetch(g, &d, &offset, &o0);
    const uint32_t len = d.z + d.w, o1 = o0 + len;
    if (len > (1u << 25) || o1 <= c0 || o0 >= c1) continue;   // the edge slices' sequences outside the chunk
    if (len > kLongSeq) {
      const uint32_t j = atomicAdd(&s_nlong, 1u);
      if (j < kMaxLong) {
        s_long[j] = g;
        continue;
      }
    }
    write_words(d, offset, o0, o0 > c0 ? o0 : c0, o1 < c1 ? o1 : c1, 1u);
  }
  __syncthreads();
  const uint32_t nlong = s_nlong < kMaxLong ? s_nlong : kMaxLong;
  for (uint32_t j = 0; j < nlong; j++) {   // uniform: a long sequence is written by all threads
    u32x4 d;
    uint32_t offset, o0;
    fetch(s_long[j], &d, &offset, &o0);
    const uint32_t o1 = o0 + d.z + d.w;
    write_words(d, offset, o0, (o0 > c0 ? o0 : c0) + tid, o1 < c1 ? o1 : c1, kBlockThreads);
  }
  __syncthreads();
  // the links that stay inside the chunk
  for (int round = 0; round < 20; round++) {        // chain depth inside a chunk < 8192: 14 rounds at most
    bool moved = false;
    for (uint32_t i = tid; i < n; i += kBlockThreads) {
      const uint32_t sx = s_link[i];
      if ((sx >> 31) || sx < lo) continue;          // known, or the source lies in an earlier chunk
      const uint32_t u = s_link[sx - lo];            // sx < position of i: inside this chunk
      if (u != sx) {
        s_link[i] = u;
        moved = true;
      }
    }
    if (!__syncthreads_or(moved ? 1 : 0)) break;
  }
  gptr<uint8_t> mark = GM<uint8_t>(a.mark);
  if ((lo & 3u) == 0) {
    for (uint32_t q = tid; 4 * q < n; q += kBlockThreads) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(&s_link[4 * q]);
      if (4 * q + 4 <= n) {
        *(gptr<u32x4>)(link + lo + 4 * q) = v;
      } else {
        for (uint32_t k = 0; 4 * q + k < n; k++) link[lo + 4 * q + k] = s_link[4 * q + k];
      }
      // what is still open points into an earlier chunk: those target words are the SKELETON the global rounds work on
      if (!(v.x >> 31) && 4 * q + 0 < n) mark[v.x] = 1;
      if (!(v.y >> 31) && 4 * q + 1 < n) mark[v.y] = 1;
      if (!(v.z >> 31) && 4 * q + 2 < n) mark[v.z] = 1;
      if (!(v.w >> 31) && 4 * q + 3 < n) mark[v.w] = 1;
    }
  } else {
    for (uint32_t i = tid; i < n; i += kBlockThreads) {
      const uint32_t v = s_link[i];
      link[lo + i] = v;
      if (!(v >> 31)) mark[v] = 1;
    }
  }
  if (bad) lz4_fail(a.status);
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

// The bytes leave the link words for the decompressed body, one workgroup per chunk (only produced bytes are touched: raw
// buffers were copied, padding is nobody's).  A word that is still open names a skeleton word, and those are all known by
// now: the last hop happens here.
__global__ __launch_bounds__(kBlockThreads) void k8_emit(Lz4Args a) {
  const uint32_t w = blockIdx.x;
  if (w >= a.round_left[kChunkCount]) return;   // uniform
  const uint32_t bi = chunk_block(a, w);
  const uint32_t n_out = a.block_out_size[bi];
  const uint32_t c0 = (w - a.chunk_base[bi]) * kChunkBytes;
  const uint32_t c1 = c0 + kChunkBytes < n_out ? c0 + kChunkBytes : n_out;
  const uint32_t n = c1 - c0;
  const uint32_t lo = static_cast<uint32_t>(a.block_out_base[bi]) + c0;
  gptr<const uint32_t> link = GC<uint32_t>(a.link);
  gptr<uint8_t> out = GM<uint8_t>(a.out);
  bool bad = false;
  if ((lo & 3u) == 0) {
    for (uint32_t q = threadIdx.x; 4 * q < n; q += kBlockThreads) {
      if (4 * q + 4 <= n) {
        u32x4 v = *(gptr<const u32x4>)(link + lo + 4 * q);
        if (!(v.x >> 31)) v.x = link[v.x];
        if (!(v.y >> 31)) v.y = link[v.y];
        if (!(v.z >> 31)) v.z = link[v.z];
        if (!(v.w >> 31)) v.w = link[v.w];
        bad |= !((v.x & v.y & v.z & v.w) >> 31);
        *(gptr<uint32_t>)(out + lo + 4 * q) = (v.x & 0xFFu) | ((v.y & 0xFFu) << 8) | ((v.z & 0xFFu) << 16) | ((v.w & 0xFFu) << 24);
      } else {
        for (uint32_t k = 0; 4 * q + k < n; k++) {
          uint32_t v = link[lo + 4 * q + k];
          if (!(v >> 31)) v = link[v];
          bad |= !(v >> 31);
          out[lo + 4 * q + k] = static_cast<uint8_t>(v);
        }
      }
    }
  } else {
    for (uint32_t i = threadIdx.x; i < n; i += kBlockThreads) {
      uint32_t v = link[lo + i];
      if (!(v >> 31)) v = link[v];
      bad |= !(v >> 31);
      out[lo + i] = static_cast<uint8_t>(v);
    }
  }
  if (bad) lz4_fail(a.status);   // never silently: a chain the rounds did not finish is an internal error, not data
}

}  // namespace

// a.round_left, a.status and a.mark must be zero; a.link needs no preset (every word a later stage reads is written first).
hipError_t LaunchLz4Decompress(const Lz4Args& a, int num_cus, hipStream_t stream) {
  MI_DROP_STALE_ERROR();
  if (a.n_blocks == 0) return hipSuccess;
  if (a.zblocks) {
#ifdef MI_ZSTD_PROBE_BUILD
    static const bool probe = std::getenv("MI_ZSTD_PROBE") != nullptr;          // diagnostics build: a few blocks print where their time went
#else
    constexpr bool probe = false;
#endif
    hipLaunchKernelGGL(zstd_entropy, dim3(a.n_blocks), dim3(kZstdThreads), 0, stream, a, probe ? 0x100u : 0u);
    hipLaunchKernelGGL(zstd_layout, dim3(a.n_buffers), dim3(64), 0, stream, a);
  } else {
    // compressed blocks below 64 KiB (64 KiB is the default block size of every writer) are walked from an LDS copy
    static const bool force_global = std::getenv("MI_LZ4_PARSE_GLOBAL") != nullptr;   // tests: the variant for blocks too large for LDS
    static const bool no_dp = std::getenv("MI_LZ4_PARSE_SPECULATIVE") != nullptr;       // tests / A-B: the speculative walk for every block
    // blocks of <= kDpMaxComp compressed bytes (all of them for columnar data in 64 KiB blocks): the table walk, whose cost does
    // not depend on the bytes; larger ones: the speculative walk (the flag in _pad tells it which blocks are not its own)
    Lz4Args b = a;
    b._pad = 0;
    const uint32_t dp_comp = a.max_block_comp < kDpMaxComp ? a.max_block_comp : kDpMaxComp;
    if (!force_global && !no_dp && a.min_block_comp <= kDpMaxComp) {
      const uint32_t comp_pad = (dp_comp + 15u) & ~15u;
      const uint32_t lds = comp_pad + 16u + 2u * comp_pad + 32u;
      static bool attr_set = false;   // once per process: dynamic LDS beyond 64 KiB has to be opted in to
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_parse_dp), hipFuncAttributeMaxDynamicSharedMemorySize, 158 << 10);
        attr_set = true;
      }
      hipLaunchKernelGGL(lz4_parse_dp, dim3(a.n_blocks), dim3(kParseLanes), lds, stream, a);
      b._pad = 1;
    }
    if (b._pad == 0 || a.max_block_comp > kDpMaxComp) {
      if (!force_global && a.max_block_comp + 32u + 6144u <= (64u << 10))   // 64 KiB of LDS per workgroup without opting in to more (the kernel's own tables: 5 KiB)
        hipLaunchKernelGGL(lz4_parse<true>, dim3(a.n_blocks), dim3(kParseLanes), ((a.max_block_comp + 15u) & ~15u) + 16u, stream, b);
      else
        hipLaunchKernelGGL(lz4_parse<false>, dim3(a.n_blocks), dim3(kParseLanes), 0, stream, b);
    }
    hipLaunchKernelGGL(lz4_layout, dim3((a.n_buffers + 63) / 64), dim3(64), 0, stream, a);
  }
  // chunks of <= 8 KiB of one block: their number is only known on the device (k8_chunk_map), the grids are an upper bound
  hipLaunchKernelGGL(k8_chunk_map, dim3(1), dim3(kBlockThreads), 0, stream, a);
  const uint32_t max_chunks = static_cast<uint32_t>(a.out_size / kChunkBytes) + a.n_blocks + 1;
  hipLaunchKernelGGL(k8_expand_local, dim3(max_chunks), dim3(kBlockThreads), 0, stream, a);
  // chains only run backwards inside one buffer, and after the local pass every hop that is left crosses a chunk boundary:
  // depth <= chunks the longest buffer touches, rounds <= log5(depth) + 1 (5 for a 3 MB buffer)
  const uint64_t depth = a.max_buffer_len / kChunkBytes + 2 + a.max_buffer_blocks;   // every block may end in a short chunk
  int rounds = 2;   // a round divides the depth by kSkelHops + 1
  for (uint64_t reach = kSkelHops + 1; rounds < 33 && reach < depth; reach *= kSkelHops + 1) rounds++;
  const uint64_t nlocal = ((a.out_size + 3) / 4 + kLocalTileQuads - 1) / kLocalTileQuads;
  hipLaunchKernelGGL(lz4_collect, dim3(static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(nlocal, static_cast<uint64_t>(num_cus) * 16)))),
                     dim3(kBlockThreads), 0, stream, a);
  for (int r = 0; r < rounds; r++) hipLaunchKernelGGL(lz4_resolve_skeleton, dim3(static_cast<uint32_t>(num_cus) * 4), dim3(kBlockThreads), 0, stream, a, r);
  hipLaunchKernelGGL(k8_emit, dim3(max_chunks), dim3(kBlockThreads), 0, stream, a);
  return hipGetLastError();
}

}  // namespace device
}  // namespace miarrow
