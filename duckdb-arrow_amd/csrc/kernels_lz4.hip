// kernels_lz4.hip -- K8: LZ4_FRAME buffer decompression in HBM.
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
//   1. lz4_parse    one LANE per block walks the tokens only (no data is copied): one descriptor per sequence
//                   {output position in the block, literal source, literal length, match length, match offset} and the
//                   block's decompressed size.  Every block of every buffer of the record batch at once.
//   2. lz4_layout   one lane per buffer: first output byte of each of its blocks (running sum), and the check the
//                   reference makes after decompressing: the sizes must add up to the declared uncompressed length.
//   3. lz4_expand   one workgroup per block, one thread per sequence: literals are copied to their final place; every
//                   byte of a match gets a LINK = the position it copies from (always an earlier byte of the buffer).
//   4. lz4_resolve  pointer jumping over the links, all bytes in parallel: link[j] <- link[link[j]] until the chain ends
//                   in a byte that is known, then the byte is fetched.  A chain of depth d needs ceil(log2 d) + 1 rounds
//                   (an overlapping run "offset 1, length 60000" is 16 rounds); a round that finds nothing left to do
//                   tells the next ones (launched blindly, no host round trip) to return at once.
// HBM traffic per decompressed byte: 4 B memset + ~5 B expand + 8 B per resolve round; a record batch of lineitem
// (21.5 MB) is ~1 GB of traffic for ~6 rounds.  The serial part is step 1: ~4000 sequences per 64 KiB block, two or three
// dependent loads each.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_common.hpp"
#include "kernels.hpp"

namespace miarrow {
namespace device {
namespace {

constexpr uint32_t kLinkDone = 0xFFFFFFFFu;

__device__ __forceinline__ void lz4_fail(uint32_t* status) { atomicOr(status, MI_ST_DECOMPRESS); }

__global__ __launch_bounds__(64) void lz4_parse(Lz4Args a) {
  const uint32_t bi = blockIdx.x * 64 + threadIdx.x;
  if (bi >= a.n_blocks) return;
  const Lz4BlockDev b = a.blocks[bi];
  const uint32_t block_max = a.buffers[b.buffer].block_max;
  if (b.stored) {  // the block holds its bytes as they are
    a.block_out_size[bi] = b.comp_size <= block_max ? b.comp_size : 0u;
    a.block_nseq[bi] = 0;
    if (b.comp_size > block_max) lz4_fail(a.status);
    return;
  }
  gptr<const uint8_t> in = GC<uint8_t>(a.comp);
  gptr<u32x4> seq = GM<u32x4>(a.seq) + b.seq_base;
  gptr<uint32_t> seq_off = GM<uint32_t>(a.seq_off) + b.seq_base;
  uint32_t ip = b.comp_off;
  const uint32_t end = b.comp_off + b.comp_size;
  uint32_t op = 0, n = 0;
  bool ok = true;
  while (ip < end) {
    const uint32_t token = in[ip++];
    uint32_t ll = token >> 4;
    if (ll == 15) {
      uint32_t x;
      do {
        if (ip >= end) { ok = false; break; }
        x = in[ip++];
        ll += x;
      } while (x == 255 && ll < (1u << 24));
      if (!ok || ll >= (1u << 24)) { ok = false; break; }
    }
    const uint32_t lit_src = ip;
    if (ll > end - ip) { ok = false; break; }
    ip += ll;
    uint32_t ml = 0, offset = 0;
    if (ip < end) {  // the last sequence of a block is literals only
      if (end - ip < 2) { ok = false; break; }
      offset = static_cast<uint32_t>(in[ip]) | (static_cast<uint32_t>(in[ip + 1]) << 8);
      ip += 2;
      ml = token & 15u;
      if (ml == 15) {
        uint32_t x;
        do {
          if (ip >= end) { ok = false; break; }
          x = in[ip++];
          ml += x;
        } while (x == 255 && ml < (1u << 24));
        if (!ok || ml >= (1u << 24)) { ok = false; break; }
      }
      ml += 4;
      if (offset == 0) { ok = false; break; }
    }
    if (n >= b.seq_cap || ll + ml > block_max - op) { ok = false; break; }
    u32x4 d;
    d.x = op;
    d.y = lit_src;
    d.z = ll;
    d.w = ml;
    seq[n] = d;
    seq_off[n] = offset;
    n++;
    op += ll + ml;
  }
  if (!ok) {
    lz4_fail(a.status);
    op = 0;
    n = 0;
  }
  a.block_out_size[bi] = op;
  a.block_nseq[bi] = n;
}

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

__global__ __launch_bounds__(kBlockThreads) void lz4_expand(Lz4Args a) {
  const uint32_t bi = blockIdx.x;
  const Lz4BlockDev b = a.blocks[bi];
  if (!a.buffer_ok[b.buffer]) return;  // uniform
  const uint64_t base = a.block_out_base[bi];      // offset in the decompressed body
  const uint64_t buffer_lo = a.buffers[b.buffer].out_off;
  gptr<const uint8_t> in = GC<uint8_t>(a.comp);
  gptr<uint8_t> out = GM<uint8_t>(a.out);
  gptr<uint32_t> link = GM<uint32_t>(a.link[0]);
  if (b.stored) {
    for (uint32_t i = threadIdx.x; i < b.comp_size; i += kBlockThreads) out[base + i] = in[b.comp_off + i];
    return;  // links of the whole body start as "done"
  }
  const uint32_t nseq = a.block_nseq[bi];
  gptr<const u32x4> seq = GC<u32x4>(a.seq) + b.seq_base;
  gptr<const uint32_t> seq_off = GC<uint32_t>(a.seq_off) + b.seq_base;
  for (uint32_t s = threadIdx.x; s < nseq; s += kBlockThreads) {
    const u32x4 d = seq[s];
    const uint64_t lit_at = base + d.x;
    for (uint32_t i = 0; i < d.z; i++) out[lit_at + i] = in[d.y + i];
    if (d.w) {
      const uint64_t m_at = lit_at + d.z;
      const uint32_t offset = seq_off[s];
      if (offset > m_at - buffer_lo) {  // reaches in front of the buffer: not a frame an encoder writes
        lz4_fail(a.status);
        for (uint32_t i = 0; i < d.w; i++) out[m_at + i] = 0;
      } else {
        for (uint32_t i = 0; i < d.w; i++) link[m_at + i] = static_cast<uint32_t>(m_at + i - offset);
      }
    }
  }
}

// One round of pointer jumping over bytes [0, n): reads link[from], writes link[to].
__global__ __launch_bounds__(kBlockThreads) void lz4_resolve(Lz4Args a, int round) {
  if (round > 0 && a.round_left[round - 1] == 0) {
    // nothing was left after the previous round; pass the word on so that every later round sees it without a chain
    if (blockIdx.x == 0 && threadIdx.x == 0) a.round_left[round] = 0;
    return;
  }
  gptr<const uint32_t> from = GC<uint32_t>(a.link[round & 1]);
  gptr<uint32_t> to = GM<uint32_t>(a.link[(round & 1) ^ 1]);
  gptr<uint8_t> out = GM<uint8_t>(a.out);
  bool left = false;
  for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kBlockThreads + threadIdx.x; j < a.out_size; j += static_cast<uint64_t>(gridDim.x) * kBlockThreads) {
    const uint32_t s = from[j];
    uint32_t next = kLinkDone;
    if (s != kLinkDone) {
      const uint32_t t = from[s];
      if (t == kLinkDone) {
        out[j] = out[s];   // out[s] was final before this launch began
      } else {
        next = t;
        left = true;
      }
    }
    to[j] = next;
  }
  if (__syncthreads_or(left ? 1 : 0) && threadIdx.x == 0) atomicAdd(&a.round_left[round], 1u);
}

}  // namespace

// a.link[0] must hold 0xFF in every byte (all links "done"), a.round_left zeros, a.out zeros where padding is expected.
hipError_t LaunchLz4Decompress(const Lz4Args& a, int num_cus, hipStream_t stream) {
  MI_DROP_STALE_ERROR();
  if (a.n_blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(lz4_parse, dim3((a.n_blocks + 63) / 64), dim3(64), 0, stream, a);
  hipLaunchKernelGGL(lz4_layout, dim3((a.n_buffers + 63) / 64), dim3(64), 0, stream, a);
  hipLaunchKernelGGL(lz4_expand, dim3(a.n_blocks), dim3(kBlockThreads), 0, stream, a);
  // chains only run backwards inside one buffer: depth < its length, rounds <= log2(length) + 1
  int rounds = 2;
  while (rounds < 34 && (1ull << (rounds - 1)) < a.max_buffer_len) rounds++;
  const uint64_t want = (a.out_size + kBlockThreads - 1) / kBlockThreads;
  const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>(want, static_cast<uint64_t>(num_cus) * 32));
  for (int r = 0; r < rounds; r++) hipLaunchKernelGGL(lz4_resolve, dim3(grid ? grid : 1), dim3(kBlockThreads), 0, stream, a, r);
  return hipGetLastError();
}

}  // namespace device
}  // namespace miarrow
