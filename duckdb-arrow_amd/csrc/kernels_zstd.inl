// kernels_zstd.inl -- K8 for ZSTD frames: the two stages in front of the copy machinery of kernels_lz4.hip (included there).
//
// A ZSTD block is entropy-coded (zstd_format.hpp): its literals are Huffman streams, its sequences one FSE bitstream -- both
// serial by construction, a symbol's length is known only once it is decoded.  What is parallel is the BLOCK: the host walk
// (WalkZstdFrame) finds every block of every buffer from the headers and names, for tables a block inherits, the earlier block
// whose bytes describe them, so no block waits for another:
//   zstd_entropy   one workgroup of two waves per block.  Wave 0 builds the Huffman table and decodes the 4 literal streams on
//                  4 lanes into the literal scratch (behind the compressed body, same allocation); wave 1 builds the three FSE
//                  tables on 3 lanes and decodes the sequences on one: the descriptors {output position, literal source,
//                  literal length, match length} + offset that lz4_expand reads, in 256 equal slices per block.
//                  Repeat offsets -- the one state that runs from block to block -- are kept symbolic (zstd_format.hpp, RepStep):
//                  a slice records its effect on the history as a function of the history it started from.
//   zstd_layout    one wave per buffer, its blocks in order: first output byte of every block, the size check of the reference
//                  (base_stream_reader.cpp:24-29), and a prefix scan over the 256 slice functions of each block that gives every
//                  slice the history it really starts from; lz4_expand resolves a symbolic offset with it in one step.
// From there on a ZSTD batch is an LZ4 batch: lz4_expand writes the link words, the resolve kernels follow them, lz4_emit
// writes the bytes.  The serial chains (one table lookup in LDS per symbol) are latency-bound and leave the chip almost idle:
// the scan runs the batches of several slots side by side.

constexpr int kZstdThreads = 128;
constexpr int kZstdWindowWords = 128;            // per literal stream: 512 bytes of the stream in LDS at a time
constexpr int kZstdSeqWindowWords = 256;         // the sequences' bitstream: 1 KiB at a time
template <typename T>
using ldsptr = T __attribute__((address_space(3)))*;

// pos != 0 (the default): the streams are decoded with the POSITIONAL decoders of zstd_format.hpp -- a field of the bitstream
// is an indexed read from a 0.5 - 1 KiB window of the stream in LDS that the decoding lane slides itself, not a turn of a
// shifting bit buffer with its counters and refill tests -- and wave 1 shares the work on the sequences (see there).  pos == 0:
// the BackBits readers of the first formulation, everything about a sequence on one lane (MI_ZSTD_WINDOWED: tests, A/B).
// (Staging the whole block in LDS instead of windows was measured and dropped: 36 - 100 KiB per workgroup leave room for two
// blocks per CU instead of eight, and the kernel is a set of serial chains -- what it needs is many blocks side by side.)
__global__ __launch_bounds__(kZstdThreads) void zstd_entropy(Lz4Args a, uint32_t pos) {
  __shared__ uint16_t s_huf[1u << zstd::kHufMaxBits];
  __shared__ zstd::FseCell s_ll[512], s_of[256], s_ml[512], s_wcells[64];   // 8 bytes a cell
  __shared__ uint8_t s_weights[256];
  __shared__ int16_t s_counts[5][64];
  __shared__ uint16_t s_next[4][64];
  __shared__ uint32_t s_al[3], s_huf_bits, s_desc, s_fail, s_bits_at;
  __shared__ uint32_t s_win[4][kZstdWindowWords];
  __shared__ uint32_t s_seqwin[kZstdSeqWindowWords];
  const uint32_t bi = blockIdx.x;
  const zstd::BlockInfo* zb = static_cast<const zstd::BlockInfo*>(a.zblocks);
  const zstd::BlockInfo z = zb[bi];
  const Lz4BlockDev b = a.blocks[bi];
  // every pointer the serial loops use names its address space: global for the body and the scratch, LDS for the tables (see
  // zstd::Mem -- one FLAT access in such a loop costs an HBM round trip per symbol)
  gptr<const uint8_t> comp = GC<uint8_t>(a.comp);
  gptr<uint8_t> arena = GM<uint8_t>(a.literals);
  gptr<const uint8_t> c = comp + z.comp_off;
  ldsptr<uint16_t> huf = (ldsptr<uint16_t>)s_huf;
  ldsptr<zstd::FseCell> t_ll = (ldsptr<zstd::FseCell>)s_ll, t_of = (ldsptr<zstd::FseCell>)s_of, t_ml = (ldsptr<zstd::FseCell>)s_ml;
  const uint32_t per = b.seq_cap / kParseLanes;   // descriptors per slice
  gptr<u32x4> seq = GM<u32x4>(a.seq) + b.seq_base;
  gptr<uint32_t> seq_off = GM<uint32_t>(a.seq_off) + b.seq_base;
  gptr<uint32_t> lane_out = GM<uint32_t>(a.lane_out) + static_cast<size_t>(bi) * kParseLanes;
  gptr<uint32_t> lane_nseq = GM<uint32_t>(a.lane_nseq) + static_cast<size_t>(bi) * kParseLanes;
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
  if (z.type == 0) {   // raw: lz4_expand copies it (Lz4BlockDev::stored)
    if (tid == 0) {
      a.block_out_size[bi] = z.comp_size;
      a.block_nseq[bi] = 0;
    }
    return;
  }
  if (z.type == 1) {   // one byte, `regen` times: a literal and a match that overlaps it
    if (tid == 0) {
      arena[z.lit_pos] = c[0];
      u32x4 d;
      d.x = 0; d.y = z.lit_pos; d.z = 1; d.w = z.regen - 1;
      seq[0] = d;
      seq_off[0] = 1;
      a.block_out_size[bi] = z.regen;
      a.block_nseq[bi] = 1;
    }
    for (uint32_t k = tid; k < kParseLanes; k += kZstdThreads) {
      lane_out[k] = k == 0 ? 0 : z.regen;
      lane_nseq[k] = k == 0 ? 1 : 0;
    }
    return;
  }
  if (tid == 0) s_fail = 0;
  const bool staged = pos != 0;   // uniform (the name is history: the positional path)
  __syncthreads();
  // --- tables ---------------------------------------------------------------------------------------------------------
  if (wave == 0) {
    if (lane == 0 && z.lit_type >= 2) {
      const zstd::BlockInfo hs = zb[z.huf_src];
      uint32_t bits = 0;
      const uint32_t desc = zstd::ReadHuffmanTable(comp + hs.comp_off + hs.lit_hdr, hs.lit_comp, huf, &bits, (ldsptr<uint8_t>)s_weights,
                                                   (ldsptr<zstd::FseCell>)s_wcells, (ldsptr<int16_t>)s_counts[3], (ldsptr<uint16_t>)s_next[3]);
      if (!desc) s_fail = 1;
      s_desc = z.lit_type == 2 ? desc : 0;
      s_huf_bits = bits;
    }
  } else if (lane < 3 && z.nseq) {
    const int t = static_cast<int>(lane);
    const zstd::BlockInfo sb = zb[t == 0 ? z.ll_src : t == 1 ? z.of_src : z.ml_src];
    const uint32_t so = sb.seq_pos + sb.seq_hdr;
    ldsptr<zstd::FseCell> tab = t == 0 ? t_ll : t == 1 ? t_of : t_ml;
    const uint32_t al = so < sb.comp_size ? zstd::BuildSequenceTable(comp + sb.comp_off + so, sb.comp_size - so, t, tab, (ldsptr<int16_t>)s_counts[t],
                                                                      (ldsptr<uint16_t>)s_next[t])
                                          : ~0u;
    if (al == ~0u) s_fail = 1;
    s_al[t] = al;
  } else if (lane == 3 && z.nseq) {   // where the block's own bitstream begins: behind its table descriptions
    const uint32_t so = z.seq_pos + z.seq_hdr;
    const uint32_t bo = zstd::SequenceBitstreamOffset(c + so, z.comp_size - so, (ldsptr<int16_t>)s_counts[4]);
    if (bo == 0 || so + bo >= z.comp_size) s_fail = 1;
    s_bits_at = so + bo;
  }
  __syncthreads();
  const bool failed = s_fail != 0;   // uniform
  const uint32_t bits_at = z.nseq && !failed ? s_bits_at : 0, bits_len = z.nseq && !failed ? z.comp_size - bits_at : 0;
  // --- streams --------------------------------------------------------------------------------------------------------
  if (wave == 0) {
    if (failed) return;
    if (z.lit_type == 1) {
      const uint8_t v = c[z.lit_hdr];
      for (uint32_t i = lane; i < z.lit_regen; i += 64) arena[z.lit_pos + i] = v;
    } else if (z.lit_type >= 2 && lane < z.lit_streams) {
      uint32_t first, nbytes, out0, nsym;
      bool ok = zstd::LiteralStream(z, c, s_desc, lane, &first, &nbytes, &out0, &nsym);
      ok = ok && first + nbytes <= z.comp_size;
      if (staged) {
        const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(c + first) & 3u);
        zstd::SlidingWords<gptr<const uint8_t>, ldsptr<uint32_t>, kZstdWindowWords> sw;
        sw.Init(c + first - mis, (ldsptr<uint32_t>)s_win[lane]);
        ok = ok && zstd::DecodeHuffmanStreamPos(sw, mis, nbytes, nsym, huf, s_huf_bits, arena + z.lit_pos + out0);
      } else {
        // the stream is read through a window in LDS that the lane refills itself: between refills the loop touches HBM only
        // to store (a load would wait for the last store -- one counter for both -- at every refill of the bit buffer)
        zstd::BackBits<gptr<const uint8_t>, zstd::WindowWords<gptr<const uint8_t>, ldsptr<uint32_t>, kZstdWindowWords>> br;
        br.src.win = (ldsptr<uint32_t>)s_win[lane];
        ok = ok && zstd::DecodeHuffmanStream(br, c + first, nbytes, nsym, huf, s_huf_bits, arena + z.lit_pos + out0);
      }
      if (!ok) lz4_fail(a.status);   // the block's size is still reported by wave 1; the batch is rejected through the status word
    }
    return;
  }
  if (staged) {
    // ---- wave 1, positional path: lane 0 runs the FSE state machine alone, 64 sequences at a time, and leaves {literal length,
    // match length, offset code} in LDS; everything else about those 64 sequences is done by the whole wave: positions by a
    // wave scan, the repeat-offset history by a segmented scan of the sequences' functions (zstd_format.hpp RepFunction; a
    // segment = one of the block's 256 descriptor slices), descriptors and offsets as coalesced stores.  On one lane that
    // bookkeeping was four fifths of the 1 us a sequence cost.
    __shared__ uint32_t s_trip[64][3];
    gptr<u32x4> rep_fn = GM<u32x4>(a.rep_state) + static_cast<size_t>(bi) * kParseLanes;
    const uint32_t I0 = zstd::RepSlot(0), I1 = zstd::RepSlot(1), I2 = zstd::RepSlot(2);
    using SeqWords = zstd::SlidingWords<gptr<const uint8_t>, ldsptr<uint32_t>, kZstdSeqWindowWords>;
    zstd::SeqPosDecoder<SeqWords, ldsptr<zstd::FseCell>> dec;
    bool ok0 = !failed;
    if (lane == 0 && ok0 && z.nseq) {
      const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(c + bits_at) & 3u);
      SeqWords sw;
      sw.Init(c + bits_at - mis, (ldsptr<uint32_t>)s_seqwin);
      ok0 = dec.Open(sw, mis, bits_len, t_ll, s_al[0], t_of, s_al[1], t_ml, s_al[2]);
    }
    bool ok = __shfl(ok0 ? 1 : 0, 0, 64) != 0;
    uint32_t out_base = 0, lit_base = 0, n_desc = 0;          // uniform: totals of the groups before this one
    uint32_t cx = I0, cy = I1, cz = I2;                        // the function of the slice that straddles the group's start, up to there
    uint32_t slice_base = 0;                                   // ... and the output position that slice began at
    auto after = [](uint32_t gx, uint32_t gy, uint32_t gz, uint32_t x, uint32_t y, uint32_t zz, uint32_t* rx, uint32_t* ry, uint32_t* rz) {
      *rx = zstd::RepResolve(gx, x, y, zz);
      *ry = zstd::RepResolve(gy, x, y, zz);
      *rz = zstd::RepResolve(gz, x, y, zz);
    };
    auto group = [&](uint32_t cnt) {   // the descriptors n_desc .. n_desc + cnt - 1, their triples in s_trip
      const bool active = lane < cnt;
      const uint32_t ll = active ? s_trip[lane][0] : 0u, ml = active ? s_trip[lane][1] : 0u, code = active ? s_trip[lane][2] : 0u;
      const uint32_t incl_ll = wave_inclusive_scan_u32(ll), incl_out = wave_inclusive_scan_u32(ll + ml);
      const uint32_t lit_i = lit_base + incl_ll - ll, out_i = out_base + incl_out - (ll + ml);
      bool bad = active && (lit_i > z.lit_regen || ll > z.lit_regen - lit_i || out_i > zstd::kBlockMax || ll + ml > zstd::kBlockMax - out_i);
      const uint32_t n = n_desc + lane, k = n / per, j = n - k * per;
      const bool head = active && j == 0;
      const uint64_t heads = __ballot(head ? 1 : 0);
      const uint64_t below = heads & ((lane == 63u) ? ~0ull : ((2ull << lane) - 1ull));   // slice starts at or before this lane
      const bool carried = below == 0;                          // the lane's slice began in an earlier group
      const int hl = carried ? 0 : 63 - __builtin_clzll(below);
      const uint32_t head_out = __shfl(out_i, hl, 64);
      const uint32_t base_i = carried ? slice_base : head_out;
      // the history: inclusive segmented scan of the sequences' functions, then the straddling slice's part on top
      uint32_t px, py, pz, used;
      zstd::RepFunction(code, active && ml != 0, &px, &py, &pz, &used);
      uint32_t h = head ? 1u : 0u;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t nx = __shfl_up(px, d, 64), ny = __shfl_up(py, d, 64), nz = __shfl_up(pz, d, 64), nh = __shfl_up(h, d, 64);
        if (lane >= static_cast<uint32_t>(d) && !h) {
          after(px, py, pz, nx, ny, nz, &px, &py, &pz);
          h |= nh;
        }
      }
      if (carried) after(px, py, pz, cx, cy, cz, &px, &py, &pz);
      // the state BEFORE the lane's sequence, in terms of its slice's start
      uint32_t ex = __shfl_up(px, 1, 64), ey = __shfl_up(py, 1, 64), ez = __shfl_up(pz, 1, 64);
      if (lane == 0) { ex = cx; ey = cy; ez = cz; }
      if (head) { ex = I0; ey = I1; ez = I2; }
      uint32_t off = 0;
      if (active && ml != 0) {
        off = zstd::RepResolve(used, ex, ey, ez);
        if (off == 0) bad = true;
      }
      if (active) {
        u32x4 d4;
        d4.x = out_i - base_i;
        d4.y = z.lit_pos + lit_i;
        d4.z = ll;
        d4.w = ml;
        seq[k * per + j] = d4;
        seq_off[k * per + j] = off;
        if (head) lane_out[k] = out_i;
        if (j + 1 == per) {   // the slice is full: its function is final
          lane_nseq[k] = per;
          u32x4 f;
          f.x = px; f.y = py; f.z = pz; f.w = 0;
          rep_fn[k] = f;
        }
      }
      // what the next group starts from
      const int last = static_cast<int>(cnt) - 1;
      const bool last_full = __shfl((active && j + 1 == per) ? 1 : 0, last, 64) != 0;
      const uint32_t lx = __shfl(px, last, 64), ly = __shfl(py, last, 64), lz = __shfl(pz, last, 64), lb = __shfl(base_i, last, 64);
      cx = last_full ? I0 : lx;
      cy = last_full ? I1 : ly;
      cz = last_full ? I2 : lz;
      slice_base = lb;
      out_base += __shfl(incl_out, 63, 64);
      lit_base += __shfl(incl_ll, 63, 64);
      n_desc += cnt;
      if (__any(bad ? 1 : 0)) ok = false;
    };
    for (uint32_t g0 = 0; g0 < z.nseq && ok; g0 += 64) {
      const uint32_t cnt = z.nseq - g0 < 64u ? z.nseq - g0 : 64u;
      if (lane == 0) {
        for (uint32_t i = 0; i < cnt && ok0; i++) {
          uint32_t ll, ml, code;
          ok0 = dec.Step(g0 + i + 1 < z.nseq, &ll, &ml, &code);
          s_trip[i][0] = ll;
          s_trip[i][1] = ml;
          s_trip[i][2] = code;
        }
        if (ok0 && g0 + cnt == z.nseq) ok0 = dec.AtEnd();
      }
      __builtin_amdgcn_wave_barrier();
      ok = __shfl(ok0 ? 1 : 0, 0, 64) != 0;
      if (!ok) break;
      group(cnt);
      __builtin_amdgcn_wave_barrier();
    }
    if (ok && lit_base < z.lit_regen) {   // the literals behind the last sequence: a descriptor without a match
      if (lane == 0) {
        s_trip[0][0] = z.lit_regen - lit_base;
        s_trip[0][1] = 0;
        s_trip[0][2] = 0;
      }
      __builtin_amdgcn_wave_barrier();
      group(1);
    }
    if (!ok) {
      if (lane == 0) lz4_fail(a.status);
      out_base = 0;
      n_desc = 0;
    }
    uint32_t k_open = n_desc / per;
    if (n_desc % per) {   // the last slice is partly filled
      if (lane == 0) {
        lane_nseq[k_open] = n_desc % per;
        u32x4 f;
        f.x = cx; f.y = cy; f.z = cz; f.w = 0;
        rep_fn[k_open] = f;
      }
      k_open++;
    }
    for (uint32_t k = k_open + lane; k < kParseLanes; k += 64) {   // empty slices begin where the block ends and leave the history as it is
      lane_out[k] = out_base;
      lane_nseq[k] = 0;
      u32x4 f;
      f.x = I0; f.y = I1; f.z = I2; f.w = 0;
      rep_fn[k] = f;
    }
    if (lane == 0) {
      a.block_out_size[bi] = out_base;
      a.block_nseq[bi] = n_desc;
    }
    return;
  }
  if (lane != 0) {
    return;
  }
  // wave 1, lane 0: the sequences (blocks too large to stage: the windowed reader, everything on this lane)
  uint32_t out_pos = 0, lit_used = 0, k = 0, j = 0, lane_base = 0, n_desc = 0;
  uint32_t S[3] = {zstd::RepSlot(0), zstd::RepSlot(1), zstd::RepSlot(2)};   // the history, as a function of the slice's start
  gptr<u32x4> rep_fn = GM<u32x4>(a.rep_state) + static_cast<size_t>(bi) * kParseLanes;
  auto close_slice = [&]() {
    u32x4 f;
    f.x = S[0]; f.y = S[1]; f.z = S[2]; f.w = 0;
    rep_fn[k] = f;
    S[0] = zstd::RepSlot(0); S[1] = zstd::RepSlot(1); S[2] = zstd::RepSlot(2);
  };
  bool ok = !failed;
  auto put = [&](uint32_t ll, uint32_t ml, uint32_t code) {
    if (j == 0) {
      lane_base = out_pos;
      lane_out[k] = out_pos;
    }
    uint32_t off = 0;
    if (ml) {
      off = zstd::RepStep(code, S);
      if (off == 0) ok = false;
    }
    u32x4 d;
    d.x = out_pos - lane_base;
    d.y = z.lit_pos + lit_used;
    d.z = ll;
    d.w = ml;
    seq[k * per + j] = d;
    seq_off[k * per + j] = off;
    lit_used += ll;
    out_pos += ll + ml;
    n_desc++;
    if (++j == per) {
      lane_nseq[k] = per;
      close_slice();
      k++;
      j = 0;
    }
  };
  if (ok && z.nseq) {
    auto emit = [&](uint32_t, uint32_t ll, uint32_t ml, uint32_t code) {
      if (ll > z.lit_regen - lit_used || ll + ml > zstd::kBlockMax - out_pos) return false;
      put(ll, ml, code);
      return ok;
    };
    // like the literal streams: read through a window in LDS, so that the loop's only traffic to HBM is its stores
    zstd::BackBits<gptr<const uint8_t>, zstd::WindowWords<gptr<const uint8_t>, ldsptr<uint32_t>, kZstdSeqWindowWords>> br;
    br.src.win = (ldsptr<uint32_t>)s_seqwin;
    ok = zstd::DecodeSequences(br, c + bits_at, bits_len, z.nseq, t_ll, s_al[0], t_of, s_al[1], t_ml, s_al[2], emit);
  }
  if (ok && lit_used < z.lit_regen) {
    ok = z.lit_regen - lit_used <= zstd::kBlockMax - out_pos;
    if (ok) put(z.lit_regen - lit_used, 0, 0);   // offset 0: not a match, zstd_layout's repeat offsets pass it by
  }
  if (!ok) {
    lz4_fail(a.status);
    out_pos = 0;
    n_desc = 0;
    k = 0;
    j = 0;
  }
  if (j) {
    lane_nseq[k] = j;
    close_slice();
    k++;
  }
  S[0] = zstd::RepSlot(0); S[1] = zstd::RepSlot(1); S[2] = zstd::RepSlot(2);
  for (; k < kParseLanes; k++) {   // empty slices begin where the block ends and leave the history as it is
    lane_out[k] = out_pos;
    lane_nseq[k] = 0;
    close_slice();
  }
  a.block_out_size[bi] = out_pos;
  a.block_nseq[bi] = n_desc;
}

__global__ __launch_bounds__(64) void zstd_layout(Lz4Args a) {
  const uint32_t u = blockIdx.x, lane = threadIdx.x;   // one wave per buffer
  const Lz4BufferDev f = a.buffers[u];
  const zstd::BlockInfo* zb = static_cast<const zstd::BlockInfo*>(a.zblocks);
  uint64_t at = f.out_off;
  uint32_t r0 = 1, r1 = 4, r2 = 8;   // the frame's history at its start (uniform across the wave)
  for (uint32_t kb = 0; kb < f.n_blocks; kb++) {
    const uint32_t bi = f.first_block + kb;
    if (lane == 0) a.block_out_base[bi] = at;
    at += a.block_out_size[bi];
    if (zb[bi].type != 2) continue;   // raw and RLE blocks leave the history alone
    // 4 slices per lane: their functions, the lane's total, an inclusive scan of the totals over the wave
    gptr<u32x4> fn = GM<u32x4>(a.rep_state) + static_cast<size_t>(bi) * kParseLanes + 4 * lane;
    const u32x4 f0 = fn[0], f1 = fn[1], f2 = fn[2], f3 = fn[3];
    auto after = [](const u32x4& g, uint32_t x, uint32_t y, uint32_t z) {   // g applied to the state (x, y, z)
      u32x4 r;
      r.x = zstd::RepResolve(g.x, x, y, z);
      r.y = zstd::RepResolve(g.y, x, y, z);
      r.z = zstd::RepResolve(g.z, x, y, z);
      r.w = 0;
      return r;
    };
    u32x4 t = after(f1, f0.x, f0.y, f0.z);
    t = after(f2, t.x, t.y, t.z);
    t = after(f3, t.x, t.y, t.z);
    u32x4 inc = t;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      const uint32_t px = __shfl_up(inc.x, d), py = __shfl_up(inc.y, d), pz = __shfl_up(inc.z, d);
      if (lane >= d) inc = after(inc, px, py, pz);
    }
    uint32_t ex = __shfl_up(inc.x, 1), ey = __shfl_up(inc.y, 1), ez = __shfl_up(inc.z, 1);
    if (lane == 0) { ex = zstd::RepSlot(0); ey = zstd::RepSlot(1); ez = zstd::RepSlot(2); }
    // the history each of the lane's slices starts from: concrete from here on
    u32x4 s0;
    s0.x = zstd::RepResolve(ex, r0, r1, r2);
    s0.y = zstd::RepResolve(ey, r0, r1, r2);
    s0.z = zstd::RepResolve(ez, r0, r1, r2);
    s0.w = 0;
    const u32x4 s1 = after(f0, s0.x, s0.y, s0.z), s2 = after(f1, s1.x, s1.y, s1.z), s3 = after(f2, s2.x, s2.y, s2.z);
    fn[0] = s0;
    fn[1] = s1;
    fn[2] = s2;
    fn[3] = s3;
    const u32x4 end = after(inc, r0, r1, r2);   // lane 63: the whole block
    r0 = __shfl(end.x, 63);
    r1 = __shfl(end.y, 63);
    r2 = __shfl(end.z, 63);
  }
  if (lane == 0) {
    const bool ok = at - f.out_off == f.out_len;
    a.buffer_ok[u] = ok ? 1u : 0u;
    if (!ok) lz4_fail(a.status);
  }
}
