// kernels_zstd.inl -- K8 for ZSTD frames: the two stages in front of the copy machinery of kernels_lz4.hip (included there).
//
// A ZSTD block is entropy-coded (zstd_format.hpp): its literals are Huffman streams, its sequences one FSE bitstream -- both
// serial by construction, a symbol's length is known only once it is decoded.  What is parallel is the BLOCK: the host walk
// (WalkZstdFrame) finds every block of every buffer from the headers and names, for tables a block inherits, the earlier block
// whose bytes describe them, so no block waits for another:
//   zstd_entropy   one workgroup of two waves per block.  Wave 0 builds the Huffman table and decodes the 4 literal streams on
//                  4 lanes into the literal scratch (behind the compressed body, same allocation); wave 1 builds the three FSE
//                  tables and decodes the sequences, the three FSE states on three lanes: the descriptors {output position, literal source,
//                  literal length, match length} + offset that k8_expand_local reads, in 256 equal slices per block.
//                  Repeat offsets -- the one state that runs from block to block -- are kept symbolic (zstd_format.hpp, RepStep):
//                  a slice records its effect on the history as a function of the history it started from.
//   zstd_layout    one wave per buffer, its blocks in order: first output byte of every block, the size check of the reference
//                  (base_stream_reader.cpp:24-29), and a prefix scan over the 256 slice functions of each block that gives every
//                  slice the history it really starts from; k8_expand_local resolves a symbolic offset with it in one step.
// From there on a ZSTD batch is an LZ4 batch: k8_expand_local writes the link words, the skeleton kernels follow them, k8_emit
// writes the bytes.  The serial chains (one table lookup in LDS per symbol) are latency-bound and leave the chip almost idle:
// the scan runs the batches of several slots side by side.

constexpr int kZstdThreads = 128;
constexpr int kZstdWindowWords = 128;            // per literal stream: 512 bytes of the stream in LDS at a time
constexpr int kZstdSeqWindowWords = 256;         // the sequences' bitstream: 1 KiB at a time
template <typename T>
using ldsptr = T __attribute__((address_space(3)))*;


// The streams are decoded POSITIONALLY: the head of a backward bitstream is a bit position, a field an indexed read from a
// 0.5 - 1 KiB window of the stream in LDS -- not a turn of a shifting bit buffer with its counters and refill tests (the first
// formulation: BackBits in zstd_format.hpp, still what the table descriptions are read with) -- and wave 1 shares the work on
// the sequences (see there).  `flags`: bit 8 = diagnostics (probe builds only).
// (Staging the whole block in LDS instead of windows was measured and dropped: 36 - 100 KiB per workgroup leave room for two
// blocks per CU instead of eight, and the kernel is a set of serial chains -- what it needs is many blocks side by side.)
__global__ __launch_bounds__(kZstdThreads) void zstd_entropy(Lz4Args a, uint32_t flags) {
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
  if (z.type == 0) {   // raw: k8_expand_local copies it (Lz4BlockDev::stored)
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
#ifdef MI_ZSTD_PROBE_BUILD
  const bool probe = (flags & 0x100u) != 0 && bi % 41u == 3u;   // MI_ZSTD_PROBE: this block prints its phases (10 ns ticks)
#else
  constexpr bool probe = false;   // diagnostics builds only: -DMI_ZSTD_PROBE_BUILD
#endif
  const uint64_t t_start = probe ? wall_clock64() : 0;
  __syncthreads();
  // --- tables ---------------------------------------------------------------------------------------------------------
  // The descriptions (a Huffman tree: <= 129 bytes; three FSE distributions: <= 153 bytes together) are first copied to LDS by
  // the whole wave -- the stream windows are not in use yet -- because parsing them is a chain of dependent byte reads, a few
  // per symbol: from HBM that chain was most of the 0.27 ms the tables took.  One lane parses (weights / normalized counts, the
  // FSE spread); the steps whose result does not depend on the order of execution -- the Huffman cells, an FSE cell's state
  // number = its rank among its symbol's cells -- are done by the whole wave.
  constexpr uint32_t kDescBytes = 256;   // staged per description; the rest of the staging area reads as zero
  if (wave == 0) {
    if (z.lit_type >= 2) {   // uniform
      const zstd::BlockInfo hs = zb[z.huf_src];
      ldsptr<uint8_t> desc_lds = (ldsptr<uint8_t>)s_win;
      gptr<const uint8_t> src = comp + hs.comp_off + hs.lit_hdr;
      const uint32_t avail = hs.lit_comp < kDescBytes ? hs.lit_comp : kDescBytes;
      for (uint32_t i = lane; i < kDescBytes + 8; i += 64) desc_lds[i] = i < avail ? src[i] : uint8_t(0);
      __builtin_amdgcn_wave_barrier();
      uint32_t n = 0, bits = 0, desc = 0;
      if (lane == 0)
        desc = zstd::ReadHuffmanWeights((ldsptr<const uint8_t>)desc_lds, avail, &n, &bits, (ldsptr<uint8_t>)s_weights, (ldsptr<zstd::FseCell>)s_wcells,
                                        (ldsptr<int16_t>)s_counts[3], (ldsptr<uint16_t>)s_next[3]);
      __builtin_amdgcn_wave_barrier();
      desc = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(desc)));
      n = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(n)));
      bits = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(bits)));
      bool ok = desc != 0;
      if (ok) {
        // cells in the order of ascending weight, symbols of one weight in symbol order (FillHuffmanTable): a symbol's first
        // cell from ballots, short runs written by the symbol's lane, long ones by the whole wave
        uint32_t wgt[4];
#pragma unroll
        for (int k = 0; k < 4; k++) wgt[k] = lane + 64u * k < n ? s_weights[lane + 64u * k] : 0u;
        const uint64_t below = (1ull << lane) - 1ull;
        uint32_t at = 0;
        for (uint32_t w = 1; w <= bits; w++) {
          const uint32_t len = 1u << (w - 1);
          const uint16_t tag = static_cast<uint16_t>((bits + 1 - w) << 8);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const bool has = wgt[k] == w;
            const uint64_t mask = __ballot(has ? 1 : 0);
            if (mask == 0) continue;   // uniform
            if (len < 64) {
              if (has) {
                const uint32_t start = at + static_cast<uint32_t>(__builtin_popcountll(mask & below)) * len;
                for (uint32_t i = 0; i < len; i++) huf[start + i] = static_cast<uint16_t>(tag | (lane + 64u * k));
              }
            } else {
              uint32_t start = at;
              for (uint64_t m = mask; m; m &= m - 1, start += len) {
                const uint16_t cell = static_cast<uint16_t>(tag | (static_cast<uint32_t>(__builtin_ctzll(m)) + 64u * k));
                for (uint32_t i = lane; i < len; i += 64) huf[start + i] = cell;
              }
            }
            at += static_cast<uint32_t>(__builtin_popcountll(mask)) * len;
          }
        }
        ok = at == (1u << bits);
      }
      if (lane == 0) {
        if (!ok) s_fail = 1;
        s_desc = z.lit_type == 2 ? desc : 0;
        s_huf_bits = bits;
      }
    }
  } else if (z.nseq) {   // uniform
    // lanes 0..2: the literal-length, offset and match-length tables, each from the block that describes it; lane 3: where
    // this block's own bitstream begins.  Four descriptions of 256 bytes in the sequences' window.
    ldsptr<uint8_t> desc_lds = (ldsptr<uint8_t>)s_seqwin;
    static_assert(kZstdSeqWindowWords * 4 == 4 * kDescBytes, "four staged descriptions");
    {
      const uint32_t r = lane >> 4;   // 16 lanes per description, 16 bytes each
      const zstd::BlockInfo sb = zb[r == 0 ? z.ll_src : r == 1 ? z.of_src : r == 2 ? z.ml_src : bi];
      const uint32_t so = sb.seq_pos + sb.seq_hdr;
      const uint32_t avail = so < sb.comp_size ? sb.comp_size - so : 0u;
      gptr<const uint8_t> src = comp + sb.comp_off + so;
      uint8_t v[16];
#pragma unroll
      for (uint32_t i = 0; i < 16; i++) {
        const uint32_t at = 16u * (lane & 15u) + i;
        v[i] = at < avail ? src[at] : uint8_t(0);
      }
#pragma unroll
      for (uint32_t i = 0; i < 16; i++) desc_lds[r * kDescBytes + 16u * (lane & 15u) + i] = v[i];
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t al = 0, nsym = 0;
    if (lane < 3) {
      const int t = static_cast<int>(lane);
      const zstd::BlockInfo sb = zb[t == 0 ? z.ll_src : t == 1 ? z.of_src : z.ml_src];
      const uint32_t so = sb.seq_pos + sb.seq_hdr;
      const uint32_t avail = so < sb.comp_size ? (sb.comp_size - so < kDescBytes - 4 ? sb.comp_size - so : kDescBytes - 4) : 0u;
      ldsptr<zstd::FseCell> tab = t == 0 ? t_ll : t == 1 ? t_of : t_ml;
      al = avail ? zstd::BuildSequenceTable<false>((ldsptr<const uint8_t>)desc_lds + lane * kDescBytes, avail, t, tab, (ldsptr<int16_t>)s_counts[t],
                                                   (ldsptr<uint16_t>)s_next[t], &nsym)
                 : ~0u;
      if (al == ~0u) s_fail = 1;
      s_al[t] = al;
    } else if (lane == 3) {
      const uint32_t so = z.seq_pos + z.seq_hdr;
      const uint32_t avail = z.comp_size - so < kDescBytes - 4 ? z.comp_size - so : kDescBytes - 4;
      const uint32_t bo = zstd::SequenceBitstreamOffset((ldsptr<const uint8_t>)desc_lds + 3 * kDescBytes, avail, (ldsptr<int16_t>)s_counts[4]);
      if (bo == 0 || so + bo >= z.comp_size) s_fail = 1;
      s_bits_at = so + bo;
    }
    __builtin_amdgcn_wave_barrier();
    // the cells: a cell's state number is its symbol's counter + its rank among that symbol's cells in table order.  Lane s
    // keeps the counter of symbol s (<= 53 symbols); 64 cells at a time, one ballot per distinct symbol among them.
#pragma unroll 1
    for (int t = 0; t < 3; t++) {
      const uint32_t t_al = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(al), t));
      const uint32_t t_nsym = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(nsym), t));
      if (t_al == ~0u || t_nsym == 0) continue;   // malformed, or an RLE table (one cell, final)
      ldsptr<zstd::FseCell> tab = t == 0 ? t_ll : t == 1 ? t_of : t_ml;
      uint32_t counter = lane < t_nsym ? s_next[t][lane] : 0u;
      const uint32_t size = 1u << t_al;
      const uint64_t below = (1ull << lane) - 1ull;
      for (uint32_t u0 = 0; u0 < size; u0 += 64) {
        const uint32_t u = u0 + lane;
        const bool valid = u < size;
        const uint32_t sym = valid ? static_cast<uint32_t>(tab[u]) & 63u : 64u;
        uint32_t ns = 1;
        for (uint64_t left = __ballot(valid ? 1 : 0); left;) {   // uniform
          const uint32_t ls = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(sym), __builtin_ctzll(left)));
          const uint64_t same = __ballot(sym == ls ? 1 : 0);
          const uint32_t base = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(counter), static_cast<int>(ls)));
          if (sym == ls) ns = base + static_cast<uint32_t>(__builtin_popcountll(same & below));
          if (lane == ls) counter = base + static_cast<uint32_t>(__builtin_popcountll(same));
          left &= ~same;
        }
        if (valid) tab[u] = zstd::FinishFseCell(sym, ns, t_al, t);
      }
    }
  }
  __syncthreads();
  const uint64_t t_tables = probe ? wall_clock64() : 0;
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
      {
        const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(c + first) & 3u);
        zstd::SlidingWords<gptr<const uint8_t>, ldsptr<uint32_t>, kZstdWindowWords> sw;
        sw.Init(c + first - mis, (ldsptr<uint32_t>)s_win[lane]);
        ok = ok && zstd::DecodeHuffmanStreamPos(sw, mis, nbytes, nsym, huf, s_huf_bits, arena + z.lit_pos + out0);
      }
      if (!ok) lz4_fail(a.status);   // the block's size is still reported by wave 1; the batch is rejected through the status word
      if (probe)
        printf("zprobe lit block %u stream %u: tables %llu literals %llu ticks, %u symbols, %u bytes\n", bi, lane,
               (unsigned long long)(t_tables - t_start), (unsigned long long)(wall_clock64() - t_tables), nsym, nbytes);
    }
    return;
  }
  {
    // ---- wave 1.  The three FSE states (offset, match length, literal length) sit on lanes 0, 1, 2: a lane
    // looks up ITS cell, the three exchange how many bits their fields take (v_readlane: the head position of the bitstream
    // and every field position are scalars), and each reads its two fields -- the extra bits of its value, the bits of its
    // next state -- from the window of the stream in LDS: one cell look-up and one pair of field reads per sequence as the
    // dependent chain, a third of the instructions one lane needed for all three states.  The window (1 KiB) is refilled
    // by the whole wave before a group of 64 sequences whenever fewer than 64 worst-case sequences (88 bits each) are left
    // in it, so the state machine itself never waits for HBM.  The three values of a sequence go to s_trip; everything else
    // about those 64 sequences is done by the whole wave: positions by a wave scan, the repeat-offset history by a segmented
    // scan of the sequences' functions (zstd_format.hpp RepFunction; a segment = one of the block's 256 descriptor slices),
    // descriptors and offsets as coalesced stores.
    __shared__ uint32_t s_trip[64][3];   // {literal length, match length, offset value} of the group's sequences
    gptr<u32x4> rep_fn = GM<u32x4>(a.rep_state) + static_cast<size_t>(bi) * kParseLanes;
    const uint32_t I0 = zstd::RepSlot(0), I1 = zstd::RepSlot(1), I2 = zstd::RepSlot(2);
    constexpr int32_t kW = kZstdSeqWindowWords;
    constexpr int32_t kGroupBits = 64 * 88;   // a sequence takes at most 30 + 16 + 16 extra bits and 9 + 9 + 8 state bits
    ldsptr<uint32_t> win = (ldsptr<uint32_t>)s_seqwin;
    const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(c + bits_at) & 3u);
    gptr<const uint8_t> wbase = c + bits_at - mis;   // word 0 of the stream: the 4-byte boundary at or before its first byte
    int32_t wlo = 0;                                  // the window holds words wlo .. wlo + kW - 1 (uniform)
    auto refill = [&](int32_t top_word) {             // whole wave: top_word becomes the window's last word
      wlo = top_word + 1 - kW;
      uint32_t v[kW / 64];
#pragma unroll
      for (int k = 0; k < kW / 64; k++) {
        const int32_t j = wlo + static_cast<int32_t>(lane) + 64 * k;
        v[k] = j >= 0 ? zstd::Mem<gptr<const uint8_t>>::Load32(wbase + 4 * j) : 0u;
      }
#pragma unroll
      for (int k = 0; k < kW / 64; k++) win[lane + 64 * k] = v[k];
      __builtin_amdgcn_wave_barrier();
    };
    auto field = [&](int32_t at, uint32_t n) -> uint32_t {   // bits [at, at + n) of the stream, n < 32, inside the window
      const int32_t i = (at >> 5) - wlo;
      return __builtin_amdgcn_alignbit(win[i + 1], win[i], static_cast<uint32_t>(at) & 31u) & ((1u << n) - 1u);
    };
    bool ok0 = !failed;    // uniform from here on
    int32_t q = 0;         // head of the backward bitstream, in bits from word 0 (uniform)
    const int32_t floor = static_cast<int32_t>(8 * mis);
    uint32_t state = 0;    // lanes 0..2: the lane's FSE state
    const ldsptr<zstd::FseCell> tab = lane == 0 ? t_of : (lane == 1 ? t_ml : t_ll);
    if (ok0 && z.nseq) {
      ok0 = bits_len != 0;
      if (ok0) {
        const uint32_t last_at = mis + bits_len - 1;
        refill(static_cast<int32_t>(last_at >> 2) + 1);
        const uint32_t last = (win[static_cast<int32_t>(last_at >> 2) - wlo] >> (8 * (last_at & 3u))) & 0xFFu;
        const uint32_t al_ll = s_al[0], al_of = s_al[1], al_ml = s_al[2];
        const uint32_t last_u = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(last)));
        ok0 = last_u != 0;
        if (ok0) {
          q = static_cast<int32_t>(8 * last_at + zstd::HighBit(last_u));
          ok0 = q - static_cast<int32_t>(al_ll + al_of + al_ml) >= floor;
        }
        if (ok0) {   // the initial states: literal length, offset, match length (sequences of <= 9 bits: inside the window)
          const int32_t q_ll = q - static_cast<int32_t>(al_ll), q_of = q_ll - static_cast<int32_t>(al_of), q_ml = q_of - static_cast<int32_t>(al_ml);
          state = lane == 0 ? field(q_of, al_of) : (lane == 1 ? field(q_ml, al_ml) : field(q_ll, al_ll));
          q = q_ml;
        }
      }
    }
    bool ok = ok0;
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
      const uint32_t ll = active ? s_trip[lane][0] : 0u, ml = active ? s_trip[lane][1] : 0u, ov = active ? s_trip[lane][2] : 1u;
      // the offset value as DecodeSequences emits it: > 3 an offset, 1..3 a repeat offset (shifted by one after an empty
      // literal run); ov == 0 only for the literals-only descriptor behind the last sequence
      const uint32_t code = ov > 3 ? ov - 3 : (ov == 0 ? 0u : (zstd::kRepMarker | (ov - 1 + (ll == 0 ? 1u : 0u))));
      const uint32_t incl_ll = wave_inclusive_scan_u32(ll), incl_out = wave_inclusive_scan_u32(ll + ml);
      const uint32_t lit_i = lit_base + incl_ll - ll, out_i = out_base + incl_out - (ll + ml);
      bool bad = active && (lit_i > z.lit_regen || ll > z.lit_regen - lit_i || out_i > zstd::kBlockMax || ll + ml > zstd::kBlockMax - out_i ||
                            (ov > 3 && ov - 3 >= zstd::kRepMarker));   // offset code 31: see DecodeSequences
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
    uint64_t t_step = 0, t_group = 0;
    for (uint32_t g0 = 0; g0 < z.nseq && ok; g0 += 64) {
      const uint32_t cnt = z.nseq - g0 < 64u ? z.nseq - g0 : 64u;
      const uint64_t t0 = probe ? wall_clock64() : 0;
      if (((q - kGroupBits) >> 5) < wlo) refill((q >> 5) + 1);   // uniform
      if (lane < 3) {
        // One step of the three lanes.  Everything the lanes share is scalar: the six field positions are computed on the scalar
        // unit and a lane picks its two.  (Dealing them with v_writelane instead of selects took the step from 36 to 25 vector
        // instructions and changed neither the kernel's time nor the scan's: the chain is latency, not issue.)  LAST: the
        // block's last sequence reads no state bits.
        const uint32_t win_bias = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(win)) - 4u * static_cast<uint32_t>(wlo);   // byte address of word 0, were it in the window
        auto word_pair = [&](int32_t at, uint32_t* lo_w, uint32_t* hi_w) {
          const ldsptr<uint32_t> w = (ldsptr<uint32_t>)(static_cast<uintptr_t>(4u * static_cast<uint32_t>(at >> 5) + win_bias));
          *lo_w = w[0];
          *hi_w = w[1];
        };
        auto step = [&](uint32_t i, auto last_tag) -> bool {
          constexpr bool LAST = decltype(last_tag)::value;
          const zstd::FseCell cell = tab[state];
          const uint32_t pack = static_cast<uint32_t>(cell >> 48);   // state bits | extra bits << 8
          const uint32_t k_of = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(pack), 0)),
                         k_ml = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(pack), 1)),
                         k_ll = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(pack), 2));
          // the fields below the head: offset extra, match-length extra, literal-length extra, then the bits of the next
          // literal-length, match-length and offset states
          const int32_t p_of_e = q - static_cast<int32_t>(k_of >> 8), p_ml_e = p_of_e - static_cast<int32_t>(k_ml >> 8),
                        p_ll_e = p_ml_e - static_cast<int32_t>(k_ll >> 8);
          const int32_t p_ll_b = LAST ? p_ll_e : p_ll_e - static_cast<int32_t>(k_ll & 0xFFu), p_ml_b = LAST ? p_ll_e : p_ll_b - static_cast<int32_t>(k_ml & 0xFFu),
                        p_of_b = LAST ? p_ll_e : p_ml_b - static_cast<int32_t>(k_of & 0xFFu);
          if (p_of_b < floor) return false;   // the stream ran out (uniform)
          const int32_t pe = lane == 0 ? p_of_e : (lane == 1 ? p_ml_e : p_ll_e);
          uint32_t e_lo, e_hi;
          word_pair(pe, &e_lo, &e_hi);
          uint32_t b_lo = 0, b_hi = 0;
          const int32_t pb = lane == 0 ? p_of_b : (lane == 1 ? p_ml_b : p_ll_b);
          if (!LAST) word_pair(pb, &b_lo, &b_hi);
          const uint32_t value = zstd::CellBase(cell) + __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(e_hi, e_lo, static_cast<uint32_t>(pe)), 0u, pack >> 8);
          if (!LAST) state = zstd::CellNext(cell) + __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(b_hi, b_lo, static_cast<uint32_t>(pb)), 0u, pack & 0xFFu);
          s_trip[i][2u - lane] = value;
          q = p_of_b;
          return true;
        };
        const uint32_t n_more = g0 + cnt == z.nseq ? cnt - 1 : cnt;   // sequences of this group that are followed by another one
#pragma clang loop unroll(disable)
        for (uint32_t i = 0; i < n_more && ok0; i++) ok0 = step(i, std::false_type());
        if (ok0 && n_more < cnt) ok0 = step(n_more, std::true_type());
      }
      // (q and ok0 were computed from v_readlane results inside the three-lane region: make them wave-uniform again)
      q = __builtin_amdgcn_readfirstlane(q);
      ok0 = __builtin_amdgcn_readfirstlane(ok0 ? 1 : 0) != 0;
      if (ok0 && g0 + cnt == z.nseq) ok0 = q == floor;
      __builtin_amdgcn_wave_barrier();
      ok = ok0;
      if (!ok) break;
      const uint64_t t1 = probe ? wall_clock64() : 0;
      group(cnt);
      __builtin_amdgcn_wave_barrier();
      if (probe) {
        t_step += t1 - t0;
        t_group += wall_clock64() - t1;
      }
    }
    if (probe && lane == 0)
      printf("zprobe seq block %u: tables %llu steps %llu groups %llu ticks, %u sequences, %u bytes of bitstream, %u literals\n", bi,
             (unsigned long long)(t_tables - t_start), (unsigned long long)t_step, (unsigned long long)t_group, z.nseq, bits_len, z.lit_regen);
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
