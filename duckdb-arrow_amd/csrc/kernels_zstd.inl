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
//   zstd_layout    one lane per buffer, its blocks in order: first output byte of every block, the size check of the reference
//                  (base_stream_reader.cpp:24-29), and the repeat offsets -- the only state that runs from block to block.
// From there on a ZSTD batch is an LZ4 batch: lz4_expand writes the link words, the resolve kernels follow them, lz4_emit
// writes the bytes.  The serial chains (one table lookup in LDS per symbol) are latency-bound and leave the chip almost idle:
// the scan runs the batches of several slots side by side.

constexpr int kZstdThreads = 128;

__global__ __launch_bounds__(kZstdThreads) void zstd_entropy(Lz4Args a) {
  __shared__ uint16_t s_huf[1u << zstd::kHufMaxBits];
  __shared__ zstd::FseCell s_ll[512], s_of[256], s_ml[512], s_wcells[64];
  __shared__ uint8_t s_weights[256];
  __shared__ int16_t s_counts[4][64];
  __shared__ uint16_t s_next[4][64];
  __shared__ uint32_t s_al[3], s_huf_bits, s_desc, s_fail;
  const uint32_t bi = blockIdx.x;
  const zstd::BlockInfo* zb = static_cast<const zstd::BlockInfo*>(a.zblocks);
  const zstd::BlockInfo z = zb[bi];
  const Lz4BlockDev b = a.blocks[bi];
  const uint8_t* comp = a.comp;
  uint8_t* arena = a.literals;
  const uint8_t* c = comp + z.comp_off;
  const uint32_t per = b.seq_cap / kParseLanes;   // descriptors per slice
  gptr<u32x4> seq = GM<u32x4>(a.seq) + b.seq_base;
  gptr<uint32_t> seq_off = GM<uint32_t>(a.seq_off) + b.seq_base;
  uint32_t* lane_out = a.lane_out + static_cast<size_t>(bi) * kParseLanes;
  uint32_t* lane_nseq = a.lane_nseq + static_cast<size_t>(bi) * kParseLanes;
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
  __syncthreads();
  // --- tables ---------------------------------------------------------------------------------------------------------
  if (wave == 0) {
    if (lane == 0 && z.lit_type >= 2) {
      const zstd::BlockInfo hs = zb[z.huf_src];
      uint32_t bits = 0;
      const uint32_t desc = zstd::ReadHuffmanTable(comp + hs.comp_off + hs.lit_hdr, hs.lit_comp, s_huf, &bits, s_weights, s_wcells, s_counts[3], s_next[3]);
      if (!desc) s_fail = 1;
      s_desc = z.lit_type == 2 ? desc : 0;
      s_huf_bits = bits;
    }
  } else if (lane < 3 && z.nseq) {
    const int t = static_cast<int>(lane);
    const zstd::BlockInfo sb = zb[t == 0 ? z.ll_src : t == 1 ? z.of_src : z.ml_src];
    const uint32_t so = sb.seq_pos + sb.seq_hdr;
    zstd::FseCell* tab = t == 0 ? s_ll : t == 1 ? s_of : s_ml;
    const uint32_t al = so < sb.comp_size ? zstd::BuildSequenceTable(comp + sb.comp_off + so, sb.comp_size - so, t, tab, s_counts[t], s_next[t]) : ~0u;
    if (al == ~0u) s_fail = 1;
    s_al[t] = al;
  }
  __syncthreads();
  const bool failed = s_fail != 0;   // uniform
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
      ok = ok && zstd::DecodeHuffmanStream(c + first, nbytes, nsym, s_huf, s_huf_bits, arena + z.lit_pos + out0);
      if (!ok) lz4_fail(a.status);   // the block's size is still reported by wave 1; the batch is rejected through the status word
    }
    return;
  }
  if (lane != 0) {
    return;
  }
  // wave 1, lane 0: the sequences
  uint32_t out_pos = 0, lit_used = 0, k = 0, j = 0, lane_base = 0, n_desc = 0;
  bool ok = !failed;
  auto put = [&](uint32_t ll, uint32_t ml, uint32_t off) {
    if (j == 0) {
      lane_base = out_pos;
      lane_out[k] = out_pos;
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
      k++;
      j = 0;
    }
  };
  if (ok && z.nseq) {
    const uint32_t so = z.seq_pos + z.seq_hdr;
    const uint32_t bo = zstd::SequenceBitstreamOffset(c + so, z.comp_size - so, s_counts[3]);
    ok = bo != 0 && so + bo < z.comp_size;
    if (ok)
      ok = zstd::DecodeSequences(c + so + bo, z.comp_size - so - bo, z.nseq, s_ll, s_al[0], s_of, s_al[1], s_ml, s_al[2],
                                 [&](uint32_t, uint32_t ll, uint32_t ml, uint32_t off) {
                                   if (ll > z.lit_regen - lit_used || ll + ml > zstd::kBlockMax - out_pos) return false;
                                   put(ll, ml, off);
                                   return true;
                                 });
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
  if (j) lane_nseq[k++] = j;
  for (; k < kParseLanes; k++) {   // empty slices begin where the block ends
    lane_out[k] = out_pos;
    lane_nseq[k] = 0;
  }
  a.block_out_size[bi] = out_pos;
  a.block_nseq[bi] = n_desc;
}

__global__ __launch_bounds__(64) void zstd_layout(Lz4Args a) {
  const uint32_t u = blockIdx.x * 64 + threadIdx.x;
  if (u >= a.n_buffers) return;
  const Lz4BufferDev f = a.buffers[u];
  const zstd::BlockInfo* zb = static_cast<const zstd::BlockInfo*>(a.zblocks);
  uint64_t at = f.out_off;
  uint32_t rep[3] = {1, 4, 8};
  bool ok = true;
  for (uint32_t kb = 0; kb < f.n_blocks; kb++) {
    const uint32_t bi = f.first_block + kb;
    a.block_out_base[bi] = at;
    at += a.block_out_size[bi];
    if (zb[bi].type != 2) continue;
    // the block's descriptors in order: slice after slice, `per` in each but the last
    const Lz4BlockDev b = a.blocks[bi];
    const uint32_t n = a.block_nseq[bi];
    gptr<uint32_t> so = GM<uint32_t>(a.seq_off) + b.seq_base;   // slices are dense: descriptor i sits at i
    uint32_t i = 0;
    for (; i + 4 <= n; i += 4) {   // the loads do not depend on the history: four in flight
      const uint32_t o0 = so[i], o1 = so[i + 1], o2 = so[i + 2], o3 = so[i + 3];
      uint32_t r;
      if (o0) { r = zstd::ResolveRepeat(o0, rep); ok &= r != 0; if (o0 >> 31) so[i] = r; }
      if (o1) { r = zstd::ResolveRepeat(o1, rep); ok &= r != 0; if (o1 >> 31) so[i + 1] = r; }
      if (o2) { r = zstd::ResolveRepeat(o2, rep); ok &= r != 0; if (o2 >> 31) so[i + 2] = r; }
      if (o3) { r = zstd::ResolveRepeat(o3, rep); ok &= r != 0; if (o3 >> 31) so[i + 3] = r; }
    }
    for (; i < n; i++) {
      const uint32_t o = so[i];
      if (!o) continue;
      const uint32_t r = zstd::ResolveRepeat(o, rep);
      ok &= r != 0;
      if (o >> 31) so[i] = r;
    }
  }
  ok = ok && at - f.out_off == f.out_len;
  a.buffer_ok[u] = ok ? 1u : 0u;
  if (!ok) lz4_fail(a.status);
}
