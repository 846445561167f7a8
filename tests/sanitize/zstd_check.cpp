// zstd_check.cpp -- CPU check of the ZSTD stages the K8 kernels run (test infrastructure, never shipped).
//
// duckdb-arrow_amd/csrc/zstd_format.hpp is compiled by hipcc into the kernels and by g++ into this program: the same
// functions decode the same frames here, block by block in the kernels' order of stages (host walk -> per-block entropy
// decode with tables taken from the blocks the walk names -> repeat offsets in frame order -> copy), and the result is compared
// with the bytes the frame was made from.  usage: zstd_check <frame file> <expected bytes file> ...   (pairs)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../duckdb-arrow_amd/csrc/ipc_stream_reader.hpp"
#include "../../duckdb-arrow_amd/csrc/zstd_format.hpp"

using namespace miarrow;

static std::vector<uint8_t> Slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static long g_seen[24];
struct Seq { uint32_t out, lit, ll, ml, off, slice; };
struct Rep { uint32_t s[3]; };

static bool DecodeFrame(const std::vector<uint8_t>& frame, size_t expect_size, std::vector<uint8_t>* out, std::string* why) {
  // the body as the device sees it: 8-byte aligned, padded
  std::vector<uint64_t> arena((frame.size() + 64 + 7) / 8 + 1, 0);
  uint8_t* comp = reinterpret_cast<uint8_t*>(arena.data());
  std::memcpy(comp, frame.data(), frame.size());
  DeferredLz4Body::Buffer buf;
  std::vector<DeferredLz4Body::Block> blocks;
  std::vector<zstd::BlockInfo> infos;
  uint32_t scratch_bytes = 0;
  if (!WalkZstdFrame(comp, 0, static_cast<int64_t>(frame.size()), 0, static_cast<int64_t>(expect_size), &buf, &blocks, &infos, &scratch_bytes)) {
    *why = "walk refused the frame";
    return false;
  }
  for (const zstd::BlockInfo& z : infos) {   // what the frames exercised (printed by main)
    g_seen[z.type]++;
    if (z.type != 2) continue;
    g_seen[4 + z.lit_type]++;
    if (z.lit_type >= 2) g_seen[8 + (z.lit_streams == 4)]++;
    if (z.lit_type == 2) g_seen[10 + (comp[z.comp_off + z.lit_hdr] >= 128)]++;
    if (z.nseq)
      for (int t = 0; t < 3; t++) g_seen[12 + 4 * t + ((comp[z.comp_off + z.seq_pos + z.seq_hdr] >> (6 - 2 * t)) & 3)]++;
  }
  std::vector<uint8_t> lits(scratch_bytes + 16);
  std::vector<std::vector<Seq>> seqs(blocks.size());
  std::vector<std::vector<Rep>> slice_fn(blocks.size());   // per block: the 256 slices' effect on the repeat offsets, then their starting state
  std::vector<uint32_t> block_out(blocks.size(), 0);
  static uint16_t huf[2048];
  static zstd::FseCell tll[512], tof[256], tml[512], wcells[64];
  static uint8_t weights[256];
  int16_t counts[64];
  uint16_t next[64];
  // stage 1: every block on its own
  for (size_t bi = 0; bi < blocks.size(); bi++) {
    const zstd::BlockInfo& z = infos[bi];
    const uint8_t* c = comp + z.comp_off;
    if (z.type == 0) { block_out[bi] = z.comp_size; continue; }
    if (z.type == 1) {
      lits[z.lit_pos] = c[0];
      seqs[bi].push_back({0, z.lit_pos, 1, z.regen - 1, 1, 0});
      block_out[bi] = z.regen;
      continue;
    }
    if (z.lit_type == 1) {
      std::memset(lits.data() + z.lit_pos, c[z.lit_hdr], z.lit_regen);
    } else if (z.lit_type >= 2) {
      const zstd::BlockInfo& hs = infos[z.huf_src];
      uint32_t max_bits = 0;
      const uint32_t desc = zstd::ReadHuffmanTable(comp + hs.comp_off + hs.lit_hdr, hs.lit_comp, huf, &max_bits, weights, wcells, counts, next);
      if (!desc) { *why = "huffman table of block " + std::to_string(bi); return false; }
      for (uint32_t s = 0; s < z.lit_streams; s++) {
        uint32_t first, nbytes, out0, nsym;
        if (!zstd::LiteralStream(z, c, z.lit_type == 2 ? desc : 0, s, &first, &nbytes, &out0, &nsym)) { *why = "literal streams of block " + std::to_string(bi); return false; }
        // the kernels' reader: a window of words refilled by the reader itself (a small one here, to cross it often)
        uint32_t window[8];
        zstd::BackBits<const uint8_t*, zstd::WindowWords<const uint8_t*, uint32_t*, 8>> br;
        br.src.win = window;
        if (!zstd::DecodeHuffmanStream(br, c + first, nbytes, nsym, huf, max_bits, lits.data() + z.lit_pos + out0)) { *why = "literal stream " + std::to_string(s) + " of block " + std::to_string(bi); return false; }
        // ... and the positional decoder of the staged path (the block addressable by aligned words): same literals
        {
          const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(c + first) & 3u);
          std::vector<uint32_t> words((mis + nbytes + 3) / 4 + 2, 0xA5A5A5A5u);   // whatever lies around the stream must not matter
          std::memcpy(reinterpret_cast<uint8_t*>(words.data()) + mis, c + first, nbytes);
          std::vector<uint8_t> again(nsym + 8, 0);
          uint8_t* dst = again.data() + ((4 - (reinterpret_cast<uintptr_t>(again.data()) & 3u)) & 3u) + ((reinterpret_cast<uintptr_t>(lits.data() + z.lit_pos + out0)) & 3u);
          if (!zstd::DecodeHuffmanStreamPos(words.data(), mis, nbytes, nsym, huf, max_bits, dst)) { *why = "positional literal stream " + std::to_string(s) + " of block " + std::to_string(bi); return false; }
          if (std::memcmp(dst, lits.data() + z.lit_pos + out0, nsym) != 0) { *why = "positional literal stream differs in block " + std::to_string(bi); return false; }
          // ... and through a sliding window of 8 words (the device's is 128): crossed every few symbols
          uint32_t win8[8];
          zstd::SlidingWords<const uint8_t*, uint32_t*, 8> sw;
          sw.Init(reinterpret_cast<const uint8_t*>(words.data()), win8);
          std::vector<uint8_t> third(nsym + 8, 0);
          uint8_t* dst3 = third.data() + (dst - again.data());
          if (!zstd::DecodeHuffmanStreamPos(sw, mis, nbytes, nsym, huf, max_bits, dst3) || std::memcmp(dst3, dst, nsym) != 0) {
            *why = "windowed positional literal stream differs in block " + std::to_string(bi);
            return false;
          }
        }
      }
    }
    uint32_t out_pos = 0, lit_used = 0;
    const bool scratch_lits = z.lit_type != 0;
    // the kernel's slices: 256 per block, `per` descriptors in each
    const uint32_t per = blocks[bi].seq_cap / 256;
    const Rep id = {{zstd::RepSlot(0), zstd::RepSlot(1), zstd::RepSlot(2)}};
    slice_fn[bi].assign(256, id);
    Rep S = id;
    uint32_t n_desc = 0;
    bool rep_ok = true;
    auto put = [&](uint32_t ll, uint32_t ml, uint32_t code) {
      uint32_t off = 0;
      if (ml) {
        off = zstd::RepStep(code, S.s);
        if (!off) rep_ok = false;
      }
      const uint32_t k = n_desc / per;
      seqs[bi].push_back({out_pos, z.lit_pos + lit_used, ll, ml, off, k});
      lit_used += ll;
      out_pos += ll + ml;
      n_desc++;
      if (n_desc % per == 0) {
        slice_fn[bi][k] = S;
        S = id;
      }
    };
    if (z.nseq) {
      const zstd::BlockInfo* src[3] = {&infos[z.ll_src], &infos[z.of_src], &infos[z.ml_src]};
      zstd::FseCell* tab[3] = {tll, tof, tml};
      uint32_t al[3];
      for (int t = 0; t < 3; t++) {
        const zstd::BlockInfo& sb = *src[t];
        const uint32_t so = sb.seq_pos + sb.seq_hdr;
        al[t] = zstd::BuildSequenceTable(comp + sb.comp_off + so, sb.comp_size - so, t, tab[t], counts, next);
        if (al[t] == ~0u) { *why = "sequence table " + std::to_string(t) + " of block " + std::to_string(bi); return false; }
      }
      const uint32_t so = z.seq_pos + z.seq_hdr;
      const uint32_t bo = zstd::SequenceBitstreamOffset(c + so, z.comp_size - so, counts);
      if (!bo || so + bo >= z.comp_size) { *why = "sequence bitstream of block " + std::to_string(bi); return false; }
      uint32_t seq_window[16];
      zstd::BackBits<const uint8_t*, zstd::WindowWords<const uint8_t*, uint32_t*, 16>> sbr;
      sbr.src.win = seq_window;
      const bool ok = zstd::DecodeSequences(sbr, c + so + bo, z.comp_size - so - bo, z.nseq, tll, al[0], tof, al[1], tml, al[2],
                                            [&](uint32_t, uint32_t ll, uint32_t ml, uint32_t code) {
                                              if (ll > z.lit_regen - lit_used || ll + ml > zstd::kBlockMax - out_pos) return false;
                                              put(ll, ml, code);
                                              return rep_ok;
                                            });
      if (!ok) { *why = "sequences of block " + std::to_string(bi); return false; }
      // the positional decoder of the staged path: the same (literal length, match length, offset code) triples
      {
        const uint8_t* first = c + so + bo;
        const uint32_t nbytes = z.comp_size - so - bo;
        const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(first) & 3u);
        std::vector<uint32_t> words((mis + nbytes + 3) / 4 + 2, 0x5A5A5A5Au);
        std::memcpy(reinterpret_cast<uint8_t*>(words.data()) + mis, first, nbytes);
        std::vector<uint32_t> a, b;
        zstd::BackBits<const uint8_t*> plain;
        const bool ok1 = zstd::DecodeSequences(plain, first, nbytes, z.nseq, tll, al[0], tof, al[1], tml, al[2],
                                               [&](uint32_t, uint32_t ll, uint32_t ml, uint32_t code) { a.insert(a.end(), {ll, ml, code}); return true; });
        const bool ok2 = zstd::DecodeSequencesPos(words.data(), mis, nbytes, z.nseq, tll, al[0], tof, al[1], tml, al[2],
                                                  [&](uint32_t, uint32_t ll, uint32_t ml, uint32_t code) { b.insert(b.end(), {ll, ml, code}); return true; });
        if (!ok1 || !ok2 || a != b) { *why = "positional sequence decoder differs in block " + std::to_string(bi); return false; }
        uint32_t win8[8];
        zstd::SlidingWords<const uint8_t*, uint32_t*, 8> sw;
        sw.Init(reinterpret_cast<const uint8_t*>(words.data()), win8);
        std::vector<uint32_t> c3;
        const bool ok3 = zstd::DecodeSequencesPos(sw, mis, nbytes, z.nseq, tll, al[0], tof, al[1], tml, al[2],
                                                  [&](uint32_t, uint32_t ll, uint32_t ml, uint32_t code) { c3.insert(c3.end(), {ll, ml, code}); return true; });
        if (!ok3 || a != c3) { *why = "windowed positional sequence decoder differs in block " + std::to_string(bi); return false; }
      }
    }
    if (lit_used < z.lit_regen) {
      if (z.lit_regen - lit_used > zstd::kBlockMax - out_pos) { *why = "block too long"; return false; }
      put(z.lit_regen - lit_used, 0, 0);
    }
    if (n_desc % per) slice_fn[bi][n_desc / per] = S;
    (void)scratch_lits;
    block_out[bi] = out_pos;
  }
  // stage 2: the frame's blocks in order -- the repeat offsets every slice starts from; stage 3: the copies
  Rep R = {{1, 4, 8}};
  out->clear();
  for (size_t bi = 0; bi < blocks.size(); bi++) {
    const zstd::BlockInfo& z = infos[bi];
    if (z.type == 0) { out->insert(out->end(), comp + z.comp_off, comp + z.comp_off + z.comp_size); continue; }
    if (z.type == 2) {
      // as zstd_layout does it: 64 lanes, 4 slices each, an inclusive scan of the lanes' totals (shuffles emulated)
      auto after = [](const Rep& g, const Rep& st) {
        Rep r;
        for (int q = 0; q < 3; q++) r.s[q] = zstd::RepResolve(g.s[q], st.s[0], st.s[1], st.s[2]);
        return r;
      };
      std::vector<Rep>& fn = slice_fn[bi];
      Rep inc[64];
      for (int l = 0; l < 64; l++) inc[l] = after(fn[4 * l + 3], after(fn[4 * l + 2], after(fn[4 * l + 1], fn[4 * l])));
      for (int d = 1; d < 64; d <<= 1) {
        Rep prev[64];
        for (int l = 0; l < 64; l++) prev[l] = inc[l >= d ? l - d : l];
        for (int l = d; l < 64; l++) inc[l] = after(inc[l], prev[l]);
      }
      std::vector<Rep> start(256);
      for (int l = 0; l < 64; l++) {
        const Rep e = l ? inc[l - 1] : Rep{{zstd::RepSlot(0), zstd::RepSlot(1), zstd::RepSlot(2)}};
        start[4 * l] = after(e, R);
        for (int q = 1; q < 4; q++) start[4 * l + q] = after(fn[4 * l + q - 1], start[4 * l + q - 1]);
      }
      R = after(inc[63], R);
      fn = start;
    }
    if (std::getenv("ZCHECK_DUMP")) std::printf("block %zu type %u ndesc %zu\n", bi, z.type, seqs[bi].size());
    for (Seq& q : seqs[bi]) {
      uint32_t off = q.off;
      if (z.type == 2 && q.ml) {
        const Rep& h = slice_fn[bi][q.slice];
        off = zstd::RepResolve(q.off, h.s[0], h.s[1], h.s[2]);
        if (!off || (off >> 31)) { *why = "repeat offset"; return false; }
      }
      if (std::getenv("ZCHECK_DUMP")) std::printf("d %u %u %u %u %u\n", q.out, q.ll, q.ml, q.ml ? off : 0, q.off >> 31);
      const uint8_t* lsrc = (z.type == 2 && z.lit_type == 0) ? comp : lits.data();
      out->insert(out->end(), lsrc + q.lit, lsrc + q.lit + q.ll);
      if (q.ml) {
        if (off > out->size()) { *why = "offset reaches in front of the buffer"; return false; }
        for (uint32_t i = 0; i < q.ml; i++) out->push_back((*out)[out->size() - off]);
      }
    }
  }
  return true;
}

// A damaged frame: offset code 31 (RLE offset table) with all extra bits set gives an offset value of 2^32 - 1; minus 3 it has
// bit 31 set, which the emitted word reserves for "repeat offset".  libzstd reports corruption_detected; so must this decoder.
static bool OffsetCode31IsRejected() {
  const uint8_t section[] = {0x54, 0, 31, 0, 0xFF, 0xFF, 0xFF, 0xFF};   // modes: LL / OF / ML all RLE; symbols; one sequence's bits
  zstd::FseCell tll[1 << 9], tof[1 << 8], tml[1 << 9];
  int16_t counts[64];
  uint16_t next[64];
  zstd::FseCell* tab[3] = {tll, tof, tml};
  uint32_t al[3];
  for (int t = 0; t < 3; t++) {
    al[t] = zstd::BuildSequenceTable(section, sizeof(section), t, tab[t], counts, next);
    if (al[t] != 0) return false;
  }
  const uint32_t bo = zstd::SequenceBitstreamOffset(section, sizeof(section), counts);
  if (bo != 4) return false;
  uint32_t seq_window[16];
  zstd::BackBits<const uint8_t*, zstd::WindowWords<const uint8_t*, uint32_t*, 16>> sbr;
  sbr.src.win = seq_window;
  bool emitted = false;
  const bool ok = zstd::DecodeSequences(sbr, section + bo, static_cast<uint32_t>(sizeof(section)) - bo, 1, tll, al[0], tof, al[1], tml, al[2],
                                        [&](uint32_t, uint32_t, uint32_t, uint32_t) { emitted = true; return true; });
  bool emitted2 = false;
  uint32_t words[4] = {0, 0, 0, 0};
  std::memcpy(words, section + bo, sizeof(section) - bo);
  const bool ok2 = zstd::DecodeSequencesPos(words, 0u, static_cast<uint32_t>(sizeof(section)) - bo, 1, tll, al[0], tof, al[1], tml, al[2],
                                            [&](uint32_t, uint32_t, uint32_t, uint32_t) { emitted2 = true; return true; });
  return !ok && !emitted && !ok2 && !emitted2;
}

int main(int argc, char** argv) {
  int bad = 0, n = 0;
  if (!OffsetCode31IsRejected()) {
    std::printf("FAIL offset code 31 was decoded as a repeat offset\n");
    bad++;
  }
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::vector<uint8_t> frame = Slurp(argv[i]), expect = Slurp(argv[i + 1]);
    std::vector<uint8_t> got;
    std::string why;
    n++;
    if (!DecodeFrame(frame, expect.size(), &got, &why)) {
      std::printf("FAIL %s: %s\n", argv[i], why.c_str());
      bad++;
    } else if (got != expect) {
      size_t k = 0;
      while (k < got.size() && k < expect.size() && got[k] == expect[k]) k++;
      std::printf("FAIL %s: %zu bytes, expected %zu, first difference at %zu\n", argv[i], got.size(), expect.size(), k);
      bad++;
    }
  }
  std::printf("blocks raw/rle/compressed %ld/%ld/%ld; literals raw/rle/huffman/treeless %ld/%ld/%ld/%ld; 1/4 streams %ld/%ld; weights fse/direct %ld/%ld\n",
              g_seen[0], g_seen[1], g_seen[2], g_seen[4], g_seen[5], g_seen[6], g_seen[7], g_seen[8], g_seen[9], g_seen[10], g_seen[11]);
  for (int t = 0; t < 3; t++)
    std::printf("table %d predefined/rle/fse/repeat %ld/%ld/%ld/%ld\n", t, g_seen[12 + 4 * t], g_seen[13 + 4 * t], g_seen[14 + 4 * t], g_seen[15 + 4 * t]);
  std::printf("%d frames, %d failed\n", n, bad);
  return bad ? 1 : 0;
}
