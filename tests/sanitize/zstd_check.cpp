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
struct Seq { uint32_t out, lit, ll, ml, off; };

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
      seqs[bi].push_back({0, z.lit_pos, 1, z.regen - 1, 1});
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
        if (!zstd::DecodeHuffmanStream(c + first, nbytes, nsym, huf, max_bits, lits.data() + z.lit_pos + out0)) { *why = "literal stream " + std::to_string(s) + " of block " + std::to_string(bi); return false; }
      }
    }
    uint32_t out_pos = 0, lit_used = 0;
    const bool scratch_lits = z.lit_type != 0;
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
      std::vector<Seq>& v = seqs[bi];
      const bool ok = zstd::DecodeSequences(c + so + bo, z.comp_size - so - bo, z.nseq, tll, al[0], tof, al[1], tml, al[2],
                                            [&](uint32_t, uint32_t ll, uint32_t ml, uint32_t off) {
                                              if (ll > z.lit_regen - lit_used || ll + ml > zstd::kBlockMax - out_pos) return false;
                                              v.push_back({out_pos, z.lit_pos + lit_used, ll, ml, off});
                                              lit_used += ll;
                                              out_pos += ll + ml;
                                              return true;
                                            });
      if (!ok) { *why = "sequences of block " + std::to_string(bi); return false; }
    }
    if (lit_used < z.lit_regen) {
      if (z.lit_regen - lit_used > zstd::kBlockMax - out_pos) { *why = "block too long"; return false; }
      seqs[bi].push_back({out_pos, z.lit_pos + lit_used, z.lit_regen - lit_used, 0, 0});
      out_pos += z.lit_regen - lit_used;
    }
    (void)scratch_lits;
    block_out[bi] = out_pos;
  }
  // stage 2: the frame's blocks in order -- output positions and repeat offsets; stage 3: the copies
  uint32_t rep[3] = {1, 4, 8};
  out->clear();
  for (size_t bi = 0; bi < blocks.size(); bi++) {
    const zstd::BlockInfo& z = infos[bi];
    if (z.type == 0) { out->insert(out->end(), comp + z.comp_off, comp + z.comp_off + z.comp_size); continue; }
    for (Seq& q : seqs[bi]) {
      uint32_t off = q.off;
      if (z.type == 2 && q.ml) {
        off = zstd::ResolveRepeat(q.off, rep);
        if (!off) { *why = "repeat offset"; return false; }
      }
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

int main(int argc, char** argv) {
  int bad = 0, n = 0;
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
