// zstd_format.hpp -- the entropy stage of a Zstandard block (RFC 8878), written once for two compilers: hipcc builds it
// into the K8 kernels (kernels_zstd.inl), g++ builds the same text into the CPU check of tests/sanitize/zstd_check.cpp, where
// its result is compared with libzstd's on the same frames.  Nothing here allocates or touches a global: callers hand in
// the tables (LDS on the device, the stack on the host).
//
// The reference decompresses ZSTD buffers with DuckDB's bundled zstd on the CPU (DuckDBDecompressZstd,
// src/ipc/stream_reader/base_stream_reader.cpp:11-32); the algorithm restated here is the published format:
//   block        = literals section + sequences section
//   literals     = raw | RLE | Huffman-coded (1 or 4 backward bitstreams; the code lengths ("weights") direct or FSE-coded)
//   sequences    = count, 3 table modes (predefined | RLE | FSE description | repeat), one backward bitstream of
//                  interleaved FSE states: {literal length, match length, offset} per sequence
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define MI_ZHD __host__ __device__ inline __attribute__((always_inline))
#else
#define MI_ZHD inline
#endif

namespace miarrow {
namespace zstd {

constexpr int kLL = 0, kOF = 1, kML = 2, kWeights = 3;
constexpr uint32_t kBlockMax = 128u << 10;          // Block_Maximum_Size
constexpr uint32_t kHufMaxBits = 11;                // literals: Max_Number_of_Bits
constexpr uint32_t kRepMarker = 0x80000000u;        // a sequence offset that names a repeat offset: kRepMarker | 0..3

// One block of a frame as the host walk (WalkZstdFrame, ipc_stream_reader.cpp) finds it from the headers alone: the block
// header, the literals section header, the sequence count and the three table modes.  Tables a block inherits (Huffman
// "treeless", FSE "repeat") are named by the index of the earlier block whose bytes describe them, so every block can be
// decoded without waiting for another.
struct BlockInfo {
  uint32_t comp_off, comp_size;              // the block's content inside the compressed body
  uint32_t type;                             // 0 raw, 1 RLE, 2 compressed
  uint32_t regen;                            // RLE: the bytes it expands to
  uint32_t lit_type;                         // 0 raw, 1 RLE, 2 Huffman, 3 Huffman with the table of block huf_src
  uint32_t lit_streams;                      // 1 or 4
  uint32_t lit_hdr, lit_regen, lit_comp;     // header bytes; decoded size; stored size behind the header
  uint32_t lit_pos;                          // where the copy stage reads the literals: raw -> inside the body, else scratch
  uint32_t seq_pos, seq_hdr, nseq;           // sequences section (from comp_off), bytes of its count, the count
  uint32_t huf_src, ll_src, of_src, ml_src;  // index of the block whose bytes hold the table in use (itself or an earlier one)
  uint32_t _pad;
};

MI_ZHD uint32_t MaxLog(int type) { return type == kLL ? 9u : type == kOF ? 8u : type == kML ? 9u : 6u; }
MI_ZHD uint32_t MaxSym(int type) { return type == kLL ? 35u : type == kOF ? 31u : type == kML ? 52u : 15u; }
MI_ZHD uint32_t HighBit(uint32_t v) { return 31u - static_cast<uint32_t>(__builtin_clz(v)); }   // v != 0

// {base value, extra bits} of a literal-length / match-length code; an offset code c is {1 << c, c}
MI_ZHD uint32_t LlBase(uint32_t c) {
  constexpr uint32_t t[20] = {16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
  return c < 16 ? c : t[c - 16];
}
MI_ZHD uint32_t LlBits(uint32_t c) {
  constexpr uint8_t t[20] = {1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  return c < 16 ? 0 : t[c - 16];
}
MI_ZHD uint32_t MlBase(uint32_t c) {
  constexpr uint32_t t[21] = {35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
  return c < 32 ? c + 3 : t[c - 32];
}
MI_ZHD uint32_t MlBits(uint32_t c) {
  constexpr uint8_t t[21] = {1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  return c < 32 ? 0 : t[c - 32];
}
// the predefined distributions (accuracy 6 / 5 / 6)
MI_ZHD int32_t DefaultCount(int type, uint32_t s) {
  constexpr int8_t ll[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
  constexpr int8_t of[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
  if (type == kLL) return ll[s];
  if (type == kOF) return of[s];
  // match lengths: {1, 4, 3, 2 x 6, 1 x 37, -1 x 7}
  return s == 0 ? 1 : s == 1 ? 4 : s == 2 ? 3 : s < 9 ? 2 : s < 46 ? 1 : -1;
}
MI_ZHD uint32_t DefaultLog(int type) { return type == kOF ? 5u : 6u; }
MI_ZHD uint32_t DefaultSymbols(int type) { return type == kLL ? 36u : type == kOF ? 29u : 53u; }

// One cell of an FSE decoding table (the state IS the index of the cell), packed into 8 bytes so that a table lives in any
// address space as plain integers: base value of the code (weights: the symbol) | next << 32 | nbits << 48 | extra << 56, with
// next state = next + the `nbits` bits read, and `extra` = additional bits of the code's value.
using FseCell = uint64_t;
MI_ZHD FseCell MakeCell(uint32_t base, uint32_t next, uint32_t nbits, uint32_t extra) {
  return static_cast<uint64_t>(base) | (static_cast<uint64_t>(next) << 32) | (static_cast<uint64_t>(nbits) << 48) | (static_cast<uint64_t>(extra) << 56);
}
MI_ZHD uint32_t CellBase(FseCell c) { return static_cast<uint32_t>(c); }
MI_ZHD uint32_t CellNext(FseCell c) { return static_cast<uint32_t>(c >> 32) & 0xFFFFu; }
MI_ZHD uint32_t CellBits(FseCell c) { return static_cast<uint32_t>(c >> 48) & 0xFFu; }
MI_ZHD uint32_t CellExtra(FseCell c) { return static_cast<uint32_t>(c >> 56); }

// ---------------------------------------------------------------------------------------------------------------------
// Backward bitstream: the last byte's highest set bit ends the stream, bits are taken from just below it towards the first
// byte.  The unread bits sit at the TOP of a 64-bit buffer (a peek is one shift), a 32-bit word enters below them whenever 32
// or fewer are left.  Words are ALIGNED loads, counted from the 4-byte boundary at or before the first byte; what lies in front
// of the stream reads as zero -- the format's rule for a stream that runs out (the caller sees Left() < 0).
// Where the words come from is a policy: DirectWords loads each one from the stream when it is needed (short streams, streams
// staged in LDS); WindowWords keeps the next W words in a small window of fast memory that the reader refills itself, W words
// at a time.  On the GPU the window is LDS: a decode loop that both loads from and stores to HBM would wait for its last store
// at every refill (loads and stores retire through one counter), with the window it touches HBM once per W words.

// How aligned words are loaded / stored through a byte pointer of type P.  Plain pointers here; the device build adds the
// specialisations for global- and LDS-address-space pointers (kernels_lz4.hip), so that the serial loops below never issue a
// FLAT access: a flat load or store counts on both wait counters, and every table lookup in LDS would wait for the last store.
template <typename P>
struct Mem {
  static MI_ZHD uint32_t Load32(P aligned) { return *reinterpret_cast<const uint32_t*>(aligned); }
  static MI_ZHD void Store32(P aligned, uint32_t v) { *reinterpret_cast<uint32_t*>(aligned) = v; }
  static MI_ZHD uintptr_t Address(P p) { return reinterpret_cast<uintptr_t>(p); }
};

template <typename BP>
struct DirectWords {
  BP base;
  MI_ZHD void Init(BP b, int32_t) { base = b; }
  MI_ZHD uint32_t Get(int32_t idx) { return Mem<BP>::Load32(base + 4 * (idx < 0 ? 0 : idx)); }   // always a readable address
};
template <typename BP, typename LP, int W>   // LP: pointer to W uint32 of fast memory, set by the caller before Open
struct WindowWords {
  BP base;
  LP win;
  int32_t lo;   // the window holds the words lo .. lo + W - 1
  MI_ZHD void Init(BP b, int32_t top) {
    base = b;
    lo = top + 1;   // nothing yet
  }
  MI_ZHD uint32_t Get(int32_t idx) {   // idx only ever goes down
    if (idx < lo) {
      lo = idx - (W - 1);
      for (int j = 0; j < W; j++) win[j] = Mem<BP>::Load32(base + 4 * (lo + j < 0 ? 0 : lo + j));
    }
    return win[idx - lo];
  }
};

template <typename BP, typename SRC = DirectWords<BP>>
struct BackBits {
  SRC src;
  uint64_t buf;            // unread bits, the next one in bit 63
  int32_t cnt;             // valid bits in buf: 33 .. 64 between calls (while the stream lasts)
  int32_t left;            // unread bits of the stream (negative: more were taken than it holds)
  int32_t next;            // index of the word that enters at the next refill (negative: in front of the stream)
  uint32_t first_mask;     // word 0 without the bytes in front of the stream

  MI_ZHD uint32_t Word(int32_t idx) {
    const uint32_t w = src.Get(idx);
    return idx > 0 ? w : idx == 0 ? (w & first_mask) : 0u;
  }
  MI_ZHD void Refill() {   // cnt <= 32
    buf |= static_cast<uint64_t>(Word(next)) << (32 - cnt);
    cnt += 32;
    next--;
  }
  // false: the stream is empty or its last byte is zero (no end mark)
  MI_ZHD bool Open(BP first, uint32_t nbytes) {
    if (nbytes == 0) return false;
    const uint8_t last = first[nbytes - 1];
    if (last == 0) return false;
    const uint32_t mis = static_cast<uint32_t>(Mem<BP>::Address(first) & 3u);
    first_mask = ~0u << (8 * mis);
    left = 8 * static_cast<int32_t>(nbytes - 1) + static_cast<int32_t>(HighBit(last));
    const int32_t p = left + 8 * static_cast<int32_t>(mis);    // position of the read head, counted from the aligned base
    const int32_t top = p > 0 ? (p - 1) >> 5 : 0;                // word of the first bit to read
    const int32_t r = p - 32 * top;                             // bits of that word below the head: 1 .. 32 (0: no bit at all)
    src.Init(first - mis, top);
    const uint32_t w = Word(top);
    buf = r > 0 ? static_cast<uint64_t>(r == 32 ? w : (w & ((1u << r) - 1u))) << (64 - r) : 0;
    cnt = r;
    next = top - 1;
    Refill();
    if (cnt <= 32) Refill();
    return true;
  }
  MI_ZHD int32_t Left() const { return left; }
  MI_ZHD uint32_t Peek(uint32_t n) const { return static_cast<uint32_t>((buf >> 1) >> (63 - n)); }   // n <= 32; 0 gives 0
  MI_ZHD void Skip(uint32_t n) {   // n <= 32
    buf <<= n;
    cnt -= static_cast<int32_t>(n);
    left -= static_cast<int32_t>(n);
    if (cnt <= 32) Refill();
  }
  MI_ZHD uint32_t Read(uint32_t n) {
    const uint32_t v = Peek(n);
    Skip(n);
    return v;
  }
};

// Forward bits of a table description (a few dozen bytes): byte loads, no state worth keeping.
template <typename BP>
struct FwdBits {
  BP src;
  uint32_t pos;
  MI_ZHD uint32_t Peek(uint32_t n) const {   // n <= 16
    const uint32_t b = pos >> 3;
    const uint32_t v = static_cast<uint32_t>(src[b]) | (static_cast<uint32_t>(src[b + 1]) << 8) | (static_cast<uint32_t>(src[b + 2]) << 16);
    return (v >> (pos & 7u)) & ((1u << n) - 1u);
  }
  MI_ZHD uint32_t Read(uint32_t n) {
    const uint32_t v = Peek(n);
    pos += n;
    return v;
  }
};

// FSE table description -> normalized counts (-1 = "less than one").  Returns the bytes it occupies, 0 = malformed.
// `src` must be readable for 3 bytes past `avail` (the bodies carry that much padding).
template <typename BP, typename CP>
MI_ZHD uint32_t ReadNCount(BP src, uint32_t avail, int type, CP counts, uint32_t* log_out, uint32_t* nsym_out) {
  if (avail == 0) return 0;
  const uint32_t max_sym = MaxSym(type);
  FwdBits<BP> f{src, 0};
  const uint32_t al = 5 + f.Read(4);
  if (al > MaxLog(type)) return 0;
  int32_t remaining = (1 << al) + 1, threshold = 1 << al;
  uint32_t nb = al + 1, sym = 0;
  bool prev0 = false;
  for (uint32_t s = 0; s <= max_sym; s++) counts[s] = 0;
  while (remaining > 1 && sym <= max_sym) {
    if ((f.pos >> 3) > avail) return 0;
    if (prev0) {
      uint32_t r;
      do {
        r = f.Read(2);
        sym += r;
        if ((f.pos >> 3) > avail) return 0;
      } while (r == 3 && sym <= max_sym);
      if (sym > max_sym) return 0;
    }
    const int32_t max = (2 * threshold - 1) - remaining;
    const uint32_t peek = f.Peek(nb);
    int32_t count;
    if (static_cast<int32_t>(peek & static_cast<uint32_t>(threshold - 1)) < max) {
      count = static_cast<int32_t>(peek & static_cast<uint32_t>(threshold - 1));
      f.pos += nb - 1;
    } else {
      count = static_cast<int32_t>(peek & static_cast<uint32_t>(2 * threshold - 1));
      if (count >= threshold) count -= max;
      f.pos += nb;
    }
    count--;   // the stored value is the count + 1
    remaining -= count < 0 ? -count : count;
    counts[sym++] = static_cast<int16_t>(count);
    prev0 = count == 0;
    while (remaining < threshold) {
      nb--;
      threshold >>= 1;
    }
  }
  if (remaining != 1) return 0;
  const uint32_t bytes = (f.pos + 7) >> 3;
  if (bytes > avail) return 0;
  *log_out = al;
  *nsym_out = sym;
  return bytes;
}

// Normalized counts -> decoding table of 1 << al cells.  `next` is scratch for one uint16 per symbol.  Two steps: the spread
// (which symbol a cell decodes; serial: the walk over the table skips the cells of the "less than one" symbols) and the cells
// themselves -- the k-th cell of a symbol in TABLE order gets state count + k, from which its bit count and base follow.  The
// device runs the second step with the whole wave (kernels_zstd.inl); FinishFseCell is the part both share.
template <typename CP, typename TP, typename NP>
MI_ZHD bool BuildFseSpread(CP counts, uint32_t nsym, uint32_t al, TP table, NP next) {
  const uint32_t size = 1u << al, mask = size - 1;
  uint32_t high = size - 1;
  for (uint32_t s = 0; s < nsym; s++) {
    if (counts[s] == -1) {
      table[high--] = s;   // until FinishFseCell a cell holds its symbol
      next[s] = 1;
    } else {
      next[s] = static_cast<uint16_t>(counts[s]);
    }
  }
  const uint32_t step = (size >> 1) + (size >> 3) + 3;
  uint32_t pos = 0;
  for (uint32_t s = 0; s < nsym; s++) {
    for (int32_t i = 0; i < counts[s]; i++) {
      table[pos] = s;
      do pos = (pos + step) & mask; while (pos > high);
    }
  }
  return pos == 0;
}
// the cell of symbol s whose state counter stands at ns
MI_ZHD FseCell FinishFseCell(uint32_t s, uint32_t ns, uint32_t al, int type) {
  const uint32_t nbits = al - HighBit(ns);
  const uint32_t nx = (ns << nbits) - (1u << al);
  if (type == kLL) return MakeCell(LlBase(s), nx, nbits, LlBits(s));
  if (type == kML) return MakeCell(MlBase(s), nx, nbits, MlBits(s));
  if (type == kOF) return MakeCell(1u << s, nx, nbits, s);
  return MakeCell(s, nx, nbits, 0);
}
template <typename CP, typename TP, typename NP>
MI_ZHD bool BuildFseTable(CP counts, uint32_t nsym, uint32_t al, int type, TP table, NP next) {
  if (!BuildFseSpread(counts, nsym, al, table, next)) return false;
  const uint32_t size = 1u << al;
  for (uint32_t u = 0; u < size; u++) {
    const uint32_t s = static_cast<uint32_t>(table[u]);
    const uint32_t ns = next[s];
    next[s] = static_cast<uint16_t>(ns + 1);
    table[u] = FinishFseCell(s, ns, al, type);
  }
  return true;
}
template <typename TP>
MI_ZHD void BuildRleTable(uint32_t s, int type, TP table) {
  table[0] = type == kLL ? MakeCell(LlBase(s), 0, 0, LlBits(s)) : type == kML ? MakeCell(MlBase(s), 0, 0, MlBits(s)) : MakeCell(1u << s, 0, 0, s);
}

// One of the three sequence tables of a block from the block's own bytes.  `seq` = the sequences section behind its count
// (at the modes byte), `avail` = bytes from there to the end of the block; mode 3 (repeat) is resolved by the caller (it
// passes the earlier block the table comes from).  Returns the accuracy log, ~0u = malformed.  counts/next: scratch
// (53 entries are enough).
// FINISH = false: the table is left after the spread (cells hold symbols, next[] the state counters) and *nsym_out says how
// many symbols it has -- 0 for an RLE table, which is final as it is.
template <bool FINISH = true, typename BP, typename TP, typename CP, typename NP>
MI_ZHD uint32_t BuildSequenceTable(BP seq, uint32_t avail, int type, TP table, CP counts, NP next, uint32_t* nsym_out = nullptr) {
  if (nsym_out) *nsym_out = 0;
  if (avail < 1) return ~0u;
  const uint32_t modes = seq[0];
  uint32_t at = 1;
  for (int t = 0; t <= type; t++) {   // the descriptions lie in the order LL, OF, ML
    const uint32_t mode = (modes >> (6 - 2 * t)) & 3u;
    if (t < type) {
      if (mode == 1) at += 1;
      else if (mode == 2) {
        uint32_t al, ns;
        const uint32_t n = at < avail ? ReadNCount(seq + at, avail - at, t, counts, &al, &ns) : 0;
        if (!n) return ~0u;
        at += n;
      }
      continue;
    }
    if (mode == 0) {
      const uint32_t ns = DefaultSymbols(type);
      for (uint32_t s = 0; s < ns; s++) counts[s] = static_cast<int16_t>(DefaultCount(type, s));
      if (!FINISH) {
        if (nsym_out) *nsym_out = ns;
        return BuildFseSpread(counts, ns, DefaultLog(type), table, next) ? DefaultLog(type) : ~0u;
      }
      return BuildFseTable(counts, ns, DefaultLog(type), type, table, next) ? DefaultLog(type) : ~0u;
    }
    if (mode == 1) {
      if (at >= avail || seq[at] > MaxSym(type)) return ~0u;
      BuildRleTable(seq[at], type, table);
      return 0;
    }
    if (mode == 2) {
      uint32_t al, ns;
      const uint32_t n = at < avail ? ReadNCount(seq + at, avail - at, type, counts, &al, &ns) : 0;
      if (!n) return ~0u;
      if (!FINISH) {
        if (nsym_out) *nsym_out = ns;
        return BuildFseSpread(counts, ns, al, table, next) ? al : ~0u;
      }
      return BuildFseTable(counts, ns, al, type, table, next) ? al : ~0u;
    }
    return ~0u;   // repeat of a repeat: the host walk resolves chains, so this is a malformed frame
  }
  return ~0u;
}
// Where the bitstream of the sequences section begins, counted from the modes byte (0 = malformed).
template <typename BP, typename CP>
MI_ZHD uint32_t SequenceBitstreamOffset(BP seq, uint32_t avail, CP counts) {
  if (avail < 1) return 0;
  const uint32_t modes = seq[0];
  if (modes & 3u) return 0;   // reserved bits
  uint32_t at = 1;
  for (int t = 0; t < 3; t++) {
    const uint32_t mode = (modes >> (6 - 2 * t)) & 3u;
    if (mode == 1) at += 1;
    else if (mode == 2) {
      uint32_t al, ns;
      const uint32_t n = at < avail ? ReadNCount(seq + at, avail - at, t, counts, &al, &ns) : 0;
      if (!n) return 0;
      at += n;
    }
  }
  return at <= avail ? at : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Huffman tree description -> decoding table of 1 << max_bits cells {symbol | nbits << 8}.  Returns the bytes of the
// description (0 = malformed).  weights: 256 bytes of scratch; cells / counts / next: scratch for the weights' own FSE table
// (64 cells, 16 counts).
// Two steps, like the FSE tables: the weights (serial: direct nibbles, or an FSE stream of two alternating states) and the
// table fill, which the device runs with the whole wave (kernels_zstd.inl).
template <typename BP, typename WP, typename TP, typename CP, typename NP>
MI_ZHD uint32_t ReadHuffmanWeights(BP src, uint32_t avail, uint32_t* n_out, uint32_t* max_bits_out, WP weights, TP cells, CP counts, NP next) {
  if (avail < 1) return 0;
  const uint32_t hb = src[0];
  uint32_t n = 0, used;
  if (hb >= 128) {   // 4-bit weights, the first of a pair in the high nibble
    n = hb - 127;
    used = 1 + (n + 1) / 2;
    if (used > avail) return 0;
    for (uint32_t i = 0; i < n; i++) weights[i] = (i & 1u) ? (src[1 + i / 2] & 15u) : (src[1 + i / 2] >> 4);
  } else {           // FSE-coded weights: two states take turns until the stream has run out
    used = 1 + hb;
    if (hb < 2 || used > avail) return 0;
    uint32_t al, ns;
    const uint32_t hdr = ReadNCount(src + 1, hb, kWeights, counts, &al, &ns);
    if (!hdr || hdr >= hb) return 0;
    if (!BuildFseTable(counts, ns, al, kWeights, cells, next)) return 0;
    BackBits<BP> br;
    if (!br.Open(src + 1 + hdr, hb - hdr)) return 0;
    uint32_t s1 = br.Read(al), s2 = br.Read(al);
    if (br.Left() < 0) return 0;
    for (;;) {
      if (n > 253) return 0;
      FseCell c = cells[s1];
      weights[n++] = static_cast<uint8_t>(CellBase(c));
      s1 = CellNext(c) + br.Read(CellBits(c));
      if (br.Left() < 0) {
        weights[n++] = static_cast<uint8_t>(CellBase(cells[s2]));
        break;
      }
      if (n > 253) return 0;
      c = cells[s2];
      weights[n++] = static_cast<uint8_t>(CellBase(c));
      s2 = CellNext(c) + br.Read(CellBits(c));
      if (br.Left() < 0) {
        weights[n++] = static_cast<uint8_t>(CellBase(cells[s1]));
        break;
      }
    }
  }
  // the last weight is implied: the sum of 2^(w-1) completes a power of two
  uint32_t sum = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (weights[i] > kHufMaxBits) return 0;
    if (weights[i]) sum += 1u << (weights[i] - 1);
  }
  if (sum == 0) return 0;
  const uint32_t max_bits = HighBit(sum) + 1;
  if (max_bits > kHufMaxBits) return 0;
  const uint32_t rest = (1u << max_bits) - sum;
  if (rest & (rest - 1)) return 0;   // not a power of two
  weights[n++] = static_cast<uint8_t>(HighBit(rest) + 1);
  *n_out = n;
  *max_bits_out = max_bits;
  return used;
}
// cells {symbol | nbits << 8} in the order of ascending weight (= descending code length), symbols of one weight in symbol
// order; a symbol of weight w owns 2^(w-1) consecutive cells
template <typename HP, typename WP>
MI_ZHD bool FillHuffmanTable(WP weights, uint32_t n, uint32_t max_bits, HP table) {
  uint32_t at = 0;
  for (uint32_t w = 1; w <= max_bits; w++) {
    const uint32_t len = 1u << (w - 1);
    const uint16_t tag = static_cast<uint16_t>((max_bits + 1 - w) << 8);
    for (uint32_t s = 0; s < n; s++) {
      if (weights[s] != w) continue;
      for (uint32_t i = 0; i < len; i++) table[at + i] = static_cast<uint16_t>(tag | s);
      at += len;
    }
  }
  return at == (1u << max_bits);
}
template <typename BP, typename HP, typename WP, typename TP, typename CP, typename NP>
MI_ZHD uint32_t ReadHuffmanTable(BP src, uint32_t avail, HP table, uint32_t* max_bits_out, WP weights, TP cells, CP counts, NP next) {
  uint32_t n = 0, max_bits = 0;
  const uint32_t used = ReadHuffmanWeights(src, avail, &n, &max_bits, weights, cells, counts, next);
  if (!used || !FillHuffmanTable(weights, n, max_bits, table)) return 0;
  *max_bits_out = max_bits;
  return used;
}

// Stream s (0..3, or 0 of 1) of a Huffman-coded literals section: its bytes (from the block's first byte), the literals it
// decodes and where they go.  desc_bytes = size of the tree description in front of the streams (0: treeless).
template <typename BP>
MI_ZHD bool LiteralStream(const BlockInfo& z, BP block, uint32_t desc_bytes, uint32_t s, uint32_t* first, uint32_t* nbytes,
                          uint32_t* out0, uint32_t* nsym) {
  if (desc_bytes >= z.lit_comp) return false;
  const uint32_t at = z.lit_hdr + desc_bytes, total = z.lit_comp - desc_bytes;
  if (z.lit_streams == 1) {
    *first = at;
    *nbytes = total;
    *out0 = 0;
    *nsym = z.lit_regen;
    return true;
  }
  if (total < 10) return false;   // jump table + one byte per stream
  const uint32_t n1 = block[at] | (static_cast<uint32_t>(block[at + 1]) << 8), n2 = block[at + 2] | (static_cast<uint32_t>(block[at + 3]) << 8),
                 n3 = block[at + 4] | (static_cast<uint32_t>(block[at + 5]) << 8);
  if (6 + n1 + n2 + n3 >= total) return false;
  const uint32_t per = (z.lit_regen + 3) / 4;
  if (3 * per > z.lit_regen) return false;
  *out0 = s * per;
  *nsym = s < 3 ? per : z.lit_regen - 3 * per;
  *first = at + 6 + (s > 0 ? n1 : 0) + (s > 1 ? n2 : 0) + (s > 2 ? n3 : 0);
  *nbytes = s == 0 ? n1 : s == 1 ? n2 : s == 2 ? n3 : total - 6 - n1 - n2 - n3;
  return true;
}

// One Huffman-coded stream -> nsym literals.  false = the stream does not end where its symbols do.  `br` is the caller's
// reader (a WindowWords reader has its window set).  Four literals leave as one aligned 32-bit store.
template <typename READER, typename BP, typename HP, typename OP>
MI_ZHD bool DecodeHuffmanStream(READER& br, BP first, uint32_t nbytes, uint32_t nsym, HP table, uint32_t max_bits, OP out) {
  if (!br.Open(first, nbytes)) return false;
  const uint32_t addr = static_cast<uint32_t>(Mem<OP>::Address(out));
  uint32_t acc = 0, fill = 0;
  for (uint32_t i = 0; i < nsym; i++) {
    const uint32_t c = table[br.Peek(max_bits)];
    br.Skip(c >> 8);
    const uint32_t sym = c & 0xFFu;
    if (fill == 0 && (((addr + i) & 3u) != 0 || i + 4 > nsym)) {
      out[i] = static_cast<uint8_t>(sym);
    } else {
      acc |= sym << (8 * fill);
      if (++fill == 4) {
        Mem<OP>::Store32(out + (i - 3), acc);
        acc = 0;
        fill = 0;
      }
    }
  }
  return br.Left() == 0;
}

// The sequences of one block.  emit(i, literal_length, match_length, offset_or_marker); offsets that name a repeat offset
// leave as kRepMarker | (value - 1 + (literal_length == 0)): the history runs across the blocks of a frame, so they are
// resolved later (RepStep / RepResolve below).
template <typename READER, typename BP, typename TP, typename EMIT>
MI_ZHD bool DecodeSequences(READER& br, BP bits, uint32_t nbytes, uint32_t nseq, TP tll, uint32_t al_ll, TP tof, uint32_t al_of, TP tml,
                            uint32_t al_ml, EMIT emit) {
  if (!br.Open(bits, nbytes)) return false;
  uint32_t sll = br.Read(al_ll), sof = br.Read(al_of), sml = br.Read(al_ml);
  for (uint32_t i = 0; i < nseq; i++) {
    const FseCell cl = tll[sll], co = tof[sof], cm = tml[sml];
    const uint32_t ov = CellBase(co) + br.Read(CellExtra(co));
    const uint32_t ml = CellBase(cm) + br.Read(CellExtra(cm));
    const uint32_t ll = CellBase(cl) + br.Read(CellExtra(cl));
    if (i + 1 < nseq) {
      sll = CellNext(cl) + br.Read(CellBits(cl));
      sml = CellNext(cm) + br.Read(CellBits(cm));
      sof = CellNext(co) + br.Read(CellBits(co));
    }
    if (br.Left() < 0) return false;
    // bit 31 of the emitted word means "repeat offset": an offset code of 31 (only a damaged frame has one; offsets reach
    // 2^31 + 2^31 - 1 with it) must not be mistaken for one
    if (ov > 3 && ov - 3 >= kRepMarker) return false;
    if (!emit(i, ll, ml, ov > 3 ? ov - 3 : (kRepMarker | (ov - 1 + (ll == 0 ? 1u : 0u))))) return false;
  }
  return br.Left() == 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same two decoders over a stream that is wholly addressable by 32-bit word (the block staged in LDS on the device, a
// plain array on the CPU): POSITIONAL reads instead of a shifting bit buffer.
//
// BackBits above pays for every field it reads: peek, shift the 64-bit buffer, two counters, a refill test -- some fifteen
// dependent instructions, six fields per sequence, on ONE lane, where every instruction costs a full issue slot.  With the
// stream addressable the head is just a bit position.  A sequence's six field widths are all known as soon as its three
// table cells are (the cell holds the width of the code's extra bits and of the next state's bits), so the six positions are a
// prefix sum, the twelve words they lie in are independent loads, and a field is one v_alignbit and a mask: the dependent
// chain per sequence is two table / stream round trips and a handful of instructions.  The Huffman streams take four
// symbols from one 64-bit fetch (at most 11 bits each), one table look-up per symbol.
//
// WP: indexable, WP[i] = aligned word i of the stream counted from the 4-byte boundary at or before its first byte; words
// up to one past the stream's last byte are read.  `mis` = byte offset of the stream's first byte in word 0.
// ... or through a window of W words of fast memory that the reader slides itself (the device: 1 KiB of LDS per stream; a block
// staged whole would cost the LDS that lets eight blocks share a CU, and the kernel lives on blocks side by side).  Indices only
// go down, a few words at a time; PosEnsure(w, lo, hi) makes words [lo, hi] addressable before they are read.
template <typename BP, typename LP, int W>
struct SlidingWords {
  BP base;      // the 4-byte boundary at or before the stream's first byte
  LP win;
  int32_t lo;   // the window holds the words lo .. lo + W - 1
  MI_ZHD void Init(BP b, LP l) {
    base = b;
    win = l;
    lo = 0x3FFFFFFF;   // nothing yet
  }
  MI_ZHD void Ensure(int32_t need_lo, int32_t need_hi) {   // need_hi - need_lo < W
    if (need_lo < lo || need_hi >= lo + W) {
      lo = need_hi + 1 - W;
      for (int j = 0; j < W; j++) win[j] = lo + j >= 0 ? Mem<BP>::Load32(base + 4 * (lo + j)) : 0u;
    }
  }
  MI_ZHD uint32_t operator[](uint32_t i) const { return win[static_cast<int32_t>(i) - lo]; }
};
template <typename WP>
MI_ZHD void PosEnsure(WP&, int32_t, int32_t) {}   // a plain array of words: everything is addressable
template <typename BP, typename LP, int W>
MI_ZHD void PosEnsure(SlidingWords<BP, LP, W>& w, int32_t lo, int32_t hi) { w.Ensure(lo, hi); }

MI_ZHD uint32_t AlignBit(uint32_t hi, uint32_t lo, uint32_t sh) {   // (hi:lo) >> sh, low 32 bits; sh < 32
  return static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> sh);
}
template <typename WP>
MI_ZHD uint32_t PosField(const WP& w, int32_t at, uint32_t n) {   // bits [at, at + n) of the stream, n <= 32, at >= 0
  const uint32_t i = static_cast<uint32_t>(at) >> 5, sh = static_cast<uint32_t>(at) & 31u;
  const uint32_t v = AlignBit(w[i + 1], w[i], sh);
  return n >= 32 ? v : (v & ((1u << n) - 1u));
}
// head position (in bits from the aligned base) of a backward stream of nbytes bytes that starts `mis` bytes into word 0;
// -1: the stream is empty or its last byte is zero (no end mark)
template <typename WP>
MI_ZHD int32_t PosOpen(WP& w, uint32_t mis, uint32_t nbytes) {
  if (nbytes == 0) return -1;
  const uint32_t last_at = mis + nbytes - 1;
  PosEnsure(w, static_cast<int32_t>(last_at >> 2), static_cast<int32_t>(last_at >> 2) + 1);
  const uint32_t last = (w[last_at >> 2] >> (8 * (last_at & 3u))) & 0xFFu;
  if (last == 0) return -1;
  return static_cast<int32_t>(8 * last_at + HighBit(last));
}

// The sequence decoder as a resumable state: Open, then Step once per sequence (the device decodes 64 at a time on one lane
// and lets the whole wave do everything else about them).
template <typename WP, typename TP>
struct SeqPosDecoder {
  WP w;
  TP tll, tof, tml;
  int32_t q, floor;
  uint32_t sll, sof, sml;
  MI_ZHD bool Open(WP words, uint32_t mis, uint32_t nbytes, TP ll, uint32_t al_ll, TP of, uint32_t al_of, TP ml, uint32_t al_ml) {
    w = words;
    tll = ll;
    tof = of;
    tml = ml;
    q = PosOpen(w, mis, nbytes);
    floor = static_cast<int32_t>(8 * mis);   // bits below belong to whatever lies in front of the stream
    if (q < 0 || q - static_cast<int32_t>(al_ll + al_of + al_ml) < floor) return false;
    PosEnsure(w, (q - static_cast<int32_t>(al_ll + al_of + al_ml)) >> 5, (q >> 5) + 1);
    q -= static_cast<int32_t>(al_ll);
    sll = PosField(w, q, al_ll);
    q -= static_cast<int32_t>(al_of);
    sof = PosField(w, q, al_of);
    q -= static_cast<int32_t>(al_ml);
    sml = PosField(w, q, al_ml);
    return true;
  }
  // one sequence: literal length, match length, offset code as DecodeSequences emits it.  `more`: another sequence follows.
  MI_ZHD bool Step(bool more, uint32_t* ll_out, uint32_t* ml_out, uint32_t* code_out) {
    const FseCell cl = tll[sll], co = tof[sof], cm = tml[sml];
    const uint32_t e_of = CellExtra(co), e_ml = CellExtra(cm), e_ll = CellExtra(cl);
    const uint32_t b_ll = more ? CellBits(cl) : 0u, b_ml = more ? CellBits(cm) : 0u, b_of = more ? CellBits(co) : 0u;
    // the order of the fields below the head: offset extra, match-length extra, literal-length extra, then the bits of the
    // next literal-length, match-length and offset states
    const int32_t p1 = q - static_cast<int32_t>(e_of), p2 = p1 - static_cast<int32_t>(e_ml), p3 = p2 - static_cast<int32_t>(e_ll),
                  p4 = p3 - static_cast<int32_t>(b_ll), p5 = p4 - static_cast<int32_t>(b_ml), p6 = p5 - static_cast<int32_t>(b_of);
    if (p6 < floor) return false;   // the stream ran out
    PosEnsure(w, p6 >> 5, (q >> 5) + 1);
    const uint32_t ov = CellBase(co) + PosField(w, p1, e_of);
    *ml_out = CellBase(cm) + PosField(w, p2, e_ml);
    const uint32_t ll = CellBase(cl) + PosField(w, p3, e_ll);
    *ll_out = ll;
    if (more) {
      sll = CellNext(cl) + PosField(w, p4, b_ll);
      sml = CellNext(cm) + PosField(w, p5, b_ml);
      sof = CellNext(co) + PosField(w, p6, b_of);
    }
    q = p6;
    if (ov > 3 && ov - 3 >= kRepMarker) return false;   // offset code 31: see DecodeSequences
    *code_out = ov > 3 ? ov - 3 : (kRepMarker | (ov - 1 + (ll == 0 ? 1u : 0u)));
    return true;
  }
  MI_ZHD bool AtEnd() const { return q == floor; }
};

template <typename WP, typename TP, typename EMIT>
MI_ZHD bool DecodeSequencesPos(WP w, uint32_t mis, uint32_t nbytes, uint32_t nseq, TP tll, uint32_t al_ll, TP tof, uint32_t al_of, TP tml,
                               uint32_t al_ml, EMIT emit) {
  SeqPosDecoder<WP, TP> dec;
  if (!dec.Open(w, mis, nbytes, tll, al_ll, tof, al_of, tml, al_ml)) return false;
  for (uint32_t i = 0; i < nseq; i++) {
    uint32_t ll, ml, code;
    if (!dec.Step(i + 1 < nseq, &ll, &ml, &code)) return false;
    if (!emit(i, ll, ml, code)) return false;
  }
  return dec.AtEnd();
}

template <typename WP, typename HP, typename OP>
MI_ZHD bool DecodeHuffmanStreamPos(WP w, uint32_t mis, uint32_t nbytes, uint32_t nsym, HP table, uint32_t max_bits, OP out) {
  int32_t q = PosOpen(w, mis, nbytes);
  const int32_t floor = static_cast<int32_t>(8 * mis);
  if (q < 0) return false;
  const uint32_t addr = static_cast<uint32_t>(Mem<OP>::Address(out));
  uint32_t i = 0;
  // symbols in front of the first aligned output word, one at a time (a look-up may peek below the stream's first bit: the
  // format reads zeros there -- the words in front of the stream hold other bytes of the block, so they are masked)
  auto peek = [&](int32_t head) -> uint32_t {
    const int32_t at = head - static_cast<int32_t>(max_bits);
    PosEnsure(w, (at > floor ? at : floor) >> 5, (head >> 5) + 1);
    if (at >= floor) return PosField(w, at, max_bits);
    const int32_t have = head - floor;   // < max_bits bits are left
    return have <= 0 ? 0u : (PosField(w, floor, static_cast<uint32_t>(have)) << (max_bits - static_cast<uint32_t>(have)));
  };
  for (; i < nsym && ((addr + i) & 3u) != 0; i++) {
    const uint32_t c = table[peek(q)];
    q -= static_cast<int32_t>(c >> 8);
    out[i] = static_cast<uint8_t>(c);
  }
  // four symbols per round from ONE 64-bit fetch: 4 x max_bits <= 44 bits lie in the two words below the head's word and
  // that word itself -- three words, fetched together
  while (i + 4 <= nsym && q - floor >= 64) {
    const uint32_t hi_idx = static_cast<uint32_t>(q - 1) >> 5;          // word of the first unread bit
    const uint32_t sh = 31u - (static_cast<uint32_t>(q - 1) & 31u);      // unread bits above it in that word: none after the shift
    PosEnsure(w, static_cast<int32_t>(hi_idx) - 2 < 0 ? 0 : static_cast<int32_t>(hi_idx) - 2, static_cast<int32_t>(hi_idx));
    const uint32_t w2 = w[hi_idx], w1 = w[hi_idx - 1], w0 = sh ? w[hi_idx - 2] : 0u;
    // the 64 bits below the head, left-aligned (bit 63 = the next bit to read)
    uint64_t buf = ((static_cast<uint64_t>(w2) << 32) | w1) << sh;
    buf |= sh ? (static_cast<uint64_t>(w0) >> (32u - sh)) : 0ull;
    uint32_t acc = 0, used = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      const uint32_t c = table[static_cast<uint32_t>(buf >> (64u - max_bits))];
      const uint32_t nb = c >> 8;
      buf <<= nb;
      used += nb;
      acc |= (c & 0xFFu) << (8 * k);
    }
    q -= static_cast<int32_t>(used);
    Mem<OP>::Store32(out + i, acc);
    i += 4;
  }
  for (; i < nsym; i++) {   // the tail, and the last bits of the stream
    const uint32_t c = table[peek(q)];
    q -= static_cast<int32_t>(c >> 8);
    out[i] = static_cast<uint8_t>(c);
  }
  return q == floor;
}

// Repeat offsets.  A sequence either brings a fresh offset or names one of the frame's three most recent ones, and that
// history runs through all blocks of a frame -- but a block is decoded without its predecessors.  So the history is kept
// SYMBOLICALLY: a state word is either a known offset (bit 31 clear; 0 = invalid) or "slot i of the history at the start of
// this slice of sequences, minus d" (kRepSym | i << 29 | d).  The decoding lane runs the state machine on such words; what it
// stores per sequence is the offset itself where that is known, else the symbolic word; what it stores per slice is the state
// at the slice's end as a function of the state at its start.  Functions compose (RepResolve slot by slot), so a prefix scan
// over the slices of a frame gives every slice its true starting state, and a symbolic offset resolves in one step.
constexpr uint32_t kRepSym = 0x80000000u;
MI_ZHD uint32_t RepSlot(uint32_t i) { return kRepSym | (i << 29); }
// the value of word `w` once the state it refers to is (f0, f1, f2) -- themselves known or symbolic
MI_ZHD uint32_t RepResolve(uint32_t w, uint32_t f0, uint32_t f1, uint32_t f2) {
  if (!(w >> 31)) return w;
  const uint32_t i = (w >> 29) & 3u, dec = w & 0x1FFFFFFFu;
  const uint32_t u = i == 0 ? f0 : i == 1 ? f1 : f2;
  if (!(u >> 31)) return u > dec ? u - dec : 0u;
  return u + dec;
}
// The same step as a FUNCTION on the history: F = the state after the sequence when the state before it is (slot 0, slot 1,
// slot 2) -- RepStep on the identity -- so that the states of many sequences come from a parallel scan: later after earlier is
// (RepResolve(later.x, earlier), RepResolve(later.y, earlier), RepResolve(later.z, earlier)).  *used = the offset the sequence
// uses, in terms of the state before it.
MI_ZHD uint32_t RepStep(uint32_t code, uint32_t* S);
MI_ZHD void RepFunction(uint32_t code, bool has_match, uint32_t* fx, uint32_t* fy, uint32_t* fz, uint32_t* used) {
  uint32_t S[3] = {RepSlot(0), RepSlot(1), RepSlot(2)};
  *used = has_match ? RepStep(code, S) : 0u;
  *fx = S[0];
  *fy = S[1];
  *fz = S[2];
}
// one sequence: `code` as DecodeSequences emits it; S[3] is updated, the offset the sequence uses is returned (0 = invalid)
MI_ZHD uint32_t RepStep(uint32_t code, uint32_t* S) {
  const uint32_t s0 = S[0], s1 = S[1], s2 = S[2];   // constant indices only: the three words stay in registers
  if (!(code & kRepMarker)) {
    S[0] = code;
    S[1] = s0;
    S[2] = s1;
    return code;
  }
  const uint32_t idx = code & 3u;
  if (idx == 0) return s0;
  if (idx == 1) {          // the second most recent moves to the front
    S[0] = s1;
    S[1] = s0;
    return s1;
  }
  // the third most recent (idx 2), or the most recent minus one (idx 3), becomes the most recent
  const uint32_t used = idx == 2 ? s2 : (s0 >> 31) ? s0 + 1u : (s0 > 1u ? s0 - 1u : 0u);
  S[0] = used;
  S[1] = s0;
  S[2] = s1;
  return used;
}

}  // namespace zstd
}  // namespace miarrow
