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

// One cell of an FSE decoding table: the state IS the index of the cell.
struct FseCell {
  uint32_t base;       // sequences: base value of the code; weights: the symbol
  uint16_t next;       // next state = next + the `nbits` bits read
  uint8_t nbits;
  uint8_t extra;       // sequences: additional bits of the code's value
};

// ---------------------------------------------------------------------------------------------------------------------
// Backward bitstream: the last byte's highest set bit ends the stream, bits are taken from just below it towards the first
// byte.  The stream is consumed strictly front to back (in its own direction), so the 8-byte words are loaded ahead of their
// use (two words: a load that the next symbol waits for would put HBM / L2 latency into every step of a serial chain).
// Words are ALIGNED loads: positions are counted from the 8-byte boundary at or before the first byte; what lies in front of
// the stream reads as zero -- the format's rule for a stream that runs out (the caller sees Left() < 0).
struct BackBits {
  const uint64_t* words;   // aligned base
  int64_t p;               // position of the read head (bits below it are unread), counted from words[0] bit 0
  int64_t wbase;           // position of lo's bit 0
  uint32_t bias;           // position of the stream's first bit
  uint64_t lo, hi, n1, n2; // words at wbase, wbase + 64, wbase - 64, wbase - 128

  MI_ZHD uint64_t Word(int64_t idx) const {
    if (idx < 0) return 0;
    uint64_t w = words[idx];
    if (idx == 0 && bias) w &= ~uint64_t(0) << bias;
    return w;
  }
  // false: the stream is empty or its last byte is zero (no end mark)
  MI_ZHD bool Open(const uint8_t* first, uint32_t nbytes) {
    if (nbytes == 0) return false;
    const uint8_t last = first[nbytes - 1];
    if (last == 0) return false;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(first);
    words = reinterpret_cast<const uint64_t*>(addr & ~uintptr_t(7));
    bias = static_cast<uint32_t>(addr & 7u) * 8u;
    p = static_cast<int64_t>(bias) + 8 * static_cast<int64_t>(nbytes - 1) + HighBit(last);
    const int64_t top = (p - 1) >> 6;   // word of the first bit to read (p >= bias, so top >= -1 only when the stream holds no bit)
    wbase = (top - 1) * 64;
    hi = Word(top);
    lo = Word(top - 1);
    n1 = Word(top - 2);
    n2 = Word(top - 3);
    return true;
  }
  MI_ZHD int64_t Left() const { return p - static_cast<int64_t>(bias); }
  MI_ZHD uint64_t Window(uint32_t n) const {   // the n bits below the read head, n <= 57
    const uint32_t off = static_cast<uint32_t>(p - static_cast<int64_t>(n) - wbase);   // 0 .. 127
    uint64_t v;
    if (off >= 64) v = hi >> (off - 64);
    else if (off == 0) v = lo;
    else v = (lo >> off) | (hi << (64 - off));
    return v & ((uint64_t(1) << n) - 1);
  }
  MI_ZHD void Skip(uint32_t n) {
    p -= n;
    if (p <= wbase + 64) {   // the head left the upper word
      hi = lo;
      lo = n1;
      n1 = n2;
      wbase -= 64;
      n2 = Word((wbase >> 6) - 2);
    }
  }
  MI_ZHD uint32_t Peek(uint32_t n) const { return n ? static_cast<uint32_t>(Window(n)) : 0u; }   // n <= 32
  MI_ZHD uint32_t Read(uint32_t n) {
    if (n == 0) return 0;
    const uint32_t v = static_cast<uint32_t>(Window(n));
    Skip(n);
    return v;
  }
};

// Forward bits of a table description (a few dozen bytes): byte loads, no state worth keeping.
struct FwdBits {
  const uint8_t* src;
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
MI_ZHD uint32_t ReadNCount(const uint8_t* src, uint32_t avail, int type, int16_t* counts, uint32_t* log_out, uint32_t* nsym_out) {
  if (avail == 0) return 0;
  const uint32_t max_sym = MaxSym(type);
  FwdBits f{src, 0};
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

// Normalized counts -> decoding table of 1 << al cells.  `next` is scratch for one uint16 per symbol.
MI_ZHD bool BuildFseTable(const int16_t* counts, uint32_t nsym, uint32_t al, int type, FseCell* table, uint16_t* next) {
  const uint32_t size = 1u << al, mask = size - 1;
  uint32_t high = size - 1;
  for (uint32_t s = 0; s < nsym; s++) {
    if (counts[s] == -1) {
      table[high--].base = s;
      next[s] = 1;
    } else {
      next[s] = static_cast<uint16_t>(counts[s]);
    }
  }
  const uint32_t step = (size >> 1) + (size >> 3) + 3;
  uint32_t pos = 0;
  for (uint32_t s = 0; s < nsym; s++) {
    for (int32_t i = 0; i < counts[s]; i++) {
      table[pos].base = s;
      do pos = (pos + step) & mask; while (pos > high);
    }
  }
  if (pos != 0) return false;
  for (uint32_t u = 0; u < size; u++) {
    const uint32_t s = table[u].base;
    const uint32_t ns = next[s]++;
    const uint32_t nbits = al - HighBit(ns);
    FseCell c;
    c.nbits = static_cast<uint8_t>(nbits);
    c.next = static_cast<uint16_t>((ns << nbits) - size);
    if (type == kLL) { c.base = LlBase(s); c.extra = static_cast<uint8_t>(LlBits(s)); }
    else if (type == kML) { c.base = MlBase(s); c.extra = static_cast<uint8_t>(MlBits(s)); }
    else if (type == kOF) { c.base = 1u << s; c.extra = static_cast<uint8_t>(s); }
    else { c.base = s; c.extra = 0; }
    table[u] = c;
  }
  return true;
}
MI_ZHD void BuildRleTable(uint32_t s, int type, FseCell* table) {
  FseCell c;
  c.nbits = 0;
  c.next = 0;
  if (type == kLL) { c.base = LlBase(s); c.extra = static_cast<uint8_t>(LlBits(s)); }
  else if (type == kML) { c.base = MlBase(s); c.extra = static_cast<uint8_t>(MlBits(s)); }
  else { c.base = 1u << s; c.extra = static_cast<uint8_t>(s); }
  table[0] = c;
}

// One of the three sequence tables of a block from the block's own bytes.  `seq` = the sequences section behind its count
// (at the modes byte), `avail` = bytes from there to the end of the block; mode 3 (repeat) is resolved by the caller (it
// passes the earlier block the table comes from).  Returns the accuracy log, ~0u = malformed.  counts/next: scratch
// (53 entries are enough).
MI_ZHD uint32_t BuildSequenceTable(const uint8_t* seq, uint32_t avail, int type, FseCell* table, int16_t* counts, uint16_t* next) {
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
      return BuildFseTable(counts, ns, al, type, table, next) ? al : ~0u;
    }
    return ~0u;   // repeat of a repeat: the host walk resolves chains, so this is a malformed frame
  }
  return ~0u;
}
// Where the bitstream of the sequences section begins, counted from the modes byte (0 = malformed).
MI_ZHD uint32_t SequenceBitstreamOffset(const uint8_t* seq, uint32_t avail, int16_t* counts) {
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
MI_ZHD uint32_t ReadHuffmanTable(const uint8_t* src, uint32_t avail, uint16_t* table, uint32_t* max_bits_out, uint8_t* weights, FseCell* cells,
                                 int16_t* counts, uint16_t* next) {
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
    BackBits br;
    if (!br.Open(src + 1 + hdr, hb - hdr)) return 0;
    uint32_t s1 = br.Read(al), s2 = br.Read(al);
    if (br.Left() < 0) return 0;
    for (;;) {
      if (n > 253) return 0;
      FseCell c = cells[s1];
      weights[n++] = static_cast<uint8_t>(c.base);
      s1 = c.next + br.Read(c.nbits);
      if (br.Left() < 0) {
        weights[n++] = static_cast<uint8_t>(cells[s2].base);
        break;
      }
      if (n > 253) return 0;
      c = cells[s2];
      weights[n++] = static_cast<uint8_t>(c.base);
      s2 = c.next + br.Read(c.nbits);
      if (br.Left() < 0) {
        weights[n++] = static_cast<uint8_t>(cells[s1].base);
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
  // cells in the order of ascending weight (= descending code length), symbols of one weight in symbol order
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
  if (at != (1u << max_bits)) return 0;
  *max_bits_out = max_bits;
  return used;
}

// Stream s (0..3, or 0 of 1) of a Huffman-coded literals section: its bytes (from the block's first byte), the literals it
// decodes and where they go.  desc_bytes = size of the tree description in front of the streams (0: treeless).
MI_ZHD bool LiteralStream(const BlockInfo& z, const uint8_t* block, uint32_t desc_bytes, uint32_t s, uint32_t* first, uint32_t* nbytes,
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

// One Huffman-coded stream -> nsym literals.  false = the stream does not end where its symbols do.
template <typename OUT>
MI_ZHD bool DecodeHuffmanStream(const uint8_t* first, uint32_t nbytes, uint32_t nsym, const uint16_t* table, uint32_t max_bits, OUT out) {
  BackBits br;
  if (!br.Open(first, nbytes)) return false;
  for (uint32_t i = 0; i < nsym; i++) {
    const uint32_t c = table[br.Peek(max_bits)];
    out[i] = static_cast<uint8_t>(c);
    br.Skip(c >> 8);
  }
  return br.Left() == 0;
}

// The sequences of one block.  emit(i, literal_length, match_length, offset_or_marker); offsets that name a repeat offset
// leave as kRepMarker | (value - 1 + (literal_length == 0)): the history runs across the blocks of a frame, so they are
// resolved by whoever walks the frame's blocks in order (ResolveRepeat below).
template <typename EMIT>
MI_ZHD bool DecodeSequences(const uint8_t* bits, uint32_t nbytes, uint32_t nseq, const FseCell* tll, uint32_t al_ll, const FseCell* tof,
                            uint32_t al_of, const FseCell* tml, uint32_t al_ml, EMIT emit) {
  BackBits br;
  if (!br.Open(bits, nbytes)) return false;
  uint32_t sll = br.Read(al_ll), sof = br.Read(al_of), sml = br.Read(al_ml);
  for (uint32_t i = 0; i < nseq; i++) {
    const FseCell cl = tll[sll], co = tof[sof], cm = tml[sml];
    const uint32_t ov = co.base + br.Read(co.extra);
    const uint32_t ml = cm.base + br.Read(cm.extra);
    const uint32_t ll = cl.base + br.Read(cl.extra);
    if (i + 1 < nseq) {
      sll = cl.next + br.Read(cl.nbits);
      sml = cm.next + br.Read(cm.nbits);
      sof = co.next + br.Read(co.nbits);
    }
    if (br.Left() < 0) return false;
    if (!emit(i, ll, ml, ov > 3 ? ov - 3 : (kRepMarker | (ov - 1 + (ll == 0 ? 1u : 0u))))) return false;
  }
  return br.Left() == 0;
}

// rep[3] = the frame's repeat offsets (1, 4, 8 at its start).  Returns the offset the sequence uses; 0 = malformed.
MI_ZHD uint32_t ResolveRepeat(uint32_t offset_or_marker, uint32_t* rep) {
  uint32_t off;
  if (!(offset_or_marker & kRepMarker)) {
    off = offset_or_marker;
    rep[2] = rep[1];
    rep[1] = rep[0];
    rep[0] = off;
    return off;
  }
  const uint32_t idx = offset_or_marker & 3u;
  if (idx == 0) return rep[0];
  off = idx == 3 ? rep[0] - 1 : rep[idx];
  if (off == 0) return 0;
  if (idx != 1) rep[2] = rep[1];
  rep[1] = rep[0];
  rep[0] = off;
  return off;
}

}  // namespace zstd
}  // namespace miarrow
