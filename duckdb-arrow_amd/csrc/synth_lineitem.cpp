// synth_lineitem.cpp -- seeded TPC-H-shaped lineitem generator (bench / test support, see include/mi_synth.h).
// Uses the product's own IPC metadata encoder (ipc_format.cpp), so every benchmark input also exercises it.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mi_synth.h"
#include "ipc_format.hpp"

namespace miarrow {
int WrapC(const std::function<void()>& f);
}
using namespace miarrow;

namespace {

inline uint64_t Mix(uint64_t x) {  // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
inline uint64_t Rnd(uint64_t seed, uint32_t stream, uint64_t row) { return Mix(seed ^ Mix(row * 0x100000001B3ull + stream)); }
inline uint64_t Uniform(uint64_t r, uint64_t lo, uint64_t hi) { return lo + r % (hi - lo + 1); }

const char* kInstruct[4] = {"DELIVER IN PERSON", "COLLECT COD", "NONE", "TAKE BACK RETURN"};
const char* kMode[7] = {"REG AIR", "AIR", "RAIL", "SHIP", "TRUCK", "MAIL", "FOB"};
const char* kWords[] = {"furiously", "quickly", "carefully", "blithely", "slyly", "regular", "express", "pending", "final", "ironic",
                        "deposits", "requests", "accounts", "packages", "foxes", "ideas", "theodolites", "pinto", "beans", "sleep",
                        "above", "the", "according", "to", "even", "bold", "special", "silent", "unusual", "dependencies"};
constexpr int kNumWords = sizeof(kWords) / sizeof(kWords[0]);

enum Stream : uint32_t { S_PART = 1, S_SUPP, S_QTY, S_DISC, S_TAX, S_ORDERDATE, S_SHIP, S_COMMIT, S_RECEIPT, S_FLAG, S_INSTR, S_MODE, S_CLEN, S_CTEXT };

struct Opts {
  double sf;
  uint64_t seed;
  int64_t rows_per_batch, n_rows, n_batches, first_row;
  bool with_validity;
  int n_threads;
};

Opts Resolve(const mi_synth_options* o) {
  if (!o) throw InvalidInputException("mi_synth: NULL options");
  Opts r;
  r.sf = o->scale_factor > 0 ? o->scale_factor : 1.0;
  r.seed = o->seed;
  r.rows_per_batch = o->rows_per_batch > 0 ? o->rows_per_batch : 122880;
  if (o->n_rows > 0) r.n_rows = o->n_rows;
  else if (r.sf == 1.0) r.n_rows = 6001215;
  else if (r.sf == 10.0) r.n_rows = 59986052;
  else if (r.sf == 100.0) r.n_rows = 600037902;
  else r.n_rows = static_cast<int64_t>(6001215.0 * r.sf);
  r.n_batches = (r.n_rows + r.rows_per_batch - 1) / r.rows_per_batch;
  r.first_row = o->first_row;
  if (r.first_row < 0 || r.first_row % r.rows_per_batch != 0)
    throw InvalidInputException("mi_synth: first_row must be a non-negative multiple of rows_per_batch");
  r.with_validity = o->with_validity != 0;
  int hw = static_cast<int>(std::thread::hardware_concurrency());
  r.n_threads = o->n_threads > 0 ? o->n_threads : std::max(1, std::min(hw, 32));
  return r;
}

ArrowSchemaModel LineitemSchema() {
  const char* names[16] = {"l_orderkey", "l_partkey", "l_suppkey", "l_linenumber", "l_quantity", "l_extendedprice", "l_discount", "l_tax",
                           "l_returnflag", "l_linestatus", "l_shipdate", "l_commitdate", "l_receiptdate", "l_shipinstruct", "l_shipmode", "l_comment"};
  const char* types[16] = {"BIGINT", "BIGINT", "BIGINT", "BIGINT", "DECIMAL(15,2)", "DECIMAL(15,2)", "DECIMAL(15,2)", "DECIMAL(15,2)",
                           "VARCHAR", "VARCHAR", "DATE", "DATE", "DATE", "VARCHAR", "VARCHAR", "VARCHAR"};
  ArrowSchemaModel s;
  for (int i = 0; i < 16; i++) s.fields.push_back(FieldFromDuckType(names[i], types[i]));
  return s;
}

inline int CommentLen(const Opts& o, int64_t row) { return static_cast<int>(Uniform(Rnd(o.seed, S_CLEN, static_cast<uint64_t>(row)), 10, 43)); }
inline int InstrIdx(const Opts& o, int64_t row) { return static_cast<int>(Rnd(o.seed, S_INSTR, static_cast<uint64_t>(row)) % 4); }
inline int ModeIdx(const Opts& o, int64_t row) { return static_cast<int>(Rnd(o.seed, S_MODE, static_cast<uint64_t>(row)) % 7); }

struct BatchLayout {
  int64_t nrows;
  std::vector<mi_buffer_span> spans;  // in schema order
  int64_t body_size;
  int64_t str_bytes[3];               // shipinstruct, shipmode, comment
};

size_t Pad8(size_t v) { return (v + 7) & ~static_cast<size_t>(7); }

BatchLayout LayoutOf(const Opts& o, int64_t batch) {
  BatchLayout L;
  const int64_t local0 = batch * o.rows_per_batch;
  const int64_t row0 = o.first_row + local0;
  L.nrows = std::min(o.rows_per_batch, o.n_rows - local0);
  int64_t instr = 0, mode = 0, comment = 0;
  static const int instr_len[4] = {17, 11, 4, 16};
  static const int mode_len[7] = {7, 3, 4, 4, 5, 4, 3};
  for (int64_t r = row0; r < row0 + L.nrows; r++) {
    instr += instr_len[InstrIdx(o, r)];
    mode += mode_len[ModeIdx(o, r)];
    comment += CommentLen(o, r);
  }
  L.str_bytes[0] = instr;
  L.str_bytes[1] = mode;
  L.str_bytes[2] = comment;
  size_t off = 0;
  const int64_t n = L.nrows;
  auto add = [&](int64_t len) {
    L.spans.push_back(mi_buffer_span{static_cast<int64_t>(off), len});
    off += Pad8(static_cast<size_t>(len));
  };
  auto validity = [&]() { add(o.with_validity ? (n + 7) / 8 : 0); };
  for (int c = 0; c < 4; c++) { validity(); add(n * 8); }
  for (int c = 0; c < 4; c++) { validity(); add(n * 16); }
  for (int c = 0; c < 2; c++) { validity(); add((n + 1) * 4); add(n); }       // returnflag, linestatus: 1 byte each
  for (int c = 0; c < 3; c++) { validity(); add(n * 4); }
  for (int c = 0; c < 3; c++) { validity(); add((n + 1) * 4); add(L.str_bytes[c]); }
  L.body_size = static_cast<int64_t>(off);
  return L;
}

std::vector<uint8_t> HeaderOf(const BatchLayout& L) {
  std::vector<std::pair<int64_t, int64_t>> nodes(16, {L.nrows, 0});
  return EncodeRecordBatchMessage(L.nrows, nodes, L.spans, L.body_size);
}

void FillBatch(const Opts& o, int64_t batch, const BatchLayout& L, uint8_t* body) {
  const int64_t row0 = o.first_row + batch * o.rows_per_batch;
  const int64_t n = L.nrows;
  std::memset(body, 0, static_cast<size_t>(L.body_size));
  size_t si = 0;
  auto next_validity = [&]() {
    const mi_buffer_span& s = L.spans[si++];
    if (s.length) {
      std::memset(body + s.offset, 0xFF, static_cast<size_t>(s.length));
      if (n & 7) body[s.offset + s.length - 1] = static_cast<uint8_t>((1u << (n & 7)) - 1u);  // Arrow writers zero the pad bits
    }
  };
  auto next = [&]() { return body + L.spans[si++].offset; };
  const uint64_t part_max = static_cast<uint64_t>(200000.0 * o.sf), supp_max = static_cast<uint64_t>(10000.0 * o.sf);
  // keys
  next_validity(); int64_t* orderkey = reinterpret_cast<int64_t*>(next());
  next_validity(); int64_t* partkey = reinterpret_cast<int64_t*>(next());
  next_validity(); int64_t* suppkey = reinterpret_cast<int64_t*>(next());
  next_validity(); int64_t* linenumber = reinterpret_cast<int64_t*>(next());
  next_validity(); int64_t* quantity = reinterpret_cast<int64_t*>(next());
  next_validity(); int64_t* extprice = reinterpret_cast<int64_t*>(next());
  next_validity(); int64_t* discount = reinterpret_cast<int64_t*>(next());
  next_validity(); int64_t* tax = reinterpret_cast<int64_t*>(next());
  next_validity(); int32_t* rf_off = reinterpret_cast<int32_t*>(next()); uint8_t* rf = next();
  next_validity(); int32_t* ls_off = reinterpret_cast<int32_t*>(next()); uint8_t* ls = next();
  next_validity(); int32_t* shipdate = reinterpret_cast<int32_t*>(next());
  next_validity(); int32_t* commitdate = reinterpret_cast<int32_t*>(next());
  next_validity(); int32_t* receiptdate = reinterpret_cast<int32_t*>(next());
  next_validity(); int32_t* si_off = reinterpret_cast<int32_t*>(next()); uint8_t* si_data = next();
  next_validity(); int32_t* sm_off = reinterpret_cast<int32_t*>(next()); uint8_t* sm_data = next();
  next_validity(); int32_t* cm_off = reinterpret_cast<int32_t*>(next()); uint8_t* cm_data = next();
  int32_t si_pos = 0, sm_pos = 0, cm_pos = 0;
  rf_off[0] = ls_off[0] = si_off[0] = sm_off[0] = cm_off[0] = 0;
  for (int64_t i = 0; i < n; i++) {
    const uint64_t row = static_cast<uint64_t>(row0 + i);
    const uint64_t order = row / 4;  // four lines per order, sparse TPC-H order keys
    orderkey[i] = static_cast<int64_t>((order / 8) * 32 + order % 8 + 1);
    linenumber[i] = static_cast<int64_t>(row % 4 + 1);
    const uint64_t pk = Uniform(Rnd(o.seed, S_PART, row), 1, std::max<uint64_t>(part_max, 1));
    partkey[i] = static_cast<int64_t>(pk);
    suppkey[i] = static_cast<int64_t>(Uniform(Rnd(o.seed, S_SUPP, row), 1, std::max<uint64_t>(supp_max, 1)));
    const int64_t qty = static_cast<int64_t>(Uniform(Rnd(o.seed, S_QTY, row), 1, 50));
    const int64_t retail = 90000 + static_cast<int64_t>((pk / 10) % 20001) + 100 * static_cast<int64_t>(pk % 1000);  // cents
    quantity[2 * i] = qty * 100;              quantity[2 * i + 1] = 0;   // decimal128 {lower, upper}
    extprice[2 * i] = qty * retail;           extprice[2 * i + 1] = 0;
    discount[2 * i] = static_cast<int64_t>(Uniform(Rnd(o.seed, S_DISC, row), 0, 10));  discount[2 * i + 1] = 0;
    tax[2 * i] = static_cast<int64_t>(Uniform(Rnd(o.seed, S_TAX, row), 0, 8));        tax[2 * i + 1] = 0;
    const int32_t orderdate = static_cast<int32_t>(Uniform(Rnd(o.seed, S_ORDERDATE, order), 8035, 10440));
    const int32_t ship = orderdate + static_cast<int32_t>(Uniform(Rnd(o.seed, S_SHIP, row), 1, 121));
    const int32_t receipt = ship + static_cast<int32_t>(Uniform(Rnd(o.seed, S_RECEIPT, row), 1, 30));
    shipdate[i] = ship;
    commitdate[i] = orderdate + static_cast<int32_t>(Uniform(Rnd(o.seed, S_COMMIT, row), 30, 90));
    receiptdate[i] = receipt;
    rf[i] = receipt <= 9298 ? ((Rnd(o.seed, S_FLAG, row) & 1) ? 'R' : 'A') : 'N';
    ls[i] = ship > 9298 ? 'O' : 'F';
    rf_off[i + 1] = static_cast<int32_t>(i + 1);
    ls_off[i + 1] = static_cast<int32_t>(i + 1);
    const char* a = kInstruct[InstrIdx(o, static_cast<int64_t>(row))];
    const size_t al = std::strlen(a);
    std::memcpy(si_data + si_pos, a, al);
    si_pos += static_cast<int32_t>(al);
    si_off[i + 1] = si_pos;
    const char* m = kMode[ModeIdx(o, static_cast<int64_t>(row))];
    const size_t ml = std::strlen(m);
    std::memcpy(sm_data + sm_pos, m, ml);
    sm_pos += static_cast<int32_t>(ml);
    sm_off[i + 1] = sm_pos;
    // comment: words separated by blanks, cut to the drawn length
    const int cl = CommentLen(o, static_cast<int64_t>(row));
    uint64_t h = Rnd(o.seed, S_CTEXT, row);
    int w = 0;
    while (w < cl) {
      const char* word = kWords[h % kNumWords];
      h = Mix(h);
      for (const char* p = word; *p && w < cl; p++) cm_data[cm_pos + w++] = static_cast<uint8_t>(*p);
      if (w < cl) cm_data[cm_pos + w++] = ' ';
    }
    cm_pos += cl;
    cm_off[i + 1] = cm_pos;
  }
}

void ParallelFor(int64_t n, int n_threads, const std::function<void(int64_t)>& fn) {
  std::atomic<int64_t> next{0};
  std::vector<std::thread> threads;
  std::atomic<bool> failed{false};
  std::string error;
  std::mutex* mu = new std::mutex();
  int nt = static_cast<int>(std::min<int64_t>(n_threads, std::max<int64_t>(n, 1)));
  for (int t = 0; t < nt; t++) {
    threads.emplace_back([&] {
      while (true) {
        int64_t i = next.fetch_add(1);
        if (i >= n || failed.load()) break;
        try {
          fn(i);
        } catch (const std::exception& e) {
          std::lock_guard<std::mutex> g(*mu);
          failed = true;
          error = e.what();
        }
      }
    });
  }
  for (auto& th : threads) th.join();
  delete mu;
  if (failed) throw InternalException("mi_synth: " + error);
}

}  // namespace

extern "C" {

int mi_synth_lineitem_layout(const mi_synth_options* o, int64_t* n_rows, int64_t* n_batches, int64_t* stream_size,
                             int64_t* batch_offsets, int64_t batch_offsets_cap) {
  return WrapC([&] {
    Opts opts = Resolve(o);
    std::vector<int64_t> sizes(static_cast<size_t>(opts.n_batches));
    ParallelFor(opts.n_batches, opts.n_threads, [&](int64_t b) {
      BatchLayout L = LayoutOf(opts, b);
      sizes[static_cast<size_t>(b)] = static_cast<int64_t>(HeaderOf(L).size()) + L.body_size;
    });
    int64_t pos = static_cast<int64_t>(EncodeSchemaMessage(LineitemSchema()).size());
    for (int64_t b = 0; b < opts.n_batches; b++) {
      if (batch_offsets && b < batch_offsets_cap) batch_offsets[b] = pos;
      pos += sizes[static_cast<size_t>(b)];
    }
    if (batch_offsets && opts.n_batches < batch_offsets_cap) batch_offsets[opts.n_batches] = pos;
    if (n_rows) *n_rows = opts.n_rows;
    if (n_batches) *n_batches = opts.n_batches;
    if (stream_size) *stream_size = pos + 8;
  });
}

int mi_synth_lineitem_fill(const mi_synth_options* o, uint8_t* out, int64_t cap) {
  return WrapC([&] {
    if (!out) throw InvalidInputException("mi_synth_lineitem_fill: NULL output");
    Opts opts = Resolve(o);
    std::vector<BatchLayout> layouts(static_cast<size_t>(opts.n_batches));
    std::vector<std::vector<uint8_t>> headers(static_cast<size_t>(opts.n_batches));
    ParallelFor(opts.n_batches, opts.n_threads, [&](int64_t b) {
      layouts[static_cast<size_t>(b)] = LayoutOf(opts, b);
      headers[static_cast<size_t>(b)] = HeaderOf(layouts[static_cast<size_t>(b)]);
    });
    std::vector<uint8_t> schema = EncodeSchemaMessage(LineitemSchema());
    std::vector<int64_t> pos(static_cast<size_t>(opts.n_batches) + 1);
    int64_t p = static_cast<int64_t>(schema.size());
    for (int64_t b = 0; b < opts.n_batches; b++) {
      pos[static_cast<size_t>(b)] = p;
      p += static_cast<int64_t>(headers[static_cast<size_t>(b)].size()) + layouts[static_cast<size_t>(b)].body_size;
    }
    pos[static_cast<size_t>(opts.n_batches)] = p;
    if (p + 8 > cap) throw InvalidInputException("mi_synth_lineitem_fill: buffer too small, need " + std::to_string(p + 8));
    std::memcpy(out, schema.data(), schema.size());
    ParallelFor(opts.n_batches, opts.n_threads, [&](int64_t b) {
      uint8_t* dst = out + pos[static_cast<size_t>(b)];
      const auto& h = headers[static_cast<size_t>(b)];
      std::memcpy(dst, h.data(), h.size());
      FillBatch(opts, b, layouts[static_cast<size_t>(b)], dst + h.size());
    });
    const uint8_t eos[8] = {0xFF, 0xFF, 0xFF, 0xFF, 0, 0, 0, 0};
    std::memcpy(out + p, eos, 8);
  });
}

}  // extern "C"
