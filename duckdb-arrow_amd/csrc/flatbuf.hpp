// flatbuf.hpp -- minimal FlatBuffers reader + builder, just enough for Arrow's Message.fbs / Schema.fbs /
// File.fbs.  The reference gets this from nanoarrow_ipc's bundled flatcc (CMakeLists.txt:7-13, not vendored);
// here it is written from the published FlatBuffers binary format.
//
// Reader: every access is bounds-checked against the buffer; a malformed buffer yields "absent" (default value)
// or ok() == false, never an out-of-range read.
// Builder: back-to-front construction like the canonical implementation; output passes pyarrow's verifier.
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace miarrow {
namespace fb {

template <typename T>
static inline T load(const uint8_t* p) {
  T v;
  std::memcpy(&v, p, sizeof(T));
  return v;
}

struct Buf {
  const uint8_t* base = nullptr;
  int64_t size = 0;
  bool in(int64_t pos, int64_t n) const { return pos >= 0 && n >= 0 && pos + n <= size; }
};

// A table view: position of the table inside the buffer. pos < 0 = absent.
struct Table {
  const Buf* b = nullptr;
  int64_t pos = -1;

  explicit operator bool() const { return pos >= 0; }

  // absolute position of field `id`'s storage, -1 when absent
  int64_t field(int id) const {
    if (pos < 0 || !b->in(pos, 4)) return -1;
    int64_t vt = pos - load<int32_t>(b->base + pos);
    if (!b->in(vt, 4)) return -1;
    uint16_t vt_size = load<uint16_t>(b->base + vt);
    int64_t slot = 4 + 2 * static_cast<int64_t>(id);
    if (slot + 2 > vt_size || !b->in(vt + slot, 2)) return -1;
    uint16_t off = load<uint16_t>(b->base + vt + slot);
    if (off == 0 || !b->in(pos + off, 1)) return -1;
    return pos + off;
  }
  template <typename T>
  T scalar(int id, T dflt) const {
    int64_t p = field(id);
    if (p < 0 || !b->in(p, sizeof(T))) return dflt;
    return load<T>(b->base + p);
  }
  // follow a uoffset stored at `p`
  int64_t indirect(int64_t p) const {
    if (p < 0 || !b->in(p, 4)) return -1;
    int64_t tgt = p + load<uint32_t>(b->base + p);
    if (!b->in(tgt, 4)) return -1;
    return tgt;
  }
  Table table(int id) const { return Table{b, indirect(field(id))}; }
  // vector field: returns position of element 0, sets len
  int64_t vector(int id, uint32_t* len) const {
    *len = 0;
    int64_t v = indirect(field(id));
    if (v < 0) return -1;
    *len = load<uint32_t>(b->base + v);
    return v + 4;
  }
  bool string(int id, std::string* out) const {
    uint32_t len;
    int64_t s = vector(id, &len);
    out->clear();
    if (s < 0) return false;
    if (!b->in(s, len)) return false;
    out->assign(reinterpret_cast<const char*>(b->base + s), len);
    return true;
  }
  // element i of a vector of tables
  Table vector_table(int64_t vec_pos, uint32_t i) const { return Table{b, indirect(vec_pos + 4 * static_cast<int64_t>(i))}; }
};

static inline Table root(const Buf* b) {
  if (!b->in(0, 4)) return Table{b, -1};
  int64_t off = load<uint32_t>(b->base);
  if (!b->in(off, 4)) return Table{b, -1};
  return Table{b, off};
}

// ------------------------------------------------------------------------------------------------ builder
class Builder {
 public:
  using Offset = uint32_t;  // distance from the END of the buffer

  explicit Builder(size_t initial = 1024) : buf_(initial), head_(initial) {}

  Offset size() const { return static_cast<Offset>(buf_.size() - head_); }

  Offset CreateString(const std::string& s) { return CreateString(s.data(), s.size()); }
  Offset CreateString(const char* s, size_t len) {
    Prep(4, len + 1);
    Fill(1);  // terminating NUL
    PushBytes(reinterpret_cast<const uint8_t*>(s), len);
    PushScalar<uint32_t>(static_cast<uint32_t>(len));
    return size();
  }

  // vector of uoffsets to previously created objects (tables / strings)
  Offset CreateOffsetVector(const std::vector<Offset>& elems) {
    StartVector(elems.size(), 4, 4);
    for (size_t i = elems.size(); i-- > 0;) PushScalar<uint32_t>(ReferTo(elems[i]));
    return EndVector(elems.size());
  }
  // vector of fixed-size structs / scalars given as raw little-endian bytes
  Offset CreateStructVector(const void* data, size_t count, size_t elem_size, size_t align) {
    StartVector(count, elem_size, align);
    PushBytes(static_cast<const uint8_t*>(data), count * elem_size);
    return EndVector(count);
  }

  void StartTable() {
    fields_.clear();
    table_start_ = size();
  }
  template <typename T>
  void AddScalar(int id, T v, T dflt) {
    if (v == dflt) return;  // defaults are not stored
    Align(sizeof(T));
    PushScalar<T>(v);
    fields_.push_back({id, size()});
  }
  template <typename T>
  void AddScalarForce(int id, T v) {
    Align(sizeof(T));
    PushScalar<T>(v);
    fields_.push_back({id, size()});
  }
  void AddOffset(int id, Offset off) {
    if (off == 0) return;
    Align(4);
    PushScalar<uint32_t>(ReferTo(off));
    fields_.push_back({id, size()});
  }
  Offset EndTable() {
    Align(4);
    PushScalar<int32_t>(0);  // soffset to the vtable, patched below
    Offset table_loc = size();
    int max_id = -1;
    for (auto& f : fields_) max_id = f.first > max_id ? f.first : max_id;
    uint16_t vt_size = static_cast<uint16_t>(4 + 2 * (max_id + 1));
    uint16_t obj_size = static_cast<uint16_t>(table_loc - table_start_);
    std::vector<uint16_t> slots(static_cast<size_t>(max_id + 1), 0);
    for (auto& f : fields_) slots[static_cast<size_t>(f.first)] = static_cast<uint16_t>(table_loc - f.second);
    // the vtable is 2-byte aligned; keep the table itself 4-aligned by padding before the vtable if needed
    for (size_t i = slots.size(); i-- > 0;) PushScalar<uint16_t>(slots[i]);
    PushScalar<uint16_t>(obj_size);
    PushScalar<uint16_t>(vt_size);
    Offset vt_loc = size();
    int32_t soffset = static_cast<int32_t>(vt_loc) - static_cast<int32_t>(table_loc);
    std::memcpy(buf_.data() + buf_.size() - table_loc, &soffset, 4);
    return table_loc;
  }

  // Finishes the buffer with `root_table` as root. Returns the final bytes.
  std::vector<uint8_t> Finish(Offset root_table) {
    Prep(min_align_, 4);
    PushScalar<uint32_t>(ReferTo(root_table));
    return std::vector<uint8_t>(buf_.begin() + static_cast<std::ptrdiff_t>(head_), buf_.end());
  }

 private:
  void Grow(size_t need) {
    if (head_ >= need) return;
    size_t old = buf_.size();
    size_t grow = old;
    while (head_ + grow < need + 16) grow *= 2;
    std::vector<uint8_t> nb(old + grow, 0);
    std::memcpy(nb.data() + grow + head_, buf_.data() + head_, old - head_);
    head_ += grow;
    buf_.swap(nb);
  }
  void Fill(size_t n) {
    Grow(n);
    head_ -= n;
    std::memset(buf_.data() + head_, 0, n);
  }
  void Align(size_t a) {
    if (a > min_align_) min_align_ = a;
    Fill((~static_cast<size_t>(size()) + 1) & (a - 1));
  }
  // make sure that after writing `additional` more bytes the buffer is aligned to `a`
  void Prep(size_t a, size_t additional) {
    if (a > min_align_) min_align_ = a;
    size_t pad = (~(static_cast<size_t>(size()) + additional) + 1) & (a - 1);
    Fill(pad);
  }
  void PushBytes(const uint8_t* p, size_t n) {
    Grow(n);
    head_ -= n;
    if (n) std::memcpy(buf_.data() + head_, p, n);
  }
  template <typename T>
  void PushScalar(T v) {
    PushBytes(reinterpret_cast<const uint8_t*>(&v), sizeof(T));
  }
  uint32_t ReferTo(Offset off) {
    Align(4);
    return size() - off + 4;
  }
  void StartVector(size_t count, size_t elem_size, size_t align) {
    Prep(4, count * elem_size);
    Prep(align, count * elem_size);
  }
  Offset EndVector(size_t count) {
    PushScalar<uint32_t>(static_cast<uint32_t>(count));
    return size();
  }

  std::vector<uint8_t> buf_;
  size_t head_;
  size_t min_align_ = 4;
  Offset table_start_ = 0;
  std::vector<std::pair<int, Offset>> fields_;
};

}  // namespace fb
}  // namespace miarrow
