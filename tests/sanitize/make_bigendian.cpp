// Test fixture generator: little-endian Arrow IPC stream -> the same stream as a big-endian producer writes it
// (Schema.endianness = Big, every multi-byte number in the record-batch bodies byte-swapped; metadata and message
// prefixes stay little-endian).  pyarrow cannot write such a stream on a little-endian host but reads it (it swaps on
// load), which is how tests/test_bigendian.py checks this tool before it trusts it.  The buffer-width table below is the
// tool's own (Arrow columnar format, "Endianness"), independent of the product's.
//   g++ -std=c++17 -O1 -I include tests/sanitize/make_bigendian.cpp duckdb-arrow_amd/csrc/ipc_format.cpp -o make_bigendian
//   ./make_bigendian in.arrows out.arrows
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../duckdb-arrow_amd/csrc/ipc_format.hpp"

using namespace miarrow;

// element layout of one buffer: 0 = bytes, N = N-byte integers, -1 = month_day_nano {4,4,8}, -2 = string view
static void Layouts(const ArrowField& f, const RecordBatchMeta& meta, size_t* variadic, bool values_only, std::vector<int>* out) {
  if (f.has_dictionary && !values_only) {
    out->push_back(0);
    out->push_back(f.dict_index_bit_width / 8);
    return;
  }
  out->push_back(0);  // validity (absent for null / union, handled below)
  switch (f.type) {
    case MI_AT_NULL: out->pop_back(); break;
    case MI_AT_UNION: out->pop_back(); out->push_back(0); if (f.unit == 1) out->push_back(4); break;
    case MI_AT_STRUCT: case MI_AT_FIXED_LIST: break;
    case MI_AT_UTF8: case MI_AT_BINARY: out->push_back(4); out->push_back(0); break;
    case MI_AT_LARGE_UTF8: case MI_AT_LARGE_BINARY: out->push_back(8); out->push_back(0); break;
    case MI_AT_LIST: case MI_AT_MAP: out->push_back(4); break;
    case MI_AT_LARGE_LIST: out->push_back(8); break;
    case MI_AT_UTF8_VIEW: case MI_AT_BINARY_VIEW: {
      out->push_back(-2);
      const int64_t vc = *variadic < meta.variadic_counts.size() ? meta.variadic_counts[(*variadic)++] : 0;
      for (int64_t k = 0; k < vc; k++) out->push_back(0);
      break;
    }
    case MI_AT_BOOL: case MI_AT_FIXED_BINARY: out->push_back(0); break;
    case MI_AT_INT: case MI_AT_DECIMAL: case MI_AT_TIME: out->push_back(f.bit_width / 8); break;
    case MI_AT_FLOAT: out->push_back(f.precision == 0 ? 2 : f.precision == 1 ? 4 : 8); break;
    case MI_AT_DATE: out->push_back(f.unit == 0 ? 4 : 8); break;
    case MI_AT_TIMESTAMP: case MI_AT_DURATION: out->push_back(8); break;
    case MI_AT_INTERVAL: out->push_back(f.unit == 2 ? -1 : 4); break;
    default: out->push_back(0); break;
  }
  for (auto& c : f.children) Layouts(c, meta, variadic, false, out);
}

static void Reverse(uint8_t* p, int w) { std::reverse(p, p + w); }

int main(int argc, char** argv) {
  if (argc != 3) {
    std::fprintf(stderr, "usage: %s in.arrows out.arrows\n", argv[0]);
    return 2;
  }
  std::ifstream in(argv[1], std::ios::binary);
  std::vector<uint8_t> src((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  std::vector<uint8_t> dst;
  ArrowSchemaModel schema;
  size_t pos = 0;
  while (pos + 8 <= src.size()) {
    uint32_t token;
    int32_t meta_len;
    std::memcpy(&token, &src[pos], 4);
    std::memcpy(&meta_len, &src[pos + 4], 4);
    if (token != 0xFFFFFFFFu) return 1;
    if (meta_len == 0) {  // end of stream
      dst.insert(dst.end(), src.begin() + static_cast<long>(pos), src.begin() + static_cast<long>(pos) + 8);
      break;
    }
    const uint8_t* meta = &src[pos + 8];
    const MessageHeader h = DecodeMessageHeader(meta, meta_len);
    const size_t body_at = (pos + 8 + static_cast<size_t>(meta_len) + 7) & ~size_t(7);
    if (h.type == MessageType::SCHEMA) {
      schema = DecodeSchema(meta, meta_len);
      schema.endianness = 1;
      const std::vector<uint8_t> msg = EncodeSchemaMessage(schema);
      dst.insert(dst.end(), msg.begin(), msg.end());
    } else {
      const RecordBatchMeta rb = DecodeRecordBatch(meta, meta_len);
      if (rb.compression != -1) return 3;
      std::vector<int> layout;
      size_t variadic = 0;
      if (rb.is_dictionary) {
        for (auto& f : schema.fields)
          if (f.has_dictionary && f.dict_id == rb.dict_id) { Layouts(f, rb, &variadic, true, &layout); break; }
      } else {
        for (auto& f : schema.fields) Layouts(f, rb, &variadic, false, &layout);
      }
      if (layout.size() != rb.buffers.size()) return 4;
      std::vector<uint8_t> msg(src.begin() + static_cast<long>(pos), src.begin() + static_cast<long>(body_at) + h.body_length);
      uint8_t* body = msg.data() + (body_at - pos);
      for (size_t i = 0; i < layout.size(); i++) {
        uint8_t* p = body + rb.buffers[i].offset;
        const int64_t n = rb.buffers[i].length;
        const int w = layout[i];
        if (w > 1) {
          for (int64_t k = 0; k + w <= n; k += w) Reverse(p + k, w);
        } else if (w == -1) {
          for (int64_t k = 0; k + 16 <= n; k += 16) { Reverse(p + k, 4); Reverse(p + k + 4, 4); Reverse(p + k + 8, 8); }
        } else if (w == -2) {
          for (int64_t k = 0; k + 16 <= n; k += 16) {
            int32_t len;
            std::memcpy(&len, p + k, 4);
            Reverse(p + k, 4);
            if (len > 12) { Reverse(p + k + 8, 4); Reverse(p + k + 12, 4); }
          }
        }
      }
      dst.insert(dst.end(), msg.begin(), msg.end());
    }
    pos = body_at + static_cast<size_t>(h.body_length);
  }
  std::ofstream out(argv[2], std::ios::binary | std::ios::trunc);
  out.write(reinterpret_cast<const char*>(dst.data()), static_cast<std::streamsize>(dst.size()));
  return 0;
}
