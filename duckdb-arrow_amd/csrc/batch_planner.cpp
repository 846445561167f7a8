// batch_planner.cpp -- see batch_planner.hpp.
#include "batch_planner.hpp"

#include <algorithm>
#include <cstring>

namespace miarrow {

namespace {
size_t RoundUp(size_t v, size_t a) { return (v + a - 1) / a * a; }
void* OffsetHandle(int64_t off) { return reinterpret_cast<void*>(static_cast<uintptr_t>(off >= 0 ? off + 1 : 0)); }
}  // namespace

void BatchPlanner::Clear() {
  nodes.clear();
  tasks.clear();
  aux.clear();
  upload.clear();
  aux_fixups.clear();
  arena_bytes = 0;
}

size_t BatchPlanner::Reserve(size_t bytes) {
  const size_t at = arena_bytes;
  arena_bytes += RoundUp(bytes, 256);
  return at;
}

// Every output array starts on an `array_align` boundary of the arena.  Measured on MI355X with fresh processes on one box
// (SF10 lineitem resident in HBM, ms per step): 256 B alignment 3.39-3.43, 4 KiB 3.39, 64 KiB 3.34-3.36, 2 MiB 3.40 -- the
// copy and dec128 kernels gain 2-3 % when a 16 KB tile never straddles a 64 KiB page fragment; that costs ~32 KiB of
// padding per (batch, column), 0.5 GB of 288 GB at SF10.  The scan operator's slots (PCIe bound) stay at 256 B.
void BatchPlanner::Alloc(int64_t rows, int32_t width, bool with_validity, size_t* data_off, int64_t* valid_off) {
  const size_t a = std::max<size_t>(opts.array_align, 256);
  *data_off = arena_bytes;
  arena_bytes += RoundUp(static_cast<size_t>(rows) * static_cast<size_t>(std::max(width, 1)) + 16, a);
  *valid_off = -1;
  if (with_validity) {
    *valid_off = static_cast<int64_t>(arena_bytes);
    arena_bytes += RoundUp(static_cast<size_t>((rows + 63) / 64) * 8 + 8, rows >= 65536 ? a : 256);
  }
}

std::pair<size_t, size_t> BatchPlanner::AddAbsentColumn(int64_t n, int32_t width) {
  size_t d;
  int64_t v;
  Alloc(n, width, true, &d, &v);
  return {d, static_cast<size_t>(v)};
}

int32_t BatchPlanner::AddColumn(const BatchPlacement& where, int32_t decoded_node, int64_t extra_rows) {
  const int64_t n = where.batch->nodes[static_cast<size_t>(decoded_node)].length;
  std::vector<int64_t> win;
  for (int64_t r = 0; r < n; r += MI_VECTOR_SIZE) win.push_back(r);
  win.push_back(n);
  if (n == 0) win.push_back(0);
  return AddNode(where, decoded_node, std::move(win), true, -1, 0, -1, extra_rows);
}

// One field node (and, recursively, its children): output slots + the transcode task.
// `win` = first row of every top-level 2048-row chunk window in this node's row space (+ the end): top-level columns and
// struct children of them have win[k] = 2048k (the tiles themselves); the child of a list starts its window k at
// offsets[win[k]], the child of a fixed_size_list at size * win[k].
int32_t BatchPlanner::AddNode(const BatchPlacement& where, int32_t ni, std::vector<int64_t> win, bool win_is_tiles,
                              int64_t parent_valid_off, int32_t parent_div, int32_t parent, int64_t extra_rows) {
  const DecodedBatch& b = *where.batch;
  const DecodedNode& nd = b.nodes[static_cast<size_t>(ni)];
  int32_t kind, w, nb;
  int64_t param;
  if (!nd.field->Plan(&kind, &param, &w, &nb, nd.value_only))
    throw NotImplementedException("Arrow type " + nd.field->Format() + " of field '" + nd.field->name + "' is not decoded by the MI355X scan path");
  const int64_t n = nd.length;
  const int32_t idx = static_cast<int32_t>(nodes.size());
  nodes.emplace_back();
  auto span = [&](size_t k) { return k < nd.spans.size() ? nd.spans[k] : mi_buffer_span{0, 0}; };
  {
    PlannedNode& o = nodes.back();
    o.kind = kind;
    o.width = w;
    o.param = param;
    o.nrows = n;
    o.null_count = nd.null_count;
    o.arrow_type = nd.field->type;
    o.depth = nd.depth;
    o.parent = parent;
    o.win = win;
    o.source_node = ni;
    // reference behaviour for plain fixed-width columns: the vector aliases the Arrow buffer (DirectConversion) and an
    // array without NULLs leaves the ValidityMask unset
    if (opts.zero_copy_direct && kind == MI_K_COPY && nd.null_count == 0 && parent_valid_off < 0 && nd.spans.size() > 1 &&
        !(where.no_alias && (*where.no_alias)[static_cast<size_t>(ni)]) && extra_rows == 0 && where.alloc_rows < 0) {
      o.alias_body_off = nd.spans[1].offset;
      return idx;
    }
    const bool all_valid = opts.unset_all_valid && nd.null_count == 0 && parent_valid_off < 0 && extra_rows == 0 && kind != MI_K_NULL;
    Alloc((where.alloc_rows >= 0 && nd.depth == 0 ? where.alloc_rows : n) + extra_rows, w, !all_valid, &o.data_off, &o.valid_off);
  }
  for (const auto& sp : nd.spans)
    if (sp.length > 0) upload.emplace_back(sp.offset, sp.length);
  const size_t data_off = nodes[static_cast<size_t>(idx)].data_off;
  const int64_t valid_off = nodes[static_cast<size_t>(idx)].valid_off;
  auto consumer_addr = [&](int64_t body_offset) { return where.consumer_base + static_cast<uint64_t>(body_offset); };
  mi_col_task t;
  std::memset(&t, 0, sizeof(t));
  t.out_data = OffsetHandle(static_cast<int64_t>(data_off));
  t.out_validity = OffsetHandle(valid_off);
  t.out_aux = OffsetHandle(parent_valid_off);
  t.flags = parent_div;
  t.depth = nd.depth;
  t.validity = span(0).length ? where.in_base + span(0).offset : nullptr;
  t.buf1 = nd.spans.size() > 1 ? where.in_base + span(1).offset : where.in_base;
  t.nrows = n;
  t.null_count = nd.null_count;
  t.kind = kind;
  t.param = param;
  std::vector<int64_t> child_win;
  int64_t aux_at = -1;
  switch (kind) {
    case MI_K_STR32: case MI_K_STR64:
      t.buf2 = where.in_base + span(2).offset;
      t.buf2_len = span(2).length;
      t.ptr_base = consumer_addr(span(2).offset);
      break;
    case MI_K_FIXED_BINARY:
      t.ptr_base = consumer_addr(span(1).offset);
      break;
    case MI_K_STRVIEW: {
      aux_at = static_cast<int64_t>(aux.size());
      for (size_t k = 2; k < nd.spans.size(); k++) {
        aux.push_back(consumer_addr(nd.spans[k].offset));
        aux.push_back(static_cast<uint64_t>(nd.spans[k].length));
      }
      if (nd.spans.size() <= 2) { aux.push_back(0); aux.push_back(0); }
      t.buf2_len = static_cast<int64_t>(nd.spans.size() > 2 ? nd.spans.size() - 2 : 0);
      break;
    }
    case MI_K_DICT: {
      if (!where.dict_len) throw InternalException("dictionary-encoded column without a dictionary resolver");
      nodes[static_cast<size_t>(idx)].dict_id = nd.field->dict_id;
      t.param2 = where.dict_len(nd.field->dict_id);
      break;
    }
    case MI_K_LIST32: case MI_K_LIST64: {
      if (nd.children.size() != 1) throw InternalException("list field without exactly one child");
      t.param = b.nodes[static_cast<size_t>(nd.children[0])].length;
      if (!win_is_tiles) {
        aux_at = static_cast<int64_t>(aux.size());
        for (int64_t r : win) aux.push_back(static_cast<uint64_t>(r));
        t.buf2_len = static_cast<int64_t>(win.size());
      }
      // the child's windows start at offsets[win[k]] (read from the host copy of the body)
      const uint8_t* offs = b.body + span(1).offset;
      child_win.reserve(win.size());
      int64_t prev = 0;
      for (int64_t r : win) {
        int64_t v = 0;
        if (n > 0) {
          if (r < 0 || r > n) throw InternalException("Arrow IPC validation failed: list window outside the column");
          if (kind == MI_K_LIST32) { int32_t x; std::memcpy(&x, offs + 4 * r, 4); v = x; }
          else std::memcpy(&v, offs + 8 * r, 8);
          // the offsets sampled here place the child vectors of every chunk: they are checked on the host (the device
          // checks all of them, but only flags) so that no window ever points outside the child column
          if (v < prev || v > t.param)
            throw InternalException("Arrow IPC validation failed: offsets buffer is not monotonically non-decreasing or exceeds the data buffer");
          prev = v;
        }
        child_win.push_back(v);
      }
      break;
    }
    default: break;
  }
  nodes[static_cast<size_t>(idx)].ptr_base = t.ptr_base;
  nodes[static_cast<size_t>(idx)].heap_size = kind == MI_K_FIXED_BINARY ? span(1).length : t.buf2_len;
  // a struct without NULLs of its own or of a parent has nothing to compute when its validity stays unset
  const bool has_work = n > 0 && !(kind == MI_K_STRUCT && valid_off < 0);
  if (has_work) {
    if (aux_at >= 0) aux_fixups.emplace_back(tasks.size(), static_cast<size_t>(aux_at));
    nodes[static_cast<size_t>(idx)].task = static_cast<int32_t>(tasks.size());
    tasks.push_back(t);
  }
  if (kind == MI_K_LIST32 || kind == MI_K_LIST64) {
    const int32_t c = AddNode(where, nd.children[0], child_win, false, -1, 0, idx, 0);
    nodes[static_cast<size_t>(idx)].children.push_back(c);
  } else if (kind == MI_K_STRUCT && !nd.children.empty()) {
    const bool fixed = nd.field->type == MI_AT_FIXED_LIST;
    const int64_t size = fixed ? param : 1;
    std::vector<int64_t> cw;
    for (int64_t r : win) cw.push_back(r * size);
    for (int32_t cn : nd.children) {
      const int32_t c = AddNode(where, cn, cw, win_is_tiles && !fixed, valid_off, static_cast<int32_t>(fixed ? size : 1), idx, 0);
      nodes[static_cast<size_t>(idx)].children.push_back(c);
    }
  }
  return idx;
}

void BatchPlanner::Rebase(size_t first_task, uint8_t* arena_base, const uint8_t* aux_base) {
  auto addr = [&](void* handle) -> void* {
    const uintptr_t h = reinterpret_cast<uintptr_t>(handle);
    return h ? arena_base + (h - 1) : nullptr;
  };
  for (size_t i = first_task; i < tasks.size(); i++) {
    mi_col_task& t = tasks[i];
    t.out_data = addr(t.out_data);
    t.out_validity = addr(t.out_validity);
    t.out_aux = addr(t.out_aux);
  }
  for (auto& fx : aux_fixups)
    if (fx.first >= first_task) tasks[fx.first].buf2 = aux_base + fx.second * 8;
}

}  // namespace miarrow
