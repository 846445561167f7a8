// engine.cpp -- device context + transcode plans.  See engine.hpp.
#include "engine.hpp"

#include "ipc_stream_reader.hpp"

#include <hip/hip_runtime.h>

#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace miarrow {

Context::Context(int device_id) : device(device_id) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    // The product path has no CPU fallback: without a HIP device it fails loudly.
    throw Exception(MI_ENODEV, std::string("No HIP device available (hipGetDeviceCount: ") +
                                   (e == hipSuccess ? "0 devices" : hipGetErrorString(e)) +
                                   "); the MI355X Arrow IPC path has no CPU fallback");
  }
  if (device_id < 0 || device_id >= count) {
    throw Exception(MI_ENODEV, "HIP device " + std::to_string(device_id) + " out of range (" + std::to_string(count) + " devices)");
  }
  MI_HIP_CHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  MI_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  MI_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  MI_HIP_CHECK(hipStreamCreateWithFlags(&h2d_stream, hipStreamNonBlocking));
  MI_HIP_CHECK(hipStreamCreateWithFlags(&d2h_stream, hipStreamNonBlocking));
  // the device's place in the host (best effort: any failure leaves numa_node = -1 and nothing is bound)
  const char* nb = std::getenv("MI_NUMA_BIND");
  char bus[64] = {0};
  if (!(nb && nb[0] == '0') && hipDeviceGetPCIBusId(bus, sizeof(bus) - 1, device) == hipSuccess) {
    for (char* q = bus; *q; q++) *q = static_cast<char>(std::tolower(static_cast<unsigned char>(*q)));
    const std::string dir = std::string("/sys/bus/pci/devices/") + bus + "/";
    std::ifstream fn(dir + "numa_node"), fc(dir + "local_cpulist");
    int node = -1;
    std::string list;
    if (fn >> node && node >= 0 && std::getline(fc, list) && !list.empty()) {
      // "0-63,128-191"
      std::vector<int> cpus;
      size_t at = 0;
      bool ok = true;
      while (at < list.size() && ok) {
        size_t end = list.find(',', at);
        if (end == std::string::npos) end = list.size();
        const std::string part = list.substr(at, end - at);
        int lo = 0, hi = 0;
        const int got = std::sscanf(part.c_str(), "%d-%d", &lo, &hi);
        if (got == 1) hi = lo;
        ok = got >= 1 && lo >= 0 && hi >= lo && hi < CPU_SETSIZE;
        for (int c = lo; ok && c <= hi; c++) cpus.push_back(c);
        at = end + 1;
      }
      if (ok && !cpus.empty()) {
        numa_node = node;
        local_cpus = std::move(cpus);
        local_cpulist = list;
      }
    }
  }
  (void)hipGetLastError();
}

void Context::BindThisThread() const { BindThisThreadToNode(numa_node, local_cpus); }

Context::PreferNode::PreferNode(const Context* c) {
  if (c && c->numa_node >= 0 && c->numa_node < static_cast<int>(sizeof(saved_mask) * 8)) {
    // the calling thread is the host program's: remember its policy (most have none: MPOL_DEFAULT)
    if (syscall(SYS_get_mempolicy, &saved_mode, saved_mask, sizeof(saved_mask) * 8, nullptr, 0) != 0) return;
    miarrow::PreferNode(c->numa_node);
    on = true;
  }
}
Context::PreferNode::~PreferNode() {
  if (on) (void)syscall(SYS_set_mempolicy, saved_mode, saved_mode == 0 ? nullptr : saved_mask, saved_mode == 0 ? 0 : sizeof(saved_mask) * 8);
}

Context::~Context() {
  if (stream) (void)hipStreamDestroy(stream);
  if (h2d_stream) (void)hipStreamDestroy(h2d_stream);
  if (d2h_stream) (void)hipStreamDestroy(d2h_stream);
}

void Context::Bind() const { MI_HIP_CHECK(hipSetDevice(device)); }

int OutWidth(int32_t kind, int64_t param) {
  switch (kind) {
    case MI_K_COPY: return static_cast<int>(param);
    case MI_K_BOOL: return 1;
    case MI_K_DEC128: return static_cast<int>(param);
    case MI_K_DATE64: return 4;
    case MI_K_MUL_I32: case MI_K_MUL_I64: case MI_K_DIV_I64: return 8;
    case MI_K_STR32: case MI_K_STR64: case MI_K_FIXED_BINARY: case MI_K_DURATION: return 16;
    case MI_K_INTERVAL_MONTHS: case MI_K_INTERVAL_MDN: return 16;
    case MI_K_NARROW: return static_cast<int>((param >> 8) & 0xFF);
    case MI_K_HALF_FLOAT: return 4;
    case MI_K_NULL: return 1;
    case MI_K_STRVIEW: case MI_K_LIST32: case MI_K_LIST64: return 16;
    case MI_K_STRUCT: return 0;
    case MI_K_DICT: return 4;
    default: return 0;
  }
}

static void ValidateTask(const mi_col_task& t, size_t i) {
  auto fail = [&](const std::string& what) {
    throw InvalidInputException("task " + std::to_string(i) + ": " + what);
  };
  if (device::ClassOfKind(t.kind) < 0) fail("unknown kind " + std::to_string(t.kind));
  if (t.sel != nullptr) {  // gather mode: decode only the rows a selection vector names, compacted per 2048-row window
    if (!device::KindCanGather(t.kind)) fail("kind " + std::to_string(t.kind) + " cannot be decoded through a selection vector");
    if (t.sel_count == nullptr) fail("gather tasks need sel_count (rows selected per 2048-row window)");
    if (t.out_aux != nullptr || t.depth != 0) fail("gather tasks are top-level columns");
    if (reinterpret_cast<uintptr_t>(t.sel) % 4 != 0 || reinterpret_cast<uintptr_t>(t.sel_count) % 4 != 0) fail("selection vector misaligned");
  }
  if (t.nrows < 0) fail("negative row count");
  if (t.row_offset < 0) fail("negative row offset");
  if (t.depth < 0 || t.depth > 64) fail("bad nesting depth");
  if (t.nrows == 0) return;
  if (t.kind == MI_K_STRUCT) {
    if (t.out_validity == nullptr) fail("STRUCT tasks produce validity only: out_validity is NULL");
    if (reinterpret_cast<uintptr_t>(t.out_validity) % 8 != 0) fail("out_validity must be 8-byte aligned");
    return;
  }
  if (t.out_data == nullptr) fail("out_data is NULL");
  if (reinterpret_cast<uintptr_t>(t.out_data) % 16 != 0) fail("out_data must be 16-byte aligned");
  if (t.out_validity && reinterpret_cast<uintptr_t>(t.out_validity) % 8 != 0) fail("out_validity must be 8-byte aligned");
  if (t.validity && reinterpret_cast<uintptr_t>(t.validity) % 8 != 0) fail("validity bitmap must be 8-byte aligned");
  if (t.buf1 == nullptr && t.kind != MI_K_NULL) fail("buf1 is NULL");
  switch (t.kind) {
    case MI_K_LIST32: case MI_K_LIST64:
      if (reinterpret_cast<uintptr_t>(t.buf1) % (t.kind == MI_K_LIST32 ? 4 : 8) != 0) fail("offsets buffer misaligned");
      if (t.param < 0) fail("LIST needs param = child length");
      if (t.buf2 && t.buf2_len <= 0) fail("window table is empty");
      break;
    case MI_K_STRVIEW:
      if (reinterpret_cast<uintptr_t>(t.buf1) % 4 != 0) fail("views buffer misaligned");
      if (t.buf2_len > 0 && t.buf2 == nullptr) fail("STRVIEW needs the variadic buffer table in buf2");
      break;
    case MI_K_NARROW: {
      const int sw = static_cast<int>(t.param & 0xFF), dw = static_cast<int>((t.param >> 8) & 0xFF);
      if (!((sw == 4 && dw == 2) || (sw == 8 && (dw == 2 || dw == 4)))) fail("NARROW needs src 4->2 or 8->2/4");
      break;
    }
    case MI_K_COPY:
      if (t.param != 1 && t.param != 2 && t.param != 4 && t.param != 8 && t.param != 16) fail("COPY width must be 1,2,4,8,16");
      break;
    case MI_K_DEC128:
      if (t.param != 2 && t.param != 4 && t.param != 8) fail("DEC128 out width must be 2,4,8");
      if (reinterpret_cast<uintptr_t>(t.buf1) % 8 != 0) fail("decimal buffer must be 8-byte aligned");
      break;
    case MI_K_STR32:
    case MI_K_STR64:
      if (t.buf2 == nullptr && t.buf2_len != 0) fail("string data buffer is NULL");
      if (reinterpret_cast<uintptr_t>(t.buf1) % (t.kind == MI_K_STR32 ? 4 : 8) != 0) fail("offsets buffer misaligned");
      if (t.buf2 && reinterpret_cast<uintptr_t>(t.buf2) % 4 != 0) fail("string data buffer must be 4-byte aligned");
      if (t.buf2_len < 0) fail("negative buf2_len");
      break;
    case MI_K_FIXED_BINARY:
      if (t.param <= 0) fail("fixed binary width must be positive");
      if (reinterpret_cast<uintptr_t>(t.buf1) % 4 != 0) fail("fixed binary buffer must be 4-byte aligned");
      break;
    case MI_K_DICT: {
      int iw = static_cast<int>(t.param & 0xFF);
      if (iw != 1 && iw != 2 && iw != 4 && iw != 8) fail("dictionary index width must be 1,2,4,8");
      break;
    }
    case MI_K_MUL_I32: case MI_K_MUL_I64:
      if (t.param <= 0) fail("multiplier must be positive");
      break;
    case MI_K_DIV_I64:
      if (t.param <= 0) fail("divisor must be positive");
      break;
    case MI_K_DURATION:
      if (t.param == 0) fail("duration factor must be non-zero");
      break;
    case MI_K_ENC_COPY:
      if (t.param != 1 && t.param != 2 && t.param != 4 && t.param != 8 && t.param != 16) fail("ENC_COPY width must be 1,2,4,8,16");
      break;
    case MI_K_ENC_DEC128:
      if (t.param != 2 && t.param != 4 && t.param != 8) fail("ENC_DEC128 in width must be 2,4,8");
      break;
    case MI_K_ENC_STR32:
      if (t.out_aux == nullptr) fail("out_aux (string data) is NULL");
      if (reinterpret_cast<uintptr_t>(t.buf1) % 16 != 0) fail("string_t vector must be 16-byte aligned");
      break;
    case MI_K_ENC_LIST32:
      if (reinterpret_cast<uintptr_t>(t.buf1) % 16 != 0) fail("list_entry_t vector must be 16-byte aligned");
      break;
    default: break;
  }
  if (t.kind >= MI_K_ENC_COPY && t.kind != MI_K_ENC_COPY && t.out_validity == nullptr)
    fail("encode tasks need out_validity (the bitmap is always emitted; only a bare ENC_COPY of list offsets has none)");
}

static int64_t TaskBytesRead(const mi_col_task& t) {
  const int64_t n = t.nrows;
  int64_t b = 0;
  if (t.kind < MI_K_ENC_COPY) {
    if (t.validity && t.null_count != 0) b += (n + 7) / 8;
    switch (t.kind) {
      case MI_K_COPY: case MI_K_FIXED_BINARY: b += n * t.param; break;
      case MI_K_BOOL: b += (n + 7) / 8; break;
      case MI_K_DEC128: b += n * 16; break;
      case MI_K_DATE64: case MI_K_MUL_I64: case MI_K_DIV_I64: case MI_K_DURATION: b += n * 8; break;
      case MI_K_MUL_I32: b += n * 4; break;
      case MI_K_STR32: b += (n ? (n + 1) * 4 : 0) + t.buf2_len; break;
      case MI_K_STR64: b += (n ? (n + 1) * 8 : 0) + t.buf2_len; break;
      case MI_K_DICT: b += n * (t.param & 0xFF); break;
      case MI_K_INTERVAL_MONTHS: b += n * 4; break;
      case MI_K_INTERVAL_MDN: b += n * 16; break;
      case MI_K_NARROW: b += n * (t.param & 0xFF); break;
      case MI_K_HALF_FLOAT: b += n * 2; break;
      case MI_K_LIST32: b += n ? (n + 1) * 4 : 0; break;
      case MI_K_LIST64: b += n ? (n + 1) * 8 : 0; break;
      case MI_K_STRVIEW: b += n * 16; break;
      default: break;
    }
    if (t.out_aux) b += ((n + 63) / 64) * 8;
  } else {
    if (t.validity) b += ((n + 63) / 64) * 8;
    switch (t.kind) {
      case MI_K_ENC_COPY: case MI_K_ENC_DEC128: b += n * t.param; break;
      case MI_K_ENC_BOOL: b += n; break;
      case MI_K_ENC_STR32: case MI_K_ENC_LIST32: b += n * 16; break;  // + payload, known only after the scan (reported via buf2_len if given)
      default: break;
    }
    if (t.kind == MI_K_ENC_STR32) b += t.buf2_len;
  }
  return b;
}

static int64_t TaskBytesWritten(const mi_col_task& t) {
  const int64_t n = t.nrows;
  if (t.kind < MI_K_ENC_COPY) {
    return n * OutWidth(t.kind, t.param) + (t.out_validity ? ((n + 63) / 64) * 8 : 0);
  }
  int64_t b = t.out_validity ? (n + 7) / 8 : 0;  // bitmap
  switch (t.kind) {
    case MI_K_ENC_COPY: b += n * t.param; break;
    case MI_K_ENC_DEC128: b += n * 16; break;
    case MI_K_ENC_BOOL: b += (n + 7) / 8; break;
    case MI_K_ENC_STR32: b += (n + 1) * ((t.flags & 1) ? 8 : 4) + t.buf2_len; break;
    case MI_K_ENC_LIST32: b += (n + 1) * ((t.flags & 1) ? 8 : 4); break;
    default: break;
  }
  return b;
}

// hipMemset() is enqueued on the null stream and may return before it has run; the plans run on non-blocking streams, which
// the null stream does not order with.  Everything that must be zero before such a stream touches it is waited for.
static void SyncMemset(void* p, int value, size_t bytes) {
  MI_HIP_CHECK(hipMemset(p, value, bytes));
  MI_HIP_CHECK(hipStreamSynchronize(nullptr));
}

Plan::Plan(Context* ctx_p, const mi_col_task* in_tasks, int32_t n_tasks) : ctx(ctx_p) {
  Set(in_tasks, n_tasks, nullptr);
}

Plan::Plan(Context* ctx_p) : ctx(ctx_p), reusable(true) {
  ctx->Bind();
  MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_status), 64));
  SyncMemset(d_status, 0, 64);
}

template <typename T>
static void EnsureDevice(T** p, size_t* cap, size_t need) {
  if (need <= *cap && *p) return;
  if (*p) MI_HIP_CHECK(hipFree(*p));
  *p = nullptr;
  size_t n = std::max<size_t>(need, *cap * 2);
  n = std::max<size_t>(n, 64);
  MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  *cap = n;
}

void Plan::Set(const mi_col_task* in_tasks, int32_t n_tasks, hipStream_t upload_stream) {
  ctx->Bind();
  tasks.clear();
  order.clear();
  tile_begin.clear();
  total_tiles = 0;
  bytes_read = bytes_written = rows = 0;
  for (int c = 0; c < device::kNumClasses; c++) class_bytes_read[c] = class_bytes_written[c] = class_rows[c] = 0;
  is_encode = false;
  // group by (nesting depth, kernel class), stable inside a group: a child task runs in a later launch than its parent,
  // so it sees the parent's finished validity words
  slices.clear();
  tile_task.clear();
  for (int c = 0; c < device::kNumClasses; c++) class_tiles[c] = 0;
  int max_depth = 0;
  std::vector<mi_col_task> staged;
  std::vector<int> staged_cls;
  int64_t null_counter = 0;
  for (int32_t i = 0; i < n_tasks; i++) {
    ValidateTask(in_tasks[i], static_cast<size_t>(i));
    mi_col_task t = in_tasks[i];
    const int cls = device::ClassOfTask(t);
    if (t.kind >= MI_K_ENC_COPY) {
      is_encode = true;
      t.depth = 0;
      t.param2 = null_counter++;  // slot of this task's NULL counter
    }
    bytes_read += TaskBytesRead(t);
    bytes_written += TaskBytesWritten(t);
    rows += t.nrows;
    class_bytes_read[cls] += TaskBytesRead(t);
    class_bytes_written[cls] += TaskBytesWritten(t);
    class_rows[cls] += t.nrows;
    max_depth = std::max(max_depth, t.depth);
    staged.push_back(t);
    staged_cls.push_back(cls);
  }
  n_null_counts = null_counter;
  order.assign(static_cast<size_t>(n_tasks), {0, 0});
  for (int depth = 0; depth <= max_depth; depth++) {
    for (int c = 0; c < device::kNumClasses; c++) {
      ClassSlice sl;
      sl.cls = c;
      sl.depth = depth;
      sl.first_task = static_cast<int32_t>(tasks.size());
      sl.tile_begin_at = static_cast<int32_t>(tile_begin.size());
      sl.tile_task_at = tile_task.size();
      const int64_t tile_rows = device::TileRowsOfClass(c);
      uint64_t tiles = 0;
      uint32_t local_task = 0;
      for (size_t i = 0; i < staged.size(); i++) {
        if (staged_cls[i] != c || staged[i].depth != depth) continue;
        const mi_col_task& t = staged[i];
        tile_begin.push_back(static_cast<uint32_t>(tiles));
        const uint64_t nt = static_cast<uint64_t>((t.nrows + tile_rows - 1) / tile_rows);
        tiles += nt;
        if (tiles > 0xFFFFFFF0ull) throw InvalidInputException("plan has too many tiles");
        tile_task.insert(tile_task.end(), static_cast<size_t>(nt), local_task);
        if (c == device::kClassMisc) sl.misc_groups |= 1u << device::MiscGroupOfKind(t.kind);
        if (c == device::kClassDec128 || c == device::kClassString)   // which of the two kernel instances has tiles (device: tile_needs_mask)
          sl.misc_groups |= ((t.validity != nullptr && t.null_count != 0) || t.out_aux != nullptr) ? 2u : 1u;
        if (c == device::kClassEncString && t.kind == MI_K_ENC_LIST32) sl.misc_groups |= 2u;
        order[i] = {static_cast<int>(slices.size()), static_cast<int32_t>(local_task)};
        local_task++;
        tasks.push_back(t);
      }
      if (local_task == 0) continue;
      tile_begin.push_back(static_cast<uint32_t>(tiles));
      sl.n_tasks = static_cast<int32_t>(local_task);
      sl.total_tiles = static_cast<uint32_t>(tiles);
      total_tiles += sl.total_tiles;
      class_tiles[c] += sl.total_tiles;
      slices.push_back(sl);
    }
  }
  // device tables
  const size_t old_cap_tasks = cap_tasks, old_cap_tb = cap_tile_begin;
  EnsureDevice(&d_tasks, &cap_tasks, tasks.size());
  EnsureDevice(&d_tile_begin, &cap_tile_begin, tile_begin.size());
  const size_t old_cap_tt = cap_tile_task;
  EnsureDevice(&d_tile_task, &cap_tile_task, tile_task.size());
  if (!d_status) {
    MI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_status), 64));
    SyncMemset(d_status, 0, 64);
  }
  if (class_tiles[device::kClassEncString]) EnsureDevice(&d_tile_sums, &cap_tile_sums, 2 * static_cast<size_t>(class_tiles[device::kClassEncString]) + 1);
  if (class_tiles[device::kClassGather]) EnsureDevice(&d_gather_bases, &cap_gather_bases, tile_task.size());
  if (n_null_counts) {
    const size_t before = cap_null_counts;
    EnsureDevice(&d_null_counts, &cap_null_counts, static_cast<size_t>(n_null_counts));
    if (cap_null_counts != before) SyncMemset(d_null_counts, 0, cap_null_counts * sizeof(int64_t));
  }
  if (reusable && upload_stream) {
    if (cap_tasks != old_cap_tasks || !h_tasks) {
      if (h_tasks) MI_HIP_CHECK(hipHostFree(h_tasks));
      MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_tasks), cap_tasks * sizeof(mi_col_task), hipHostMallocDefault));
    }
    if (cap_tile_begin != old_cap_tb || !h_tile_begin) {
      if (h_tile_begin) MI_HIP_CHECK(hipHostFree(h_tile_begin));
      MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_tile_begin), cap_tile_begin * sizeof(uint32_t), hipHostMallocDefault));
    }
    if (!tasks.empty()) {
      std::memcpy(h_tasks, tasks.data(), tasks.size() * sizeof(mi_col_task));
      MI_HIP_CHECK(hipMemcpyAsync(d_tasks, h_tasks, tasks.size() * sizeof(mi_col_task), hipMemcpyHostToDevice, upload_stream));
    }
    std::memcpy(h_tile_begin, tile_begin.data(), tile_begin.size() * sizeof(uint32_t));
    MI_HIP_CHECK(hipMemcpyAsync(d_tile_begin, h_tile_begin, tile_begin.size() * sizeof(uint32_t), hipMemcpyHostToDevice, upload_stream));
    if (cap_tile_task != old_cap_tt || !h_tile_task) {
      if (h_tile_task) MI_HIP_CHECK(hipHostFree(h_tile_task));
      MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_tile_task), cap_tile_task * sizeof(uint32_t), hipHostMallocDefault));
    }
    if (!tile_task.empty()) {
      std::memcpy(h_tile_task, tile_task.data(), tile_task.size() * sizeof(uint32_t));
      MI_HIP_CHECK(hipMemcpyAsync(d_tile_task, h_tile_task, tile_task.size() * sizeof(uint32_t), hipMemcpyHostToDevice, upload_stream));
    }
  } else {
    if (!tasks.empty())
      MI_HIP_CHECK(hipMemcpy(d_tasks, tasks.data(), tasks.size() * sizeof(mi_col_task), hipMemcpyHostToDevice));
    MI_HIP_CHECK(hipMemcpy(d_tile_begin, tile_begin.data(), tile_begin.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (!tile_task.empty())
      MI_HIP_CHECK(hipMemcpy(d_tile_task, tile_task.data(), tile_task.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
}

Plan::~Plan() {
  if (d_tasks) (void)hipFree(d_tasks);
  if (d_tile_begin) (void)hipFree(d_tile_begin);
  if (d_status) (void)hipFree(d_status);
  if (d_tile_sums) (void)hipFree(d_tile_sums);
  if (d_gather_bases) (void)hipFree(d_gather_bases);
  if (d_null_counts) (void)hipFree(d_null_counts);
  if (h_tasks) (void)hipHostFree(h_tasks);
  if (h_tile_begin) (void)hipHostFree(h_tile_begin);
  if (d_tile_task) (void)hipFree(d_tile_task);
  if (h_tile_task) (void)hipHostFree(h_tile_task);
}

void Plan::LaunchSlice(const ClassSlice& cs, hipStream_t s) {
  if (cs.total_tiles == 0) return;
  const mi_col_task* t = d_tasks + cs.first_task;
  const uint32_t* tb = d_tile_begin + cs.tile_begin_at;
  const uint32_t* tt = d_tile_task + cs.tile_task_at;
  switch (cs.cls) {
    case device::kClassEncFixed:
      MI_HIP_CHECK(device::LaunchEncodeFixed(t, tb, tt, cs.n_tasks, cs.total_tiles, d_null_counts, s));
      break;
    case device::kClassEncString:
      MI_HIP_CHECK(device::LaunchEncodeString(t, tb, tt, cs.n_tasks, cs.total_tiles, d_tile_sums, d_null_counts, d_status, (cs.misc_groups & 2u) != 0, s));
      break;
    case device::kClassGather:
      MI_HIP_CHECK(device::LaunchGather(t, tb, tt, cs.n_tasks, cs.total_tiles, d_gather_bases + cs.tile_task_at, d_status, s));
      break;
    default:
      MI_HIP_CHECK(device::LaunchTranscode(cs.cls, t, tb, tt, cs.n_tasks, cs.total_tiles, d_status, cs.misc_groups, s));
      break;
  }
}

void Plan::Launch(hipStream_t s) {
  ctx->Bind();
  if (!s) s = ctx->stream;
  last_stream = s;
  for (const auto& sl : slices) LaunchSlice(sl, s);
}

void Plan::LaunchTimed(hipStream_t s, float* ms_per_class) {
  ctx->Bind();
  if (!s) s = ctx->stream;
  last_stream = s;
  std::vector<hipEvent_t> ev(slices.size() + 1);
  for (auto& e : ev) MI_HIP_CHECK(hipEventCreate(&e));
  MI_HIP_CHECK(hipEventRecord(ev[0], s));
  for (size_t i = 0; i < slices.size(); i++) {
    LaunchSlice(slices[i], s);
    MI_HIP_CHECK(hipEventRecord(ev[i + 1], s));
  }
  MI_HIP_CHECK(hipStreamSynchronize(s));
  for (int c = 0; c < device::kNumClasses; c++) ms_per_class[c] = 0;
  for (size_t i = 0; i < slices.size(); i++) {
    float ms = 0;
    MI_HIP_CHECK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
    ms_per_class[slices[i].cls] += ms;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
}

uint32_t Plan::Status() {
  ctx->Bind();
  hipStream_t s = last_stream ? last_stream : ctx->stream;
  MI_HIP_CHECK(hipStreamSynchronize(s));
  uint32_t bits = 0;
  MI_HIP_CHECK(hipMemcpy(&bits, d_status, sizeof(bits), hipMemcpyDeviceToHost));
  if (bits) {   // on the plan's own stream: a memset on the null stream is not ordered with the non-blocking streams the plans run on
    MI_HIP_CHECK(hipMemsetAsync(d_status, 0, sizeof(bits), s));
    MI_HIP_CHECK(hipStreamSynchronize(s));
  }
  return bits;
}

std::vector<int64_t> Plan::NullCounts(bool reset) {
  ctx->Bind();
  std::vector<int64_t> per_slot(static_cast<size_t>(n_null_counts), 0);
  if (n_null_counts) {
    hipStream_t s = last_stream ? last_stream : ctx->stream;
    MI_HIP_CHECK(hipStreamSynchronize(s));
    MI_HIP_CHECK(hipMemcpy(per_slot.data(), d_null_counts, per_slot.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (reset) {
      // on the plan's own stream and finished before this returns: hipMemset() goes to the null stream, which is not ordered
      // with the non-blocking streams the plans run on -- the next row group's kernels could add their NULLs to the counters
      // BEFORE the memset of this row group cleared them (seen as null_count 0 in a file written by several sink threads)
      MI_HIP_CHECK(hipMemsetAsync(d_null_counts, 0, per_slot.size() * sizeof(int64_t), s));
      MI_HIP_CHECK(hipStreamSynchronize(s));
    }
  }
  return MapNullCounts(per_slot.data());
}

std::vector<int64_t> Plan::MapNullCounts(const int64_t* per_slot) const {
  // back to the caller's task order (non-encode tasks report 0)
  std::vector<int64_t> out(order.size(), 0);
  for (size_t i = 0; i < order.size(); i++) {
    const mi_col_task& t = tasks[static_cast<size_t>(slices[static_cast<size_t>(order[i].first)].first_task + order[i].second)];
    if (t.kind >= MI_K_ENC_COPY) out[i] = per_slot[static_cast<size_t>(t.param2)];
  }
  return out;
}

void ThrowForStatus(uint32_t bits) {
  if (bits == 0) return;
  if (bits & MI_ST_BAD_OFFSETS)
    throw InternalException("Arrow IPC validation failed: offsets buffer is not monotonically non-decreasing or exceeds the data buffer");
  if (bits & MI_ST_STRING_TOO_LARGE) throw ConversionException("DuckDB does not support Strings over 4GB");
  if (bits & MI_ST_MUL_OVERFLOW) throw ConversionException("Could not convert Timestamp to Microsecond");
  if (bits & MI_ST_INDEX_RANGE) throw ConversionException("DuckDB only supports indices that fit on an uint32");
  if (bits & MI_ST_DICT_INDEX) throw InternalException("Arrow IPC validation failed: dictionary index out of range");
  if (bits & MI_ST_DECIMAL_RANGE) throw ConversionException("Decimal value does not fit the physical type of its declared precision");
  if (bits & MI_ST_DECOMPRESS)
    throw IOException("LZ4_FRAME compressed buffer is malformed or does not decompress to its declared size (Expected decompressed size mismatch)");
  if (bits & MI_ST_INTERNAL) throw InternalException("a kernel gave up waiting for another workgroup (bounded spin exceeded)");
  if (bits & MI_ST_OFFSET_OVERFLOW)
    throw InvalidInputException(
        "Arrow Appender: The maximum total string size for regular string buffers is 2147483647 but the offset exceeds this.\n"
        "* SET arrow_large_buffer_size=true to use large string buffers");
  throw InternalException("unknown device status " + std::to_string(bits));
}

}  // namespace miarrow
