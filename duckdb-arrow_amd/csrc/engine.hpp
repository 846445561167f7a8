// engine.hpp -- device context and transcode plans (host side of the fused kernel launch).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>
#include <vector>

#include "ipc_format.hpp"
#include "kernels.hpp"

namespace miarrow {

#define MI_HIP_CHECK(expr)                                                                                  \
  do {                                                                                                      \
    hipError_t _e = (expr);                                                                                 \
    if (_e != hipSuccess) {                                                                                 \
      throw ::miarrow::Exception(_e == hipErrorOutOfMemory ? MI_ENOMEM : MI_EIO,                            \
                                 std::string(#expr) + " failed: " + hipGetErrorString(_e));                 \
    }                                                                                                       \
  } while (0)

struct Context {
  int device = 0;
  int num_cus = 256;
  hipStream_t stream = nullptr;       // compute
  hipStream_t h2d_stream = nullptr;   // pinned host -> HBM
  hipStream_t d2h_stream = nullptr;   // HBM -> pinned host

  // Where the device hangs in the host: its NUMA node and that node's CPUs (/sys/bus/pci/devices/<bus id>/numa_node and
  // local_cpulist; -1 / empty when the platform does not say).  The library's OWN host threads -- read-ahead producers, the
  // I/O pool while it works for this context -- run there and allocate their pinned buffers there (BindThisThread): a pread
  // into pinned memory followed by a DMA to the device crosses the socket interconnect twice when it happens on the other
  // socket (two-socket boxes: 199 against 300 M rows/s for the same host-consumer scan, by where the threads happened to
  // run).  The caller's threads are never touched.  MI_NUMA_BIND=0 turns it off.
  int numa_node = -1;
  std::vector<int> local_cpus;
  std::string local_cpulist;            // as the kernel prints it ("0-63,128-191")
  void BindThisThread() const;          // affinity = local_cpus (within what the thread may use), allocations prefer numa_node
  // RAII: allocations of the calling thread prefer the device's node inside the scope (pinned buffers the caller's thread makes)
  struct PreferNode {
    explicit PreferNode(const Context* c);
    ~PreferNode();
    bool on = false;
    int saved_mode = 0;                 // the caller's own policy, put back when the scope ends
    unsigned long saved_mask[16] = {0};
  };

  explicit Context(int device_id);
  ~Context();
  void Bind() const;  // hipSetDevice
};

int OutWidth(int32_t kind, int64_t param);

struct ClassSlice {
  int32_t cls = 0;            // device::KernelClass
  int32_t depth = 0;          // nesting depth of the tasks in this slice
  size_t tile_task_at = 0;    // index into Plan::tile_task / d_tile_task
  int32_t first_task = 0;     // index into Plan::tasks / d_tasks
  int32_t n_tasks = 0;
  int32_t tile_begin_at = 0;  // index into Plan::tile_begin / d_tile_begin (n_tasks + 1 entries)
  uint32_t total_tiles = 0;
  uint32_t misc_groups = 0;   // which kernels of the class have work.  misc: bit 0 common flat kinds, bit 1 nested kinds
                              // (list / string view / struct), bit 2 rare flat kinds; enc_string: bit 0 strings, bit 1 lists
};

// A plan = any number of (record batch, column) tasks, grouped by kernel class; Launch() enqueues one kernel per
// non-empty class (lineitem: copy + dec128 + string = 3 launches for the whole table, however many batches).
struct Plan {
  Context* ctx;
  std::vector<mi_col_task> tasks;                 // grouped by class
  std::vector<std::pair<int, int32_t>> order;     // caller order -> (class, index inside the class)
  std::vector<uint32_t> tile_begin;
  std::vector<ClassSlice> slices;                 // launch order: depth by depth, class by class
  uint32_t class_tiles[device::kNumClasses] = {0};
  mi_col_task* d_tasks = nullptr;
  uint32_t* d_tile_begin = nullptr;
  uint32_t* d_tile_task = nullptr;   // per class slice: task index (within the slice) of every tile
  uint32_t* h_tile_task = nullptr;
  size_t cap_tile_task = 0;
  std::vector<uint32_t> tile_task;
  uint32_t* d_status = nullptr;
  int64_t* d_tile_sums = nullptr;    // encode plans with string columns
  int64_t* d_gather_bases = nullptr; // gather plans: first output row of every window (indexed like tile_task)
  size_t cap_gather_bases = 0;
  int64_t* d_null_counts = nullptr;  // encode plans: one counter per task
  int64_t n_null_counts = 0;
  uint32_t total_tiles = 0;
  bool is_encode = false;
  hipStream_t last_stream = nullptr;
  int64_t bytes_read = 0, bytes_written = 0, rows = 0;
  // capacities + pinned mirrors (reusable plans)
  size_t cap_tasks = 0, cap_tile_begin = 0, cap_tile_sums = 0, cap_null_counts = 0;
  mi_col_task* h_tasks = nullptr;
  uint32_t* h_tile_begin = nullptr;
  bool reusable = false;

  Plan(Context* ctx, const mi_col_task* tasks, int32_t n_tasks);
  //! Reusable plan (scan / writer pipelines): tables are re-filled with Set() and uploaded asynchronously from
  //! pinned host mirrors, so a new record batch costs no hipMalloc and no synchronous copy.
  explicit Plan(Context* ctx);
  void Set(const mi_col_task* tasks, int32_t n_tasks, hipStream_t upload_stream);
  ~Plan();
  Plan(const Plan&) = delete;
  Plan& operator=(const Plan&) = delete;
  void Launch(hipStream_t stream);
  void LaunchSlice(const ClassSlice& cs, hipStream_t stream);
  //! Launch with hipEvents around each class; returns milliseconds per class after synchronising the stream
  void LaunchTimed(hipStream_t stream, float* ms_per_class);
  int64_t class_bytes_read[device::kNumClasses] = {0}, class_bytes_written[device::kNumClasses] = {0},
          class_rows[device::kNumClasses] = {0};
  uint32_t Status();
  std::vector<int64_t> NullCounts(bool reset);
  //! per_slot = a host copy of d_null_counts (n_null_counts entries) -> NULL counts in the caller's task order
  std::vector<int64_t> MapNullCounts(const int64_t* per_slot) const;
};

// status word -> the exception the reference would throw
void ThrowForStatus(uint32_t bits);

}  // namespace miarrow
