// common.hpp — shared host/device helpers of the gfx950 execution kernel.
// Wave size is 64 everywhere (CDNA4); nothing here is written for 32-wide warps.
#ifndef QSX_CSRC_COMMON_HPP_
#define QSX_CSRC_COMMON_HPP_

#include <hip/hip_runtime.h>

#include <map>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/qsx.h"
#include "device_common.hpp"

namespace qsx {

// ---- host-side error plumbing -------------------------------------------
void set_last_error(const char *what, hipError_t err);
int device_ready();  // QSX_OK or QSX_ERR_NO_DEVICE

#define QSX_HIP_TRY(expr)                                        \
  do {                                                           \
    hipError_t qsx_err__ = (expr);                               \
    if (qsx_err__ != hipSuccess) {                               \
      ::qsx::set_last_error(#expr, qsx_err__);                   \
      return qsx_err__ == hipErrorOutOfMemory ? QSX_ERR_OUT_OF_MEMORY : QSX_ERR_HIP; \
    }                                                            \
  } while (0)

#define QSX_REQUIRE_DEVICE()                 \
  do {                                       \
    int qsx_dev__ = ::qsx::device_ready();   \
    if (qsx_dev__ != QSX_OK) return qsx_dev__; \
  } while (0)

#define QSX_CHECK_LAUNCH() QSX_HIP_TRY(hipGetLastError())

inline hipStream_t as_stream(qsx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int grid_for(int64_t work_items, int items_per_block) {
  int64_t blocks = (work_items + items_per_block - 1) / items_per_block;
  if (blocks < 1) blocks = 1;
  if (blocks > kMaxGridBlocks) blocks = kMaxGridBlocks;
  return static_cast<int>(blocks);
}

inline int type_width(int type) {
  switch (type) {
    case QSX_INT: case QSX_FLOAT: return 4;
    case QSX_LONG: case QSX_DOUBLE: case QSX_DATE: return 8;
    default: return 0;
  }
}

inline uint64_t next_pow2(uint64_t v) {
  uint64_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

// A by-value kernel argument -> device memory.  Kernels that index a table of pointers / a configuration in their loops read it through a
// pointer (large kernarg segments slow them down); the runtime owns the lifetime of a kernel argument, unlike that of
// a pageable host buffer handed to hipMemcpyAsync, so this is the stream-ordered way to get a host struct there.
template <typename T>
__global__ void store_struct_kernel(T value, T *__restrict__ dst) {
  static_assert(sizeof(T) % 4 == 0, "word copy");
  const uint32_t *src = reinterpret_cast<const uint32_t *>(&value);
  uint32_t *out = reinterpret_cast<uint32_t *>(dst);
  for (unsigned i = threadIdx.x; i < sizeof(T) / 4; i += blockDim.x) out[i] = src[i];
}
// Every device allocation of the library: hipMalloc, and on hipErrorOutOfMemory the engine's relief hook
// (qsx_set_out_of_memory_hook: the host layer gives back its pooled output-block allocations and scratch caches) before one
// more attempt.  The library's own per-thread buffers are trimmed by the call sites that can afford it (CallScratch,
// device_slot): a call that has uploaded a table into its staging buffer must not lose it to a nested allocation.
hipError_t device_malloc(void **ptr, size_t bytes);
template <typename T>
inline hipError_t device_malloc(T **ptr, size_t bytes) { return device_malloc(reinterpret_cast<void **>(ptr), bytes); }
// The counterpart.  device_free_idle: no queued work uses the memory any more (the destroy entry points have waited for the
// device) — the allocation is kept for the next device_malloc of exactly its size instead of going back to the runtime:
// an engine creates and destroys a join table and an aggregation state per query, and a hipMalloc / hipFree pair of a few MB
// costs 0.2-0.5 ms (tools/ubench/alloc_cost.hip; the DestroyHash / DestroyAggregationState work orders of the operator bench
// took 1 ms each).  At most kIdleKeepBytes are kept; qsx_trim_scratch and an out-of-memory condition release them.
hipError_t device_free(void *ptr);
hipError_t device_free_idle(void *ptr);
size_t trim_idle_allocations();
// Allocations remember the device they were made on (calls follow the calling thread's device; an object stays on the
// device it was created on): an idle allocation is only handed out again on that device, and a destroy entry point waits
// for THAT device's queued work — not the caller's current one — before its memory is recycled.
int device_of_allocation(const void *ptr);              // -1: not a live device_malloc
hipError_t synchronize_owner_device(const void *ptr);   // hipDeviceSynchronize on the device that owns ptr

// ---- per-(host thread, device, stream) device resources ------------------------------------------------------------------
// (the key includes the calling thread's current device: the null stream is a different queue on every device)
// Calls issued by one thread on one stream are ordered on the device, so such a pair can own buffers that every call
// reuses without an allocator on the hot path: a grow-only scratch arena, a pinned + device staging pair for host tables,
// small device slots for by-value structs.  ONE registry for the whole library (runtime.hip) holds them — per thread, so no
// lock on the hot path — and gives them back:
//   * when the thread exits (the registry's destructor),
//   * when the stream is destroyed through qsx_stream_destroy (the calling thread's entries for it),
//   * on qsx_trim_scratch() (everything the calling thread holds; the host layer calls it on its out-of-memory path),
//   * before a failed allocation inside the library is retried once — then only the entries of the thread's OTHER streams
//     (trim_thread_resources_sparing): the call that is allocating may already have staged a table into its stream's
//     StagedBuffer, filled a device slot or carved its arena, and its kernels are about to be launched on them.
struct ScratchArena {
  void *base = nullptr;
  size_t capacity = 0;
  // Reservations of calls that are still on the host's stack (an entry point that calls another entry point's launcher
  // while it holds scratch of its own: the LIP bitmaps of qsx_join_probe_lip under launch_probe's two-pass scratch).
  // `floor` = first byte no live reservation has handed out; a nested reservation is carved above it and the arena is
  // neither grown nor freed while `live` > 0.
  size_t floor = 0;
  int live = 0;
};
struct StagedBuffer {
  void *pinned = nullptr;
  void *device = nullptr;
  size_t capacity = 0;
  hipEvent_t copied = nullptr;
};
ScratchArena &thread_scratch_arena(hipStream_t stream);
StagedBuffer &thread_staged_buffer(hipStream_t stream);
void *&thread_device_slot(hipStream_t stream, const void *type_tag);
void release_thread_stream(hipStream_t stream);     // this thread's entries for `stream` (the stream is idle or being destroyed)
size_t trim_thread_resources();                      // everything this thread holds; returns the device bytes released
size_t trim_thread_resources_sparing(hipStream_t stream);   // everything this thread holds for OTHER (device, stream) pairs

// One device slot per (host thread, stream) for a by-value struct: work on one stream is ordered, so the store of the next
// call cannot overtake the kernel still reading the slot, and no allocator is involved on the update path.
template <typename T>
static T *device_slot(hipStream_t stream) {
  static const char tag = 0;
  void *&p = thread_device_slot(stream, &tag);
  if (p != nullptr) return static_cast<T *>(p);
  if (device_malloc(&p, sizeof(T)) != hipSuccess) {
    (void)hipGetLastError();
    (void)trim_thread_resources_sparing(stream);   // never this stream's own buffers: the caller may have staged into them
    if (device_malloc(&p, sizeof(T)) != hipSuccess) {
      (void)hipGetLastError();
      p = nullptr;
      return nullptr;
    }
  }
  return static_cast<T *>(p);
}

// Host tables of a call -> device memory, stream-ordered, without handing pageable memory to hipMemcpyAsync: one pinned
// staging buffer and one device buffer per (host thread, stream), grown on demand.  The pinned buffer is rewritten only
// after the event recorded behind its last copy has completed; the device buffer is rewritten by a copy that the stream
// orders behind the kernels of the previous call that read it.
static StagedBuffer *staged_buffer_for(hipStream_t stream, size_t bytes) {
  StagedBuffer &b = thread_staged_buffer(stream);
  if (b.capacity >= bytes) return &b;
  if (b.copied != nullptr) (void)hipEventSynchronize(b.copied);
  if (b.device != nullptr) {
    (void)hipStreamSynchronize(stream);   // kernels of earlier calls may still read the old table
    (void)device_free(b.device);
    (void)hipHostFree(b.pinned);
    b.device = b.pinned = nullptr;
    b.capacity = 0;
  }
  size_t cap = 64 * 1024;
  while (cap < bytes) cap *= 2;
  if (b.copied == nullptr && hipEventCreateWithFlags(&b.copied, hipEventDisableTiming) != hipSuccess) return nullptr;
  if (hipHostMalloc(&b.pinned, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
  if (device_malloc(&b.device, cap) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipHostFree(b.pinned);
    b.pinned = nullptr;
    return nullptr;
  }
  b.capacity = cap;
  return &b;
}
// The device address the next staged_upload of this (thread, stream) will fill: tables that hold their own addresses need
// it before they are complete.
static void *staged_device_buffer(hipStream_t stream, size_t bytes) {
  StagedBuffer *b = staged_buffer_for(stream, bytes);
  return b != nullptr ? b->device : nullptr;
}
static int staged_upload(hipStream_t stream, const void *host_src, size_t bytes) {
  StagedBuffer *b = staged_buffer_for(stream, bytes);
  if (b == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  QSX_HIP_TRY(hipEventSynchronize(b->copied));   // the previous copy out of the pinned buffer (a fresh event is complete)
  std::memcpy(b->pinned, host_src, bytes);
  QSX_HIP_TRY(hipMemcpyAsync(b->device, b->pinned, bytes, hipMemcpyHostToDevice, stream));
  QSX_HIP_TRY(hipEventRecord(b->copied, stream));
  return QSX_OK;
}

// Device scratch of ONE API call.  The stream-ordered allocator (hipMallocAsync / hipFreeAsync) is not used for it: on this
// stack a plain hipFree issued while stream-ordered allocations are in use occasionally leaves a kernel's writes to such an
// allocation lost (tools/ubench/pool_readback.hip: ~1 in 40 000 iterations read back zeros, with one host thread or four,
// any allocation size, release threshold raised or not; never without the interleaved hipFree, never with plain
// allocations) — and a library cannot keep its callers, or torch's allocator, from calling hipFree.
// Instead every (host thread, stream) owns one grow-only arena: calls issued by one thread on one stream are ordered on
// the device, so the next call may reuse the arena while the previous call's kernels are still queued.  An arena is a
// power of two up to 256 MiB and the request rounded to 64 MiB beyond (a 2.1 GB request keeps 2.1 GB, not 4 GiB); requests
// beyond kScratchKeepBytes are one-off allocations, handed to the idle pool (device_free_idle) when the call's work has finished.
constexpr size_t kScratchKeepBytes = size_t(1) << 30;
constexpr size_t kScratchPow2Bytes = size_t(256) << 20;
class CallScratch {
 public:
  explicit CallScratch(hipStream_t stream) : stream_(stream) {}
  CallScratch(const CallScratch &) = delete;
  CallScratch &operator=(const CallScratch &) = delete;
  ~CallScratch() { release(); }
  // Bytes a take() of `bytes` consumes of the reservation.
  static size_t padded(size_t bytes) { return (bytes + 255) / 256 * 256; }
  // All the scratch of the call at once (sum of padded() sizes); QSX_OK or an error status.  Reservations nest: a
  // CallScratch opened on the same (thread, stream) while another is alive gets the arena's bytes ABOVE what the live ones
  // hold — never their bytes, and never a grown (freed and reallocated) arena; when the arena has no room left for it the
  // nested reservation is a one-off allocation released when its call's work has finished.
  int reserve(size_t total) {
    release();
    total = padded(total ? total : 1);
    used_ = 0;
    ScratchArena *a = total > kScratchKeepBytes ? nullptr : &thread_scratch_arena(stream_);
    if (a != nullptr && a->live > 0) {
      if (a->capacity - a->floor >= total) {
        attach(a, a->floor, total);
        return QSX_OK;
      }
      a = nullptr;   // an outer reservation is live: the arena stays where and what it is
    }
    if (a == nullptr) {
      if (device_malloc(&one_off_, total) != hipSuccess) {
        (void)hipGetLastError();
        (void)trim_thread_resources_sparing(stream_);   // what this thread keeps for its other streams goes first
        QSX_HIP_TRY(device_malloc(&one_off_, total));
      }
      base_ = static_cast<char *>(one_off_);
      capacity_ = total;
      return QSX_OK;
    }
    if (a->capacity < total) {
      if (a->base != nullptr) {
        QSX_HIP_TRY(hipStreamSynchronize(stream_));   // kernels of earlier calls may still use the old arena
        QSX_HIP_TRY(device_free(a->base));
        a->base = nullptr;
        a->capacity = 0;
      }
      size_t cap = 1 << 20;
      while (cap < total && cap < kScratchPow2Bytes) cap *= 2;
      if (cap < total) cap = (total + (size_t(64) << 20) - 1) / (size_t(64) << 20) * (size_t(64) << 20);
      if (device_malloc(&a->base, cap) != hipSuccess) {
        (void)hipGetLastError();
        a->base = nullptr;
        (void)trim_thread_resources_sparing(stream_);   // (this stream's staged table and slots may belong to this very call)
        QSX_HIP_TRY(device_malloc(&a->base, cap));
      }
      a->capacity = cap;
    }
    attach(a, 0, total);
    return QSX_OK;
  }
  // The next piece of the reservation (256-byte aligned), nullptr when the reservation is exhausted.
  void *take(size_t bytes) {
    const size_t need = padded(bytes ? bytes : 1);
    if (base_ == nullptr || used_ + need > capacity_) return nullptr;
    void *p = base_ + used_;
    used_ += need;
    return p;
  }

 private:
  void attach(ScratchArena *a, size_t offset, size_t total) {
    arena_ = a;
    saved_floor_ = a->floor;
    a->floor = offset + total;
    a->live += 1;
    base_ = static_cast<char *>(a->base) + offset;
    capacity_ = total;
  }
  void release() {
    if (arena_ != nullptr) {
      arena_->floor = saved_floor_;
      arena_->live -= 1;
      arena_ = nullptr;
    }
    if (one_off_ != nullptr) {
      (void)hipStreamSynchronize(stream_);
      (void)device_free_idle(one_off_);   // (kept for the next request of its size: runtime.hip kIdleKeepBytes)
      one_off_ = nullptr;
    }
    base_ = nullptr;
    capacity_ = used_ = 0;
  }
  hipStream_t stream_;
  ScratchArena *arena_ = nullptr;
  size_t saved_floor_ = 0;
  void *one_off_ = nullptr;
  char *base_ = nullptr;
  size_t capacity_ = 0, used_ = 0;
};

}  // namespace qsx

#endif  // QSX_CSRC_COMMON_HPP_
