// host_runtime.cpp — errors, the calling thread's stream, the pooled device allocations of blocks and scratch (see quickstep_gpu.hpp; what the files share: quickstep_gpu_internal.hpp)
#include "quickstep_gpu_internal.hpp"

namespace quickstep {

// ---------------------------------------------------------------------------
// errors / stream
// ---------------------------------------------------------------------------
ExecutionError::ExecutionError(const std::string &where, int status)
    : std::runtime_error(where + ": " + qsx_status_string(status) + " [" + qsx_last_error() + "]"), status_(status) {}

void CheckStatus(int status, const char *where) {
  if (status != QSX_OK) throw ExecutionError(where, status);
}

namespace host_internal {
namespace {
thread_local qsx_stream_t tls_stream = nullptr;
}  // namespace
qsx_stream_t ThreadStream() { return tls_stream; }
void SetThreadStream(qsx_stream_t s) { tls_stream = s; }


// The allocations of output blocks (one per block: stripes + null bitmaps), kept for the next block of their size class instead
// of going back to the runtime: a device allocation of a few MB costs 170-250 us (tools/ubench/alloc_cost.hip) — as much as
// all the kernels of a work order over a run of blocks.  A block is destroyed when its last BlockReference goes, and every
// work order waits for its stream before it returns, so no queued work can still use a slab that comes back here.
class BlockSlabPool {
 public:
  static BlockSlabPool &instance() {
    static BlockSlabPool *pool = new BlockSlabPool;   // never destroyed: it outlives the HIP runtime's own teardown order
    return *pool;
  }
  static constexpr std::size_t kPoolFrom = 256 * 1024, kKeepBytes = std::size_t(8) << 30;
  // *granted = the bytes to hand back with give()
  void *take(std::size_t bytes, std::size_t *granted) {
    if (bytes < kPoolFrom || std::getenv("QSX_HOST_BLOCK_POOL_OFF") != nullptr) {
      *granted = 0;
      void *p = nullptr;
      CheckStatus(qsx_device_alloc(bytes ? bytes : 8, &p), "qsx_device_alloc(block)");
      return p;
    }
    std::size_t cls = kPoolFrom;
    while (cls < bytes) cls *= 2;
    *granted = cls;
    {
      std::lock_guard<std::mutex> lock(mutex_);
      auto it = free_.find(keyOf(cls));
      if (it != free_.end() && !it->second.empty()) {
        void *p = it->second.back();
        it->second.pop_back();
        kept_ -= cls;
        return p;
      }
    }
    void *p = nullptr;
    if (qsx_device_alloc(cls, &p) != QSX_OK) {
      trim();                                    // what the pool keeps goes back to the device before giving up
      DeviceBuffer::trimThisThread();
      (void)qsx_trim_scratch(nullptr);           // and what libqsx.so keeps for this thread between calls
      CheckStatus(qsx_device_alloc(cls, &p), "qsx_device_alloc(block)");
    }
    {
      std::lock_guard<std::mutex> lock(mutex_);
      owner_[p] = keyOf(cls).first;              // the slab's device: where it may be handed out again
    }
    return p;
  }
  void trim() {
    std::lock_guard<std::mutex> lock(mutex_);
    for (auto &cls : free_) {
      for (void *q : cls.second) {
        owner_.erase(q);
        qsx_device_free(q);
      }
      cls.second.clear();
    }
    kept_ = 0;
  }
  void give(void *p, std::size_t granted) {
    if (p == nullptr) return;
    if (granted != 0) {
      std::lock_guard<std::mutex> lock(mutex_);
      auto owner = owner_.find(p);
      if (owner != owner_.end() && kept_ + granted <= kKeepBytes) {
        free_[std::make_pair(owner->second, granted)].push_back(p);     // filed under the device it was made on
        kept_ += granted;
        return;
      }
      if (owner != owner_.end()) owner_.erase(owner);
    }
    qsx_device_free(p);
  }

 private:
  std::mutex mutex_;
  // slabs by (device, size class): the pool is process-wide, Worker pools are per device
  static std::pair<int, std::size_t> keyOf(std::size_t cls) {
    int device = 0;
    (void)qsx_current_device(&device);
    return std::make_pair(device, cls);
  }
  std::map<std::pair<int, std::size_t>, std::vector<void *>> free_;
  std::map<void *, int> owner_;                  // every pooled slab, out or in: its device
  std::size_t kept_ = 0;
};

void TrimBlockSlabPool() { BlockSlabPool::instance().trim(); }
void *TakePooled(std::size_t bytes, std::size_t *granted) { return BlockSlabPool::instance().take(bytes, granted); }
void GivePooled(void *p, std::size_t granted) { BlockSlabPool::instance().give(p, granted); }

// qsx_set_out_of_memory_hook: a device allocation inside libqsx.so (a join table, an aggregation state, a scratch arena)
// found no memory — the pooled output-block allocations and the failing thread's scratch cache go back before its retry.
void HostOutOfMemoryHook(void *) {
  TrimBlockSlabPool();
  DeviceBuffer::trimThisThread();
}
struct RegisterOutOfMemoryHook {
  RegisterOutOfMemoryHook() { (void)qsx_set_out_of_memory_hook(&HostOutOfMemoryHook, nullptr); }
} g_register_out_of_memory_hook;

std::int64_t ReadCount(const void *dev_count) {
  std::int64_t v = 0;
  CheckStatus(qsx_copy_to_host(&v, dev_count, sizeof(v), CurrentStream()), "qsx_copy_to_host");
  return v;
}

std::uint64_t NowMicros() {
  return static_cast<std::uint64_t>(std::chrono::duration_cast<std::chrono::microseconds>(
                                        std::chrono::steady_clock::now().time_since_epoch()).count());
}
}  // namespace host_internal

qsx_stream_t CurrentStream() { return host_internal::ThreadStream(); }
void SetCurrentStream(qsx_stream_t stream) { host_internal::SetThreadStream(stream); }

}  // namespace quickstep
