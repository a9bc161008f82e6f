// runtime.hip — status strings, device discovery and the memory plumbing of
// the C ABI (include/qsx.h).  No compute here.
#include <cstdlib>
#include "common.hpp"

#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace qsx {

static thread_local std::string g_last_error;

void set_last_error_text(const char *text) { g_last_error = text; }

void set_last_error(const char *what, hipError_t err) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(err);
}

static int probe_devices() {
  int count = 0;
  hipError_t err = hipGetDeviceCount(&count);
  if (err != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int usable = 0;
  for (int d = 0; d < count; ++d) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) != hipSuccess) continue;
    // This library carries gfx950 code objects only.
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++usable;
  }
  return usable;
}

static int usable_devices() {
  static std::once_flag once;
  static int usable = 0;
  std::call_once(once, []() { usable = probe_devices(); });
  return usable;
}

// ---- the per-(host thread, device, stream) resources of common.hpp -----------------------------------------------------------
namespace {
// Calls follow the calling thread's device (qsx_set_current_device), and the null stream names a different queue on every
// device: the key of a thread's buffers is (device, stream), never the stream alone.
int current_device() {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) (void)hipGetLastError();
  return device;
}
// Makes `device` current for a scope (buffers are freed, and their streams waited for, on the device that owns them).
struct DeviceScope {
  int previous;
  bool switched;
  explicit DeviceScope(int device) : previous(current_device()), switched(false) {
    if (device != previous && hipSetDevice(device) == hipSuccess) switched = true;
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(previous);
  }
};
using StreamKey = std::pair<int, hipStream_t>;

struct ThreadResources {
  std::map<StreamKey, ScratchArena> arenas;
  std::map<StreamKey, StagedBuffer> staged;
  std::map<std::pair<StreamKey, const void *>, void *> slots;

  static size_t drop(ScratchArena &a) {
    const size_t bytes = a.capacity;
    if (a.base != nullptr) (void)device_free(a.base);
    a = ScratchArena();
    return bytes;
  }
  static size_t drop(StagedBuffer &b) {
    const size_t bytes = b.capacity;
    if (b.copied != nullptr) {
      (void)hipEventSynchronize(b.copied);
      (void)hipEventDestroy(b.copied);
    }
    if (b.device != nullptr) (void)device_free(b.device);
    if (b.pinned != nullptr) (void)hipHostFree(b.pinned);
    b = StagedBuffer();
    return bytes;
  }
  // Which entries go: those of `key` only, all of them, or all BUT those of `key`.  The stream's queued work still uses them:
  // wait for it first, on its own device.
  enum Which { kOnly, kAll, kAllBut };
  static bool chosen(const StreamKey &entry, const StreamKey &key, Which which) {
    return which == kAll || (which == kOnly ? entry == key : entry != key);
  }
  size_t release(const StreamKey &key, Which which) {
    size_t bytes = 0;
    for (auto it = arenas.begin(); it != arenas.end();) {
      if (chosen(it->first, key, which)) {
        DeviceScope scope(it->first.first);
        (void)hipStreamSynchronize(it->first.second);
        bytes += drop(it->second);
        it = arenas.erase(it);
      } else {
        ++it;
      }
    }
    for (auto it = staged.begin(); it != staged.end();) {
      if (chosen(it->first, key, which)) {
        DeviceScope scope(it->first.first);
        (void)hipStreamSynchronize(it->first.second);
        bytes += drop(it->second);
        it = staged.erase(it);
      } else {
        ++it;
      }
    }
    for (auto it = slots.begin(); it != slots.end();) {
      if (chosen(it->first.first, key, which)) {
        DeviceScope scope(it->first.first.first);
        (void)hipStreamSynchronize(it->first.first.second);
        if (it->second != nullptr) (void)device_free(it->second);
        it = slots.erase(it);
      } else {
        ++it;
      }
    }
    (void)hipGetLastError();   // (a stream destroyed behind the library's back: its synchronize fails, the buffers still go)
    return bytes;
  }
  ~ThreadResources() { (void)release(StreamKey(), kAll); }   // thread exit
};
ThreadResources &thread_resources() {
  thread_local ThreadResources r;
  return r;
}
StreamKey key_of(hipStream_t stream) { return StreamKey(current_device(), stream); }
}  // namespace

ScratchArena &thread_scratch_arena(hipStream_t stream) { return thread_resources().arenas[key_of(stream)]; }
StagedBuffer &thread_staged_buffer(hipStream_t stream) { return thread_resources().staged[key_of(stream)]; }
void *&thread_device_slot(hipStream_t stream, const void *type_tag) { return thread_resources().slots[std::make_pair(key_of(stream), type_tag)]; }
void release_thread_stream(hipStream_t stream) { (void)thread_resources().release(key_of(stream), ThreadResources::kOnly); }
size_t trim_thread_resources() { return thread_resources().release(StreamKey(), ThreadResources::kAll); }
size_t trim_thread_resources_sparing(hipStream_t stream) { return thread_resources().release(key_of(stream), ThreadResources::kAllBut); }

namespace {
std::mutex g_hook_mutex;
void (*g_oom_hook)(void *) = nullptr;
void *g_oom_hook_user = nullptr;
}  // namespace

namespace {
// (16 GiB of a 288 GB device: a call's scratch beyond what its arena keeps — the partition passes over 100 M rows take 2.5 GB —
// comes from here too, and a hipMalloc / hipFree pair of that size costs anything between 0.2 and 140 ms, by the state the
// driver's address space is in: tools/agg_large_groups.py saw both within one process)
constexpr size_t kIdleKeepBytes = size_t(16) << 30;
struct Owned {
  size_t bytes;
  int device;
};
struct Allocations {
  std::mutex mutex;
  std::map<void *, Owned> live;                                          // every live device_malloc: its size and its device
  std::map<std::pair<int, size_t>, std::vector<void *>> idle;             // given back by device_free_idle, by (device, exact size)
  size_t idle_bytes = 0;
};
Allocations &allocations() {
  static Allocations *a = new Allocations;   // never destroyed: frees may arrive during process teardown
  return *a;
}
}  // namespace

size_t trim_idle_allocations() {
  Allocations &a = allocations();
  std::vector<void *> doomed;
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> lock(a.mutex);
    for (auto &cls : a.idle) {
      for (void *p : cls.second) doomed.push_back(p);
    }
    a.idle.clear();
    bytes = a.idle_bytes;
    a.idle_bytes = 0;
  }
  for (void *p : doomed) (void)hipFree(p);   // (hipFree finds the owning device from the pointer)
  return bytes;
}

int device_of_allocation(const void *ptr) {
  Allocations &a = allocations();
  std::lock_guard<std::mutex> lock(a.mutex);
  auto it = a.live.find(const_cast<void *>(ptr));
  return it != a.live.end() ? it->second.device : -1;
}

hipError_t synchronize_owner_device(const void *ptr) {
  const int owner = ptr != nullptr ? device_of_allocation(ptr) : -1;
  if (owner < 0) return hipDeviceSynchronize();
  DeviceScope scope(owner);
  return hipDeviceSynchronize();
}

hipError_t device_free(void *ptr) {
  if (ptr == nullptr) return hipSuccess;
  {
    Allocations &a = allocations();
    std::lock_guard<std::mutex> lock(a.mutex);
    a.live.erase(ptr);
  }
  return hipFree(ptr);
}

hipError_t device_free_idle(void *ptr) {
  if (ptr == nullptr) return hipSuccess;
  std::vector<void *> evicted;
  bool kept = false;
  {
    Allocations &a = allocations();
    std::lock_guard<std::mutex> lock(a.mutex);
    auto it = a.live.find(ptr);
    if (it != a.live.end() && it->second.bytes <= kIdleKeepBytes) {
      // room for the newcomer: idle allocations go back to the runtime first, the largest classes first — a size nobody asks
      // for again must not keep the pool full for good
      while (a.idle_bytes + it->second.bytes > kIdleKeepBytes && !a.idle.empty()) {
        auto cls = std::prev(a.idle.end());
        if (!cls->second.empty()) {
          evicted.push_back(cls->second.back());
          cls->second.pop_back();
          a.idle_bytes -= cls->first.second;
        }
        if (cls->second.empty()) a.idle.erase(cls);
      }
      a.idle[std::make_pair(it->second.device, it->second.bytes)].push_back(ptr);
      a.idle_bytes += it->second.bytes;
      kept = true;
    }
    if (it != a.live.end()) a.live.erase(it);
  }
  for (void *p : evicted) (void)hipFree(p);
  return kept ? hipSuccess : hipFree(ptr);
}

hipError_t device_malloc(void **ptr, size_t bytes) {
  Allocations &a = allocations();
  const int device = current_device();     // an idle allocation is only ever handed back out on the device it lives on
  {
    std::lock_guard<std::mutex> lock(a.mutex);
    auto it = a.idle.find(std::make_pair(device, bytes));
    if (it != a.idle.end() && !it->second.empty()) {
      *ptr = it->second.back();
      it->second.pop_back();
      a.idle_bytes -= bytes;
      a.live[*ptr] = Owned{bytes, device};
      return hipSuccess;
    }
  }
  hipError_t err = hipMalloc(ptr, bytes);
  if (err == hipErrorOutOfMemory && trim_idle_allocations() != 0) {
    (void)hipGetLastError();
    err = hipMalloc(ptr, bytes);
  }
  if (err == hipSuccess) {
    std::lock_guard<std::mutex> lock(a.mutex);
    a.live[*ptr] = Owned{bytes, device};
    return err;
  }
  if (err != hipErrorOutOfMemory) return err;
  (void)hipGetLastError();
  void (*hook)(void *) = nullptr;
  void *user = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_hook_mutex);
    hook = g_oom_hook;
    user = g_oom_hook_user;
  }
  if (hook == nullptr) return err;
  hook(user);
  err = hipMalloc(ptr, bytes);
  if (err == hipSuccess) {
    std::lock_guard<std::mutex> lock(a.mutex);
    a.live[*ptr] = Owned{bytes, device};
  }
  return err;
}

int device_ready() {
  if (usable_devices() > 0) return QSX_OK;
  g_last_error = "no gfx950 (MI355X) device is visible to HIP; the execution kernel has no CPU path";
  return QSX_ERR_NO_DEVICE;
}

}  // namespace qsx

namespace qsx {
// qsx_copy_segments: blockIdx.y = segment, blockIdx.x strides over its chunks of kCopyChunk units.
constexpr int kCopyChunk = 4096;
typedef unsigned int CopyUnit16 __attribute__((ext_vector_type(4)));
template <typename V>
__global__ __launch_bounds__(256) void copy_segments_kernel(const long long *__restrict__ segments) {
  const long long *seg = segments + 3 * static_cast<size_t>(blockIdx.y);
  const V *src = as_global(reinterpret_cast<const V *>(seg[0]));
  V *dst = as_global(reinterpret_cast<V *>(seg[1]));
  const long long n = seg[2];
  for (long long base = static_cast<long long>(blockIdx.x) * kCopyChunk; base < n; base += static_cast<long long>(gridDim.x) * kCopyChunk) {
    V v[kCopyChunk / 256];
#pragma unroll
    for (int i = 0; i < kCopyChunk / 256; ++i) {
      const long long at = base + i * 256 + threadIdx.x;
      v[i] = load_global(&src[at < n ? at : n - 1]);
    }
#pragma unroll
    for (int i = 0; i < kCopyChunk / 256; ++i) {
      const long long at = base + i * 256 + threadIdx.x;
      if (at < n) store_global(v[i], &dst[at]);
    }
  }
}
}  // namespace qsx

extern "C" {

const char *qsx_status_string(int status) {
  switch (status) {
    case QSX_OK: return "ok";
    case QSX_ERR_INVALID_ARGUMENT: return "invalid argument";
    case QSX_ERR_NO_DEVICE: return "no gfx950 device available (no CPU fallback exists)";
    case QSX_ERR_OUT_OF_MEMORY: return "out of device memory";
    case QSX_ERR_HIP: return "HIP runtime error (see qsx_last_error)";
    case QSX_ERR_CAPACITY: return "caller-provided capacity too small";
    case QSX_ERR_UNSUPPORTED: return "unsupported type / configuration";
    case QSX_ERR_TOO_MANY_GROUPS: return "aggregation table overflow: more groups than the state can hold";
    case QSX_ERR_HASH_COLLISION: return "wide group-by key: two keys shared a 64-bit hash, the result is void (re-run the operator)";
    case QSX_ERR_COMM: return "RCCL unavailable or a collective failed (see qsx_last_error)";
    default: return "unknown status";
  }
}

int qsx_abi_version(void) { return QSX_ABI_VERSION; }

size_t qsx_abi_sizeof_agg_config(void) { return sizeof(qsx_agg_config_t); }

int qsx_device_count(void) { return qsx::usable_devices(); }

int qsx_current_device(int *out_device) {
  QSX_REQUIRE_DEVICE();
  if (out_device == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  QSX_HIP_TRY(hipGetDevice(out_device));
  return QSX_OK;
}

int qsx_set_current_device(int device) {
  QSX_REQUIRE_DEVICE();
  int count = 0;
  QSX_HIP_TRY(hipGetDeviceCount(&count));
  if (device < 0 || device >= count) return QSX_ERR_INVALID_ARGUMENT;
  QSX_HIP_TRY(hipSetDevice(device));
  return QSX_OK;
}

const char *qsx_last_error(void) { return qsx::g_last_error.c_str(); }

int qsx_device_alloc(size_t bytes, void **out_dev) {
  QSX_REQUIRE_DEVICE();
  if (out_dev == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  *out_dev = nullptr;
  if (bytes == 0) return QSX_OK;
  QSX_HIP_TRY(qsx::device_malloc(out_dev, bytes));
  return QSX_OK;
}

int qsx_set_out_of_memory_hook(void (*hook)(void *user), void *user) {
  std::lock_guard<std::mutex> lock(qsx::g_hook_mutex);
  qsx::g_oom_hook = hook;
  qsx::g_oom_hook_user = user;
  return QSX_OK;
}

int qsx_device_free(void *dev) {
  if (dev == nullptr) return QSX_OK;
  QSX_REQUIRE_DEVICE();
  QSX_HIP_TRY(qsx::device_free(dev));
  return QSX_OK;
}

int qsx_copy_to_device(void *dst_dev, const void *src_host, size_t bytes, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (bytes == 0) return QSX_OK;
  QSX_HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, qsx::as_stream(stream)));
  return QSX_OK;
}

int qsx_copy_on_device(void *dst_dev, const void *src_dev, size_t bytes, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (bytes == 0) return QSX_OK;
  QSX_HIP_TRY(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, qsx::as_stream(stream)));
  return QSX_OK;
}

int qsx_copy_segments(int64_t num_segments, const void *const *src_dev, void *const *dst_dev, const int64_t *bytes, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (num_segments < 0 || (num_segments > 0 && (src_dev == nullptr || dst_dev == nullptr || bytes == nullptr))) return QSX_ERR_INVALID_ARGUMENT;
  hipStream_t s = qsx::as_stream(stream);
  // a segment goes in the widest unit its two addresses and its length are multiples of — 16, 8, 4, 2 bytes or 1 (a stripe of
  // an adopted block image lies at any byte address; a stripe of 349 525 INTs is no multiple of 16 bytes long): byte by byte
  // costs 2.4 x the 16-byte path.  {source, destination, units} per segment, grouped by unit.
  std::vector<long long> groups[5];
  long long widest[5] = {0, 0, 0, 0, 0};
  for (int64_t i = 0; i < num_segments; ++i) {
    if (bytes[i] < 0) return QSX_ERR_INVALID_ARGUMENT;
    if (bytes[i] == 0) continue;
    if (src_dev[i] == nullptr || dst_dev[i] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
    const uintptr_t a = reinterpret_cast<uintptr_t>(src_dev[i]), b = reinterpret_cast<uintptr_t>(dst_dev[i]);
    const uintptr_t all = a | b | static_cast<uintptr_t>(bytes[i]);
    const int g = (all & 15u) == 0 ? 0 : ((all & 7u) == 0 ? 1 : ((all & 3u) == 0 ? 2 : ((all & 1u) == 0 ? 3 : 4)));
    const long long units = bytes[i] >> (4 - g);
    groups[g].push_back(static_cast<long long>(a));
    groups[g].push_back(static_cast<long long>(b));
    groups[g].push_back(units);
    widest[g] = std::max(widest[g], units);
  }
  std::vector<long long> table;
  size_t first_word[6] = {0, 0, 0, 0, 0, 0};
  for (int g = 0; g < 5; ++g) {
    first_word[g] = table.size();
    table.insert(table.end(), groups[g].begin(), groups[g].end());
  }
  first_word[5] = table.size();
  if (table.empty()) return QSX_OK;
  const size_t table_bytes = table.size() * sizeof(long long);
  const long long *dev_table = static_cast<const long long *>(qsx::staged_device_buffer(s, table_bytes));
  if (dev_table == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  const int rc = qsx::staged_upload(s, table.data(), table_bytes);
  if (rc != QSX_OK) return rc;
  for (int g = 0; g < 5; ++g) {
    const long long count = static_cast<long long>(first_word[g + 1] - first_word[g]) / 3;
    if (count == 0) continue;
    const long long *first = dev_table + first_word[g];
    const unsigned chunks = static_cast<unsigned>(std::min<long long>((widest[g] + qsx::kCopyChunk - 1) / qsx::kCopyChunk, 256));
    for (long long at = 0; at < count; at += 65535) {
      const dim3 grid(chunks, static_cast<unsigned>(std::min<long long>(count - at, 65535)));
      switch (g) {
        case 0: hipLaunchKernelGGL(qsx::copy_segments_kernel<qsx::CopyUnit16>, grid, dim3(256), 0, s, first + 3 * at); break;
        case 1: hipLaunchKernelGGL(qsx::copy_segments_kernel<unsigned long long>, grid, dim3(256), 0, s, first + 3 * at); break;
        case 2: hipLaunchKernelGGL(qsx::copy_segments_kernel<unsigned int>, grid, dim3(256), 0, s, first + 3 * at); break;
        case 3: hipLaunchKernelGGL(qsx::copy_segments_kernel<unsigned short>, grid, dim3(256), 0, s, first + 3 * at); break;
        default: hipLaunchKernelGGL(qsx::copy_segments_kernel<unsigned char>, grid, dim3(256), 0, s, first + 3 * at); break;
      }
      QSX_CHECK_LAUNCH();
    }
  }
  return QSX_OK;
}

int qsx_copy_to_host(void *dst_host, const void *src_dev, size_t bytes, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (bytes == 0) return QSX_OK;
  QSX_HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, qsx::as_stream(stream)));
  QSX_HIP_TRY(hipStreamSynchronize(qsx::as_stream(stream)));
  return QSX_OK;
}

int qsx_memset_device(void *dst_dev, int byte, size_t bytes, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (bytes == 0) return QSX_OK;
  QSX_HIP_TRY(hipMemsetAsync(dst_dev, byte, bytes, qsx::as_stream(stream)));
  return QSX_OK;
}

int qsx_stream_create(qsx_stream_t *out_stream) {
  QSX_REQUIRE_DEVICE();
  if (out_stream == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  hipStream_t s;
  QSX_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *out_stream = reinterpret_cast<qsx_stream_t>(s);
  return QSX_OK;
}

int qsx_stream_destroy(qsx_stream_t stream) {
  if (stream == nullptr) return QSX_OK;
  QSX_REQUIRE_DEVICE();
  qsx::release_thread_stream(qsx::as_stream(stream));   // the scratch arena, staging buffers and slots this thread kept for it
  QSX_HIP_TRY(hipStreamDestroy(qsx::as_stream(stream)));
  return QSX_OK;
}

int qsx_trim_scratch(size_t *out_bytes_released) {
  QSX_REQUIRE_DEVICE();
  const size_t bytes = qsx::trim_thread_resources() + qsx::trim_idle_allocations();
  if (out_bytes_released != nullptr) *out_bytes_released = bytes;
  return QSX_OK;
}

int qsx_stream_synchronize(qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  QSX_HIP_TRY(hipStreamSynchronize(qsx::as_stream(stream)));
  return QSX_OK;
}

}  // extern "C"
