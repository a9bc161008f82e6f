// quickstep_gpu.cpp — see quickstep_gpu.hpp.  Every GPU work order body is a
// short sequence of C-ABI calls (include/qsx.h); INTEGRATION.md lists the same
// sequences against the reference's own classes.
#include "quickstep_gpu.hpp"

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>

namespace quickstep {

// ---------------------------------------------------------------------------
// errors / stream
// ---------------------------------------------------------------------------
ExecutionError::ExecutionError(const std::string &where, int status)
    : std::runtime_error(where + ": " + qsx_status_string(status) + " [" + qsx_last_error() + "]"), status_(status) {}

void CheckStatus(int status, const char *where) {
  if (status != QSX_OK) throw ExecutionError(where, status);
}

namespace {
thread_local qsx_stream_t tls_stream = nullptr;

// Scratch owned by one work order.  A device allocation of a few MB costs ~170 us (hipMalloc + hipFree, whatever the
// size; 0.7 us up to 64 KiB: tools/ubench/alloc_cost.hip) — more than the kernels of a work order over a 4 MB block take —
// and the stream-ordered pool is not an option on this stack (csrc/common.hpp, CallScratch).  So buffers above 64 KiB come
// from a per-thread cache of plain allocations in power-of-two size classes: a worker thread issues all its work on one
// stream, so a buffer handed back while its last kernel is still queued can be handed out again to the same thread —
// the next use is ordered behind it.  (A buffer built by one thread and consumed by another — DISTINCT chunks — is
// published after a stream synchronisation, like a storage block.)  The cache keeps at most kCacheBytes per thread.
void TrimBlockSlabPool();   // (defined behind BlockSlabPool)
void *TakePooled(std::size_t bytes, std::size_t *granted);
void GivePooled(void *p, std::size_t granted);
struct DeviceBuffer {
  static constexpr std::size_t kCacheFrom = 64 * 1024 + 1;
  static constexpr std::size_t kCacheBytes = std::size_t(2) << 30;
  struct Cache {
    std::map<std::size_t, std::vector<void *>> free_by_class;
    std::size_t bytes = 0;
    ~Cache() {
      for (auto &cls : free_by_class) for (void *p : cls.second) qsx_device_free(p);
    }
  };
  static Cache &cache() {
    thread_local Cache c;
    return c;
  }
  static void trimThisThread() {
    Cache &c = cache();
    for (auto &cls : c.free_by_class) {
      for (void *q : cls.second) qsx_device_free(q);
      cls.second.clear();
    }
    c.bytes = 0;
  }
  void *ptr = nullptr;
  std::size_t size_class = 0;   // 0: a plain allocation of its own
  std::size_t pooled = 0;       // != 0: from the shared pool (its size class there)
  static constexpr std::size_t kSharedFrom = std::size_t(32) << 20;
  static bool cacheEnabled() {   // QSX_HOST_SCRATCH_CACHE=0: every buffer a plain allocation (debugging)
    static const bool on = []() {
      const char *e = std::getenv("QSX_HOST_SCRATCH_CACHE");
      return e == nullptr || e[0] != '0';
    }();
    return on;
  }
  explicit DeviceBuffer(std::size_t bytes) {
    if (bytes >= kSharedFrom && cacheEnabled()) {
      // pair lists / operand columns of a run of blocks: hundreds of MB — from the process-wide pool of the output blocks,
      // not one cached copy per Worker thread (whichever thread happens to pick the join next would allocate its own)
      ptr = TakePooled(bytes, &pooled);
      return;
    }
    if (bytes >= kCacheFrom && cacheEnabled()) {
      size_class = 128 * 1024;
      while (size_class < bytes) size_class *= 2;
      Cache &c = cache();
      auto it = c.free_by_class.find(size_class);
      if (it != c.free_by_class.end() && !it->second.empty()) {
        ptr = it->second.back();
        it->second.pop_back();
        c.bytes -= size_class;
        return;
      }
      if (qsx_device_alloc(size_class, &ptr) != QSX_OK) {
        // this thread's cached buffers, the pooled block allocations and what libqsx.so keeps for this thread go back first
        // (the library has already called HostOutOfMemoryHook once from inside qsx_device_alloc)
        trimThisThread();
        TrimBlockSlabPool();
        (void)qsx_trim_scratch(nullptr);
        CheckStatus(qsx_device_alloc(size_class, &ptr), "qsx_device_alloc");
      }
      return;
    }
    CheckStatus(qsx_device_alloc(bytes ? bytes : 8, &ptr), "qsx_device_alloc");
  }
  ~DeviceBuffer() {
    if (pooled != 0) {
      (void)qsx_stream_synchronize(CurrentStream());   // another thread may take it next: this thread's queued work first
      GivePooled(ptr, pooled);
      return;
    }
    if (size_class != 0) {
      Cache &c = cache();
      if (c.bytes + size_class <= kCacheBytes) {
        c.free_by_class[size_class].push_back(ptr);
        c.bytes += size_class;
        return;
      }
    }
    qsx_device_free(ptr);
  }
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
};

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
      auto it = free_.find(cls);
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
    return p;
  }
  void trim() {
    std::lock_guard<std::mutex> lock(mutex_);
    for (auto &cls : free_) {
      for (void *q : cls.second) qsx_device_free(q);
      cls.second.clear();
    }
    kept_ = 0;
  }
  void give(void *p, std::size_t granted) {
    if (p == nullptr) return;
    if (granted != 0) {
      std::lock_guard<std::mutex> lock(mutex_);
      if (kept_ + granted <= kKeepBytes) {
        free_[granted].push_back(p);
        kept_ += granted;
        return;
      }
    }
    qsx_device_free(p);
  }

 private:
  std::mutex mutex_;
  std::map<std::size_t, std::vector<void *>> free_;
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
}  // namespace

void CheckRepartition(const char *op, bool has_repartition, const InsertDestination *dest);   // (behind QueryContext)

qsx_stream_t CurrentStream() { return tls_stream; }
void SetCurrentStream(qsx_stream_t stream) { tls_stream = stream; }

// ---------------------------------------------------------------------------
// catalog + storage
// ---------------------------------------------------------------------------
attribute_id CatalogRelation::addAttribute(const std::string &name, Type type) {
  names_.push_back(name);
  types_.push_back(type);
  return static_cast<attribute_id>(types_.size() - 1);
}
attribute_id CatalogRelation::getAttributeByName(const std::string &name) const {
  for (std::size_t i = 0; i < names_.size(); ++i) {
    if (names_[i] == name) return static_cast<attribute_id>(i);
  }
  return kInvalidAttributeID;
}
void CatalogRelation::addBlock(block_id b) {
  std::lock_guard<std::mutex> lock(mutex_);
  blocks_.push_back(b);
  if (num_partitions_ > 0) partition_blocks_.at(0).push_back(b);
}
std::vector<block_id> CatalogRelation::getBlocksSnapshot() const {
  std::lock_guard<std::mutex> lock(mutex_);
  return blocks_;
}
void CatalogRelation::setPartitionScheme(std::size_t num_partitions, attribute_id partition_attribute) {
  std::lock_guard<std::mutex> lock(mutex_);
  num_partitions_ = num_partitions;
  partition_attribute_ = partition_attribute;
  partition_blocks_.assign(num_partitions, {});
}
void CatalogRelation::addBlockToPartition(block_id b, partition_id part) {
  std::lock_guard<std::mutex> lock(mutex_);
  blocks_.push_back(b);
  if (num_partitions_ > 0) partition_blocks_.at(part).push_back(b);
}
std::vector<block_id> CatalogRelation::getBlocksInPartition(partition_id part) const {
  std::lock_guard<std::mutex> lock(mutex_);
  if (num_partitions_ == 0) return part == 0 ? blocks_ : std::vector<block_id>();
  return partition_blocks_.at(part);
}

namespace {
bool g_host_memory = false;  // CPU plumbing mode (BASELINE config 1): blocks live in host memory
}
void UseHostMemoryForBlocks(bool on) { g_host_memory = on; }

StorageBlock::StorageBlock(const CatalogRelation &relation, std::int64_t capacity, std::int64_t first_row, bool one_allocation)
    : relation_(relation), capacity_(capacity), num_tuples_(0), first_row_(first_row) {
  if (one_allocation && !g_host_memory && std::getenv("QSX_HOST_BLOCK_SLAB_OFF") == nullptr) {
    // an output block: all stripes and null bitmaps in one allocation (a device allocation of a few MB costs ~170 us
    // whatever its size, and a block has one per attribute otherwise)
    auto round_up = [](std::size_t v) { return (v + 255) / 256 * 256; };
    std::size_t total = 0, null_bytes = 0;
    for (std::size_t a = 0; a < relation.size(); ++a) {
      const Type &t = relation.getAttributeType(static_cast<attribute_id>(a));
      total += round_up(static_cast<std::size_t>(capacity) * t.width + 8);
      if (t.nullable) null_bytes += round_up(static_cast<std::size_t>((capacity + 63) / 64) * 8 + 8);
    }
    slab_bytes_ = total + null_bytes;
    slab_ = BlockSlabPool::instance().take(slab_bytes_, &slab_granted_);
    char *at = static_cast<char *>(slab_);
    char *nulls_at = at + total;
    if (null_bytes != 0) CheckStatus(qsx_memset_device(nulls_at, 0, null_bytes, CurrentStream()), "qsx_memset_device(null bitmaps)");
    for (std::size_t a = 0; a < relation.size(); ++a) {
      const Type &t = relation.getAttributeType(static_cast<attribute_id>(a));
      stripes_.push_back(at);
      at += round_up(static_cast<std::size_t>(capacity) * t.width + 8);
      void *nulls = nullptr;
      if (t.nullable) {
        nulls = nulls_at;
        nulls_at += round_up(static_cast<std::size_t>((capacity + 63) / 64) * 8 + 8);
      }
      null_bitmaps_.push_back(nulls);
    }
    return;
  }
  for (std::size_t a = 0; a < relation.size(); ++a) {
    void *p = nullptr;
    const std::size_t bytes = static_cast<std::size_t>(capacity) * relation.getAttributeType(static_cast<attribute_id>(a)).width;
    if (g_host_memory) {
      p = std::malloc(bytes ? bytes : 8);
    } else {
      CheckStatus(qsx_device_alloc(bytes ? bytes : 8, &p), "qsx_device_alloc(stripe)");
    }
    stripes_.push_back(p);
    void *nulls = nullptr;
    if (relation.getAttributeType(static_cast<attribute_id>(a)).nullable) {
      const std::size_t nbytes = static_cast<std::size_t>((capacity + 63) / 64) * 8 + 8;
      if (g_host_memory) {
        nulls = std::calloc(nbytes, 1);
      } else {
        CheckStatus(qsx_device_alloc(nbytes, &nulls), "qsx_device_alloc(null bitmap)");
        CheckStatus(qsx_memset_device(nulls, 0, nbytes, CurrentStream()), "qsx_memset_device(null bitmap)");
      }
    }
    null_bitmaps_.push_back(nulls);
  }
}
StorageBlock::StorageBlock(std::shared_ptr<StorageBlock> parent, std::int64_t first_tuple, std::int64_t num_tuples)
    : relation_(parent->relation_), capacity_(num_tuples), num_tuples_(num_tuples), first_row_(0), view_parent_(std::move(parent)) {
  for (std::size_t a = 0; a < relation_.size(); ++a) {
    const Type &t = relation_.getAttributeType(static_cast<attribute_id>(a));
    stripes_.push_back(static_cast<char *>(view_parent_->stripe(static_cast<attribute_id>(a))) + first_tuple * t.width);
    void *nulls = nullptr;
    if (t.nullable) {
      const std::size_t nbytes = static_cast<std::size_t>((num_tuples + 63) / 64) * 8 + 8;
      if (g_host_memory) {
        nulls = std::calloc(nbytes, 1);
      } else {
        CheckStatus(qsx_device_alloc(nbytes, &nulls), "qsx_device_alloc(null bitmap)");
        CheckStatus(qsx_memset_device(nulls, 0, nbytes, CurrentStream()), "qsx_memset_device(null bitmap)");
      }
    }
    null_bitmaps_.push_back(nulls);
  }
}
StorageBlock::StorageBlock(const CatalogRelation &relation, std::int64_t num_tuples, const std::vector<void *> &stripes,
                           const std::vector<void *> &null_bitmaps)
    : relation_(relation), capacity_(num_tuples), num_tuples_(num_tuples), first_row_(0), external_memory_(true),
      stripes_(stripes), null_bitmaps_(null_bitmaps) {}

StorageBlock::~StorageBlock() {
  if (external_memory_) {
    for (void *&p : stripes_) p = nullptr;
    for (void *&p : null_bitmaps_) p = nullptr;
  }
  // A block that dies while an exception unwinds a work order may still be the target of kernels that work order has
  // queued: they must finish before its memory goes back to a pool another Worker takes from (the regular path has
  // synchronised its stream before the last reference goes).
  if (std::uncaught_exceptions() > 0 && !g_host_memory) (void)qsx_stream_synchronize(CurrentStream());
  if (view_parent_ != nullptr) {
    for (void *&p : stripes_) p = nullptr;   // the parent's
  }
  if (slab_ != nullptr) {
    // (a stripe outside the slab was materialised later, stripe(): freed on its own below)
    auto in_slab = [&](void *p) { return p >= slab_ && p < static_cast<char *>(slab_) + slab_bytes_; };
    for (void *&p : stripes_) if (in_slab(p)) p = nullptr;
    for (void *&p : null_bitmaps_) if (in_slab(p)) p = nullptr;
    BlockSlabPool::instance().give(slab_, slab_granted_);
  }
  for (void *p : stripes_) {
    if (g_host_memory) std::free(p); else qsx_device_free(p);
  }
  for (void *p : null_bitmaps_) {
    if (p == nullptr) continue;
    if (g_host_memory) std::free(p); else qsx_device_free(p);
  }
  for (CompressedAttribute &c : compressed_) {
    qsx_device_free(c.codes);
    qsx_device_free(c.dictionary);
  }
}
void StorageBlock::copyNullBitmapToHost(attribute_id a, std::uint64_t *dst) const {
  const std::size_t bytes = static_cast<std::size_t>((num_tuples_ + 63) / 64) * 8;
  if (null_bitmaps_.at(a) == nullptr) {
    std::memset(dst, 0, bytes);
  } else if (g_host_memory) {
    std::memcpy(dst, null_bitmaps_.at(a), bytes);
  } else {
    CheckStatus(qsx_copy_to_host(dst, null_bitmaps_.at(a), bytes, CurrentStream()), "qsx_copy_to_host(null bitmap)");
  }
}
void StorageBlock::copyAttributeToHost(attribute_id a, void *dst) const {
  const std::size_t bytes = static_cast<std::size_t>(num_tuples_) * relation_.getAttributeType(a).width;
  if (g_host_memory) {
    std::memcpy(dst, stripes_.at(a), bytes);
  } else {
    CheckStatus(qsx_copy_to_host(dst, stripe(a), bytes, CurrentStream()), "qsx_copy_to_host");
  }
}

void *StorageBlock::stripe(attribute_id a) const {
  if (compressed_.empty() || compressed_.at(a).kind == CompressedAttribute::kUncompressed) return stripes_.at(a);
  std::lock_guard<std::mutex> lock(decode_mutex_);
  if (stripes_.at(a) == nullptr) {
    const CompressedAttribute &c = compressed_.at(a);
    const int width = relation_.getAttributeType(a).width;
    void *values = nullptr;
    CheckStatus(qsx_device_alloc(static_cast<std::size_t>(capacity_ ? capacity_ : 1) * width, &values), "qsx_device_alloc(decoded stripe)");
    CheckStatus(qsx_decode_codes(c.code_width, c.codes, num_tuples_, c.dictionary, width, values, CurrentStream()), "qsx_decode_codes");
    CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");   // other workers' streams read it next
    stripes_.at(a) = values;
  }
  return stripes_.at(a);
}

namespace {
// CompressedBlockBuilder's per-attribute decision for fixed-width numeric attributes without NULLs
// (storage/CompressedBlockBuilder.cpp:508-566 truncated width, :590-650 truncation vs dictionary;
// compression/CompressionDictionaryBuilder.{hpp:87-110,cpp:128-134} code length and dictionary size).
template <typename T>
void BuildCompressedAttribute(TypeID type, const T *values, std::int64_t n, CompressedAttribute *out,
                              std::vector<unsigned char> *codes_host) {
  std::vector<T> dict(values, values + n);
  std::sort(dict.begin(), dict.end());
  dict.erase(std::unique(dict.begin(), dict.end()), dict.end());
  unsigned code_bits = 0;
  for (std::size_t num_values = 1; num_values <= dict.size(); ++num_values) {
    if (code_bits == 0 || num_values == (1ull << code_bits) + 1) ++code_bits;
  }
  const std::size_t dict_code_bytes = code_bits < 9 ? 1 : (code_bits < 17 ? 2 : 4);
  const std::size_t dictionary_bytes = 2 * sizeof(std::uint32_t) + dict.size() * sizeof(T) + static_cast<std::size_t>(n) * dict_code_bytes;
  std::size_t truncated_width = sizeof(T);
  if ((type == kInt || type == kLong) && n > 0) {
    bool negative = false;
    std::int64_t mx = 0;
    for (std::int64_t i = 0; i < n; ++i) {
      const std::int64_t v = static_cast<std::int64_t>(values[i]);
      negative = negative || v < 0;
      mx = std::max(mx, v);
    }
    if (!negative && !(type == kLong && mx == 0xFFFFFFFFll)) {
      unsigned needed_bits = 0;
      while (needed_bits < 64 && (static_cast<std::uint64_t>(mx) >> needed_bits) != 0) ++needed_bits;
      if (needed_bits < 9) truncated_width = 1;
      else if (needed_bits < 17) truncated_width = 2;
      else if (needed_bits < 33) truncated_width = 4;
    }
  }
  auto put_code = [&](std::int64_t i, std::uint64_t code, std::size_t width) {
    switch (width) {
      case 1: (*codes_host)[i] = static_cast<unsigned char>(code); break;
      case 2: reinterpret_cast<std::uint16_t *>(codes_host->data())[i] = static_cast<std::uint16_t>(code); break;
      default: reinterpret_cast<std::uint32_t *>(codes_host->data())[i] = static_cast<std::uint32_t>(code); break;
    }
  };
  if (static_cast<std::size_t>(n) * truncated_width < dictionary_bytes) {
    if (truncated_width == sizeof(T)) {
      out->kind = CompressedAttribute::kUncompressed;
      return;
    }
    out->kind = CompressedAttribute::kTruncated;
    out->code_width = static_cast<int>(truncated_width);
    codes_host->assign(static_cast<std::size_t>(n) * truncated_width, 0);
    for (std::int64_t i = 0; i < n; ++i) put_code(i, static_cast<std::uint64_t>(static_cast<std::int64_t>(values[i])), truncated_width);
    return;
  }
  out->kind = CompressedAttribute::kDictionary;
  out->code_width = static_cast<int>(dict_code_bytes);
  out->num_codes = static_cast<std::uint32_t>(dict.size());
  out->dictionary_host.assign(reinterpret_cast<const unsigned char *>(dict.data()),
                              reinterpret_cast<const unsigned char *>(dict.data() + dict.size()));
  codes_host->assign(static_cast<std::size_t>(n) * dict_code_bytes, 0);
  for (std::int64_t i = 0; i < n; ++i) {
    put_code(i, static_cast<std::uint64_t>(std::lower_bound(dict.begin(), dict.end(), values[i]) - dict.begin()), dict_code_bytes);
  }
}

template <typename T>
PredicateTransformResult TransformT(const CompressedAttribute &attr, ComparisonID comparison, T lit) {
  PredicateTransformResult r;   // kNone
  auto basic = [&](qsx_code_cmp_t comp, std::uint32_t code) {
    r.type = PredicateTransformResult::kBasicComparison;
    r.comp = comp;
    r.first_literal = code;
  };
  constexpr std::uint32_t kMax = 0xFFFFFFFFu;
  std::pair<std::uint32_t, std::uint32_t> range(0, 0);
  if (attr.kind == CompressedAttribute::kDictionary) {
    const T *dict = reinterpret_cast<const T *>(attr.dictionary_host.data());
    const T *end = dict + attr.num_codes;
    const std::uint32_t lower = static_cast<std::uint32_t>(std::lower_bound(dict, end, lit) - dict);
    const std::uint32_t upper = static_cast<std::uint32_t>(std::upper_bound(dict, end, lit) - dict);
    if (comparison == ComparisonID::kEqual) {           // TransformEqualPredicateOnCompressedAttribute (:425-470)
      if (lower != upper) basic(QSX_CODE_EQ, lower);
      return r;
    }
    if (comparison == ComparisonID::kNotEqual) {        // TransformNotEqualPredicate... (:472-535), no null code
      if (lower == upper) r.type = PredicateTransformResult::kAll;
      else basic(QSX_CODE_NE, lower);
      return r;
    }
    switch (comparison) {                               // getLimitCodesForComparisonTyped (CompressionDictionary.cpp:276-305)
      case ComparisonID::kLess: range = {0, lower}; break;
      case ComparisonID::kLessOrEqual: range = {0, upper}; break;
      case ComparisonID::kGreater: range = {upper, attr.num_codes}; break;
      default: range = {lower, attr.num_codes}; break;
    }
    if (range.first >= range.second) return r;
    if (range.second == attr.num_codes) range.second = kMax;
  } else {
    // truncated attribute (:144-236 TruncationHelper, :266-420 always-true / always-false)
    const std::int64_t max_truncated = attr.code_width == 4 ? 0xFFFFFFFFll : (1ll << (8 * attr.code_width)) - 1;
    const double as_double = static_cast<double>(lit);
    const bool long_exact = std::is_integral<T>::value || as_double == static_cast<double>(static_cast<std::int64_t>(as_double));
    const std::int64_t as_long = static_cast<std::int64_t>(lit);
    const bool in_range = as_long >= 0 && as_long <= max_truncated;
    if (comparison == ComparisonID::kEqual) {
      if (long_exact && in_range) basic(QSX_CODE_EQ, static_cast<std::uint32_t>(as_long));
      return r;
    }
    if (comparison == ComparisonID::kNotEqual) {
      if (!long_exact || !in_range) r.type = PredicateTransformResult::kAll;
      else basic(QSX_CODE_NE, static_cast<std::uint32_t>(as_long));
      return r;
    }
    const bool lower_side = comparison == ComparisonID::kLess || comparison == ComparisonID::kGreaterOrEqual;
    const std::int64_t eff = long_exact ? as_long
                                        : static_cast<std::int64_t>(lower_side ? std::ceil(as_double) : std::floor(as_double));
    bool always_true = false, always_false = false;
    switch (comparison) {
      case ComparisonID::kLess: always_true = eff > max_truncated; always_false = eff <= 0; break;
      case ComparisonID::kLessOrEqual: always_true = eff >= max_truncated; always_false = eff < 0; break;
      case ComparisonID::kGreater: always_true = eff < 0; always_false = eff >= max_truncated; break;
      default: always_true = eff <= 0; always_false = eff > max_truncated; break;
    }
    if (always_true) { r.type = PredicateTransformResult::kAll; return r; }
    if (always_false) return r;
    switch (comparison) {
      case ComparisonID::kLess: range = {0, static_cast<std::uint32_t>(eff)}; break;
      case ComparisonID::kLessOrEqual: range = {0, static_cast<std::uint32_t>(eff + 1)}; break;
      case ComparisonID::kGreater: range = {static_cast<std::uint32_t>(eff + 1), kMax}; break;
      default: range = {static_cast<std::uint32_t>(eff), kMax}; break;
    }
  }
  if (range.first == 0) {                                // :590-612
    if (range.second == kMax) r.type = PredicateTransformResult::kAll;
    else basic(QSX_CODE_LT, range.second);
  } else if (range.second == kMax) {
    basic(QSX_CODE_GE, range.first);
  } else {
    r.type = PredicateTransformResult::kRangeComparison;
    r.comp = QSX_CODE_RANGE;
    r.first_literal = range.first;
    r.second_literal = range.second;
  }
  return r;
}

// ---- DATE and CHAR(n) attributes: dictionaries of fixed-width byte values under the type's own order -------------------
// (CompressionDictionaryBuilder keeps its values in a set ordered by the type's less comparison,
// compression/CompressionDictionaryBuilder.cpp:40-90; there is no truncation for these types.)
// strcmpHelper of the reference (types/operations/comparisons/AsciiStringComparators.hpp:218-251)
int CompareAsciiStrings(const char *left, std::size_t left_length, const char *right, std::size_t right_length) {
  if (right_length > left_length) {
    const int res = std::strncmp(left, right, left_length);
    if (res) return res;
    return strnlen(right, right_length) > left_length ? -1 : res;
  } else if (left_length > right_length) {
    const int res = std::strncmp(left, right, right_length);
    if (res) return res;
    return strnlen(left, left_length) > right_length ? 1 : res;
  }
  return std::strncmp(left, right, left_length);
}
// <0, 0, >0: value (value_width bytes) against the literal
int CompareBytesWithLiteral(TypeID type, const unsigned char *value, int value_width, const void *literal, std::size_t literal_length) {
  if (type == kDate) {
    DateLit a, b;
    std::memcpy(&a, value, 8);
    std::memcpy(&b, literal, 8);
    return a < b ? -1 : (b < a ? 1 : 0);
  }
  return CompareAsciiStrings(reinterpret_cast<const char *>(value), static_cast<std::size_t>(value_width),
                             static_cast<const char *>(literal), literal_length);
}

void BuildByteDictionary(TypeID type, int width, const unsigned char *values, std::int64_t n, CompressedAttribute *out,
                         std::vector<unsigned char> *codes_host) {
  auto less = [&](const unsigned char *a, const unsigned char *b) { return CompareBytesWithLiteral(type, a, width, b, width) < 0; };
  std::vector<const unsigned char *> dict(static_cast<std::size_t>(n));
  for (std::int64_t i = 0; i < n; ++i) dict[static_cast<std::size_t>(i)] = values + i * width;
  std::sort(dict.begin(), dict.end(), less);
  dict.erase(std::unique(dict.begin(), dict.end(), [&](const unsigned char *a, const unsigned char *b) { return !less(a, b) && !less(b, a); }),
             dict.end());
  unsigned code_bits = 0;
  for (std::size_t num_values = 1; num_values <= dict.size(); ++num_values) {
    if (code_bits == 0 || num_values == (1ull << code_bits) + 1) ++code_bits;
  }
  const std::size_t code_bytes = code_bits < 9 ? 1 : (code_bits < 17 ? 2 : 4);
  const std::size_t dictionary_bytes = 2 * sizeof(std::uint32_t) + dict.size() * width + static_cast<std::size_t>(n) * code_bytes;
  if (static_cast<std::size_t>(n) * width < dictionary_bytes) {   // CompressedBlockBuilder.cpp:590-650: compress only if it is smaller
    out->kind = CompressedAttribute::kUncompressed;
    return;
  }
  out->kind = CompressedAttribute::kDictionary;
  out->code_width = static_cast<int>(code_bytes);
  out->num_codes = static_cast<std::uint32_t>(dict.size());
  out->value_width = width;
  out->dictionary_host.assign(dict.size() * width, 0);
  for (std::size_t e = 0; e < dict.size(); ++e) {
    // a DATE entry keeps year, month, day only; a CHAR entry ends at its terminator (what follows is not part of the value)
    const std::size_t keep = type == kDate ? 6 : strnlen(reinterpret_cast<const char *>(dict[e]), static_cast<std::size_t>(width));
    std::memcpy(out->dictionary_host.data() + e * width, dict[e], keep);
  }
  codes_host->assign(static_cast<std::size_t>(n) * code_bytes, 0);
  for (std::int64_t i = 0; i < n; ++i) {
    const std::uint32_t code = static_cast<std::uint32_t>(std::lower_bound(dict.begin(), dict.end(), values + i * width, less) - dict.begin());
    switch (code_bytes) {
      case 1: (*codes_host)[static_cast<std::size_t>(i)] = static_cast<unsigned char>(code); break;
      case 2: reinterpret_cast<std::uint16_t *>(codes_host->data())[i] = static_cast<std::uint16_t>(code); break;
      default: reinterpret_cast<std::uint32_t *>(codes_host->data())[i] = code; break;
    }
  }
}

// The dictionary branch of TransformT for byte dictionaries (same rules: CompressedStoreUtil.cpp:425-616,
// CompressionDictionary.cpp:276-305).
PredicateTransformResult TransformBytes(const CompressedAttribute &attr, TypeID type, ComparisonID comparison, const void *literal,
                                        std::size_t literal_length) {
  PredicateTransformResult r;
  const int width = attr.value_width;
  std::uint32_t lower = 0, upper = attr.num_codes;   // first entry >= literal, first entry > literal
  {
    std::uint32_t lo = 0, hi = attr.num_codes;
    while (lo < hi) {
      const std::uint32_t mid = lo + (hi - lo) / 2;
      if (CompareBytesWithLiteral(type, attr.dictionary_host.data() + static_cast<std::size_t>(mid) * width, width, literal, literal_length) < 0) lo = mid + 1;
      else hi = mid;
    }
    lower = lo;
    hi = attr.num_codes;
    while (lo < hi) {
      const std::uint32_t mid = lo + (hi - lo) / 2;
      if (CompareBytesWithLiteral(type, attr.dictionary_host.data() + static_cast<std::size_t>(mid) * width, width, literal, literal_length) <= 0) lo = mid + 1;
      else hi = mid;
    }
    upper = lo;
  }
  auto basic = [&](qsx_code_cmp_t comp, std::uint32_t code) {
    r.type = PredicateTransformResult::kBasicComparison;
    r.comp = comp;
    r.first_literal = code;
  };
  constexpr std::uint32_t kMax = 0xFFFFFFFFu;
  if (comparison == ComparisonID::kEqual) {
    if (lower != upper) basic(QSX_CODE_EQ, lower);
    return r;
  }
  if (comparison == ComparisonID::kNotEqual) {
    if (lower == upper) r.type = PredicateTransformResult::kAll;
    else basic(QSX_CODE_NE, lower);
    return r;
  }
  std::pair<std::uint32_t, std::uint32_t> range(0, 0);
  switch (comparison) {
    case ComparisonID::kLess: range = {0, lower}; break;
    case ComparisonID::kLessOrEqual: range = {0, upper}; break;
    case ComparisonID::kGreater: range = {upper, attr.num_codes}; break;
    default: range = {lower, attr.num_codes}; break;
  }
  if (range.first >= range.second) return r;
  if (range.second == attr.num_codes) range.second = kMax;
  if (range.first == 0) {
    if (range.second == kMax) r.type = PredicateTransformResult::kAll;
    else basic(QSX_CODE_LT, range.second);
  } else if (range.second == kMax) {
    basic(QSX_CODE_GE, range.first);
  } else {
    r.type = PredicateTransformResult::kRangeComparison;
    r.comp = QSX_CODE_RANGE;
    r.first_literal = range.first;
    r.second_literal = range.second;
  }
  return r;
}
}  // namespace

PredicateTransformResult TransformPredicateOnCompressedAttribute(const CompressedAttribute &attribute, TypeID type,
                                                                 ComparisonID comparison, const TypedLiteral &literal) {
  switch (type) {
    case kInt: return TransformT<std::int32_t>(attribute, comparison, literal.v.i32);
    case kLong: return TransformT<std::int64_t>(attribute, comparison, literal.v.i64);
    case kFloat: return TransformT<float>(attribute, comparison, literal.v.f32);
    case kDouble: return TransformT<double>(attribute, comparison, literal.v.f64);
    case kDate: return TransformBytes(attribute, type, comparison, &literal.v.i64, 8);
    case kChar: return TransformBytes(attribute, type, comparison, literal.text.data(), literal.text.size());
    default: throw ExecutionError("compressed attributes: INT / LONG / FLOAT / DOUBLE / DATE / CHAR(n)", QSX_ERR_UNSUPPORTED);
  }
}

void CompressValues(TypeID type, const void *values, std::int64_t n, CompressedAttribute *out,
                    std::vector<unsigned char> *codes_host, int value_width) {
  codes_host->clear();
  switch (type) {
    case kDate: BuildByteDictionary(type, 8, static_cast<const unsigned char *>(values), n, out, codes_host); break;
    case kChar:
      if (value_width > 0) BuildByteDictionary(type, value_width, static_cast<const unsigned char *>(values), n, out, codes_host);
      else out->kind = CompressedAttribute::kUncompressed;
      break;
    case kInt: BuildCompressedAttribute(type, static_cast<const std::int32_t *>(values), n, out, codes_host); break;
    case kLong: BuildCompressedAttribute(type, static_cast<const std::int64_t *>(values), n, out, codes_host); break;
    case kFloat: BuildCompressedAttribute(type, static_cast<const float *>(values), n, out, codes_host); break;
    case kDouble: BuildCompressedAttribute(type, static_cast<const double *>(values), n, out, codes_host); break;
    default: out->kind = CompressedAttribute::kUncompressed; break;
  }
}

void StorageBlock::compressAttribute(attribute_id a, const void *host_values) {
  if (g_host_memory) return;   // CPU plumbing mode keeps plain stripes
  const Type &t = relation_.getAttributeType(a);
  if (compressed_.empty()) compressed_.resize(relation_.size());
  CompressedAttribute &c = compressed_.at(a);
  std::vector<unsigned char> codes_host;
  CompressValues(t.id, host_values, num_tuples_, &c, &codes_host, t.width);
  if (c.kind == CompressedAttribute::kUncompressed) return;
  CheckStatus(qsx_device_alloc(codes_host.size() + 8, &c.codes), "qsx_device_alloc(codes)");
  CheckStatus(qsx_copy_to_device(c.codes, codes_host.data(), codes_host.size(), nullptr), "qsx_copy_to_device(codes)");
  if (c.kind == CompressedAttribute::kDictionary) {
    CheckStatus(qsx_device_alloc(c.dictionary_host.size() + 8, &c.dictionary), "qsx_device_alloc(dictionary)");
    CheckStatus(qsx_copy_to_device(c.dictionary, c.dictionary_host.data(), c.dictionary_host.size(), nullptr), "qsx_copy_to_device(dictionary)");
  }
  CheckStatus(qsx_stream_synchronize(nullptr), "qsx_stream_synchronize");
  if (slab_ == nullptr) qsx_device_free(stripes_.at(a));      // the values are gone until somebody asks for them (stripe())
  stripes_.at(a) = nullptr;
}

block_id StorageManager::createBlock(CatalogRelation *relation, std::int64_t capacity) {
  std::lock_guard<std::mutex> lock(mutex_);
  const block_id id = next_id_++;
  // first_row is fixed when the block is registered with its final size (returnBlock / loadBlock)
  blocks_[id] = std::make_shared<StorageBlock>(*relation, capacity, 0, /*one_allocation=*/true);
  return id;
}

block_id StorageManager::loadBlock(CatalogRelation *relation, const std::vector<const void *> &host_columns,
                                   std::int64_t num_tuples, partition_id part, const std::vector<bool> *compress,
                                   const std::vector<const std::uint64_t *> *null_bitmaps) {
  BlockReference block;
  block_id id;
  {
    std::lock_guard<std::mutex> lock(mutex_);
    id = next_id_++;
    std::int64_t &rows = rows_in_relation_[relation->getID()];
    block = std::make_shared<StorageBlock>(*relation, num_tuples, rows);
    rows += num_tuples;
    blocks_[id] = block;
  }
  for (std::size_t a = 0; a < relation->size(); ++a) {
    const std::size_t bytes = static_cast<std::size_t>(num_tuples) * relation->getAttributeType(static_cast<attribute_id>(a)).width;
    if (g_host_memory) {
      std::memcpy(block->stripe(static_cast<attribute_id>(a)), host_columns.at(a), bytes);
    } else {
      CheckStatus(qsx_copy_to_device(block->stripe(static_cast<attribute_id>(a)), host_columns.at(a), bytes, nullptr),
                  "qsx_copy_to_device");
    }
  }
  for (std::size_t a = 0; null_bitmaps != nullptr && a < relation->size() && a < null_bitmaps->size(); ++a) {
    if ((*null_bitmaps)[a] == nullptr) continue;
    std::uint64_t *dst = block->nullBitmap(static_cast<attribute_id>(a));
    if (dst == nullptr) throw ExecutionError("loadBlock: null bitmap given for a non-nullable attribute", QSX_ERR_INVALID_ARGUMENT);
    const std::size_t bytes = static_cast<std::size_t>((num_tuples + 63) / 64) * 8;
    if (g_host_memory) {
      std::memcpy(dst, (*null_bitmaps)[a], bytes);
    } else {
      CheckStatus(qsx_copy_to_device(dst, (*null_bitmaps)[a], bytes, nullptr), "qsx_copy_to_device(null bitmap)");
    }
  }
  if (!g_host_memory) CheckStatus(qsx_stream_synchronize(nullptr), "qsx_stream_synchronize");
  block->setNumTuples(num_tuples);
  if (compress != nullptr) {
    for (std::size_t a = 0; a < relation->size() && a < compress->size(); ++a) {
      if ((*compress)[a]) block->compressAttribute(static_cast<attribute_id>(a), host_columns.at(a));
    }
  }
  relation->addBlockToPartition(id, part);
  return id;
}

BlockReference StorageManager::getBlock(block_id id) const {
  std::lock_guard<std::mutex> lock(mutex_);
  auto it = blocks_.find(id);
  if (it == blocks_.end()) throw std::out_of_range("StorageManager::getBlock: unknown block");
  return it->second;
}

std::int64_t StorageManager::reserveRows(relation_id relation, std::int64_t num_tuples) {
  std::lock_guard<std::mutex> lock(mutex_);
  std::int64_t &rows = rows_in_relation_[relation];
  const std::int64_t first = rows;
  rows += num_tuples;
  return first;
}

// ---- reference block images ------------------------------------------------------------------------------------------------
namespace {
// protobuf wire format, as far as a StorageBlockHeader needs it (varints, fixed64 / fixed32, length-delimited fields)
struct WireReader {
  const unsigned char *at, *end;
  bool ok = true;
  std::uint64_t varint() {
    std::uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      if (at >= end) { ok = false; return 0; }
      const unsigned char b = *at++;
      v |= static_cast<std::uint64_t>(b & 0x7F) << shift;
      if ((b & 0x80) == 0) return v;
    }
    ok = false;
    return 0;
  }
  std::uint64_t fixed(int bytes) {
    if (end - at < bytes) { ok = false; return 0; }
    std::uint64_t v = 0;
    std::memcpy(&v, at, static_cast<std::size_t>(bytes));
    at += bytes;
    return v;
  }
  WireReader sub() {   // a length-delimited field
    const std::uint64_t len = varint();
    if (!ok || static_cast<std::uint64_t>(end - at) < len) { ok = false; return WireReader{at, at}; }
    WireReader r{at, at + len};
    at += len;
    return r;
  }
  void skip(int wire_type) {
    switch (wire_type) {
      case 0: (void)varint(); break;
      case 1: (void)fixed(8); break;
      case 2: (void)sub(); break;
      case 5: (void)fixed(4); break;
      default: ok = false;
    }
  }
};
[[noreturn]] void Malformed(const char *what) {
  throw ExecutionError(std::string("malformed block image: ") + what, QSX_ERR_INVALID_ARGUMENT);
}
}  // namespace

ReferenceBlockLayout ParseReferenceBlockImage(const CatalogRelation &relation, const void *prefix, std::size_t prefix_bytes,
                                              std::size_t image_bytes) {
  const unsigned char *bytes = static_cast<const unsigned char *>(prefix);
  if (prefix_bytes < sizeof(std::int32_t) || prefix_bytes > image_bytes) Malformed("shorter than its length word");
  std::int32_t header_length = 0;
  std::memcpy(&header_length, bytes, sizeof(header_length));
  if (header_length <= 0 || static_cast<std::size_t>(header_length) + sizeof(std::int32_t) > image_bytes) Malformed("header length");   // StorageBlock.cpp:112-117
  if (static_cast<std::size_t>(header_length) + sizeof(std::int32_t) + 8 > prefix_bytes) Malformed("the prefix handed in does not cover the block header");
  // StorageBlockHeader { layout = 1 (StorageBlockLayoutDescription { num_slots = 1; tuple_store_description = 2 {
  //   sub_block_type = 1; [sort_attribute_id = 64] }; index_description = 3 }); fixed64 tuple_store_size = 2; ... }
  WireReader header{bytes + sizeof(std::int32_t), bytes + sizeof(std::int32_t) + header_length};
  ReferenceBlockLayout out;
  bool have_layout = false, have_size = false;
  std::uint64_t sub_block_type = ~0ull;
  while (header.ok && header.at < header.end) {
    const std::uint64_t tag = header.varint();
    const int field = static_cast<int>(tag >> 3), wire = static_cast<int>(tag & 7);
    if (field == 1 && wire == 2) {
      WireReader layout = header.sub();
      have_layout = true;
      while (layout.ok && layout.at < layout.end) {
        const std::uint64_t ltag = layout.varint();
        if ((ltag >> 3) == 2 && (ltag & 7) == 2) {
          WireReader store = layout.sub();
          while (store.ok && store.at < store.end) {
            const std::uint64_t stag = store.varint();
            if ((stag >> 3) == 1 && (stag & 7) == 0) sub_block_type = store.varint();
            else if ((stag >> 3) == 64 && (stag & 7) == 0) out.sort_attribute = static_cast<attribute_id>(static_cast<std::int32_t>(store.varint()));
            else store.skip(static_cast<int>(stag & 7));
          }
          if (!store.ok) Malformed("tuple store description");
        } else {
          layout.skip(static_cast<int>(ltag & 7));
        }
      }
      if (!layout.ok) Malformed("layout description");
    } else if (field == 2 && wire == 1) {
      out.tuple_store_size = static_cast<std::size_t>(header.fixed(8));
      have_size = true;
    } else {
      header.skip(wire);
    }
  }
  if (!header.ok || !have_layout || !have_size || sub_block_type == ~0ull) Malformed("block header");   // !IsInitialized()
  if (sub_block_type != 0) {   // TupleStorageSubBlockDescription::BASIC_COLUMN_STORE
    throw ExecutionError("block image: the tuple store is not a BasicColumnStore (compressed / row stores are not adopted in place)", QSX_ERR_UNSUPPORTED);
  }
  out.tuple_store_offset = sizeof(std::int32_t) + static_cast<std::size_t>(header_length);
  if (out.tuple_store_offset + out.tuple_store_size > image_bytes) Malformed("sub-block sizes exceed the block");   // :141-143
  if (out.tuple_store_size < 8) Malformed("tuple store smaller than its header");   // BlockMemoryTooSmall
  // BasicColumnStoreTupleStorageSubBlock.cpp:131-147
  std::size_t row_bytes = 0, nullable = 0;
  for (std::size_t a = 0; a < relation.size(); ++a) {
    const Type &t = relation.getAttributeType(static_cast<attribute_id>(a));
    row_bytes += static_cast<std::size_t>(t.width);
    nullable += t.nullable ? 1 : 0;
  }
  if (row_bytes == 0) Malformed("relation without attributes");
  auto bitmap_bytes = [](std::size_t bits) { return (bits + 63) / 64 * 8; };   // BitVector<false>::BytesNeeded
  std::size_t max_tuples = ((out.tuple_store_size - 8) << 3) / ((row_bytes << 3) + nullable);
  if (max_tuples == 0) Malformed("no room for one tuple");
  if (nullable * bitmap_bytes(max_tuples) + 8 > out.tuple_store_size) Malformed("no room for the null bitmaps");
  max_tuples = (out.tuple_store_size - 8 - nullable * bitmap_bytes(max_tuples)) / row_bytes;
  if (max_tuples == 0) Malformed("no room for one tuple");
  const std::size_t per_bitmap = bitmap_bytes(max_tuples);
  out.max_tuples = static_cast<std::int64_t>(max_tuples);
  std::int32_t num_tuples = 0;
  std::memcpy(&num_tuples, bytes + out.tuple_store_offset, sizeof(num_tuples));
  if (num_tuples < 0 || static_cast<std::size_t>(num_tuples) > max_tuples) Malformed("num_tuples");
  out.num_tuples = num_tuples;
  std::size_t at = out.tuple_store_offset + 8;
  for (std::size_t a = 0; a < relation.size(); ++a) {
    if (relation.getAttributeType(static_cast<attribute_id>(a)).nullable) {
      out.null_bitmap_offset.push_back(at);
      at += per_bitmap;
    } else {
      out.null_bitmap_offset.push_back(static_cast<std::size_t>(-1));
    }
  }
  for (std::size_t a = 0; a < relation.size(); ++a) {
    out.stripe_offset.push_back(at);
    at += max_tuples * static_cast<std::size_t>(relation.getAttributeType(static_cast<attribute_id>(a)).width);
  }
  if (at > out.tuple_store_offset + out.tuple_store_size) Malformed("stripes exceed the tuple store");
  if (out.sort_attribute != kInvalidAttributeID && (out.sort_attribute < 0 || static_cast<std::size_t>(out.sort_attribute) >= relation.size())) {
    Malformed("sort attribute");
  }
  return out;
}

block_id StorageManager::adoptBlockImage(CatalogRelation *relation, void *image_dev, std::size_t image_bytes, partition_id part) {
  // the block header is a few hundred bytes: fetch a prefix, parse on the host
  const std::size_t prefix_bytes = std::min<std::size_t>(image_bytes, 16384);
  std::vector<unsigned char> prefix(prefix_bytes);
  if (g_host_memory) {
    std::memcpy(prefix.data(), image_dev, prefix_bytes);
  } else {
    CheckStatus(qsx_copy_to_host(prefix.data(), image_dev, prefix_bytes, CurrentStream()), "qsx_copy_to_host(block header)");
  }
  const ReferenceBlockLayout layout = ParseReferenceBlockImage(*relation, prefix.data(), prefix_bytes, image_bytes);
  std::vector<void *> stripes, nulls;
  char *base = static_cast<char *>(image_dev);
  for (std::size_t a = 0; a < relation->size(); ++a) {
    stripes.push_back(base + layout.stripe_offset[a]);
    nulls.push_back(layout.null_bitmap_offset[a] == static_cast<std::size_t>(-1) ? nullptr : base + layout.null_bitmap_offset[a]);
  }
  BlockReference block = std::make_shared<StorageBlock>(*relation, layout.num_tuples, stripes, nulls);
  block->setSortColumn(layout.sort_attribute);
  block_id id;
  {
    std::lock_guard<std::mutex> lock(mutex_);
    id = next_id_++;
    std::int64_t &rows = rows_in_relation_[relation->getID()];
    block->setFirstRow(rows);
    rows += layout.num_tuples;
    blocks_[id] = block;
  }
  relation->addBlockToPartition(id, part);
  return id;
}

block_id StorageManager::createViewBlock(block_id parent, std::int64_t first_tuple, std::int64_t num_tuples) {
  BlockReference p = getBlock(parent);
  BlockReference view = std::make_shared<StorageBlock>(p, first_tuple, num_tuples);
  std::lock_guard<std::mutex> lock(mutex_);
  const block_id id = next_id_++;
  blocks_[id] = view;
  return id;
}

void StorageManager::deleteBlockOrBlobFile(block_id id) {
  std::lock_guard<std::mutex> lock(mutex_);
  blocks_.erase(id);
}

namespace {
// The tuples of `block` selected by `filter` (nullptr = all) whose attributes `attrs` are all non-NULL, as a device
// bitmap — or nullptr when no attribute is nullable (the caller keeps using `filter`).  This is the reference's
// check_for_null_keys skip (HashTable.hpp:1409-1418, 2158-2160) and the NULL argument skip of the aggregate handles
// (AggregationHandleSum.hpp:105-120), done once per block on the TupleIdSequence instead of per value.
std::unique_ptr<DeviceBuffer> NotNullFilter(const StorageBlock &block, const std::vector<attribute_id> &attrs, const std::uint64_t *filter) {
  const std::int64_t n = block.numTuples();
  std::unique_ptr<DeviceBuffer> out;
  for (attribute_id a : attrs) {
    const std::uint64_t *nulls = block.nullBitmap(a);
    if (nulls == nullptr) continue;
    const bool first = out == nullptr;
    if (first) out.reset(new DeviceBuffer(static_cast<std::size_t>((n + 63) / 64) * 8 + 8));
    if (n == 0) continue;
    if (first && filter == nullptr) {
      CheckStatus(qsx_bitmap_combine(3, nulls, nullptr, n, static_cast<std::uint64_t *>(out->ptr), CurrentStream()), "qsx_bitmap_combine");
    } else {
      CheckStatus(qsx_bitmap_combine(2, first ? filter : static_cast<const std::uint64_t *>(out->ptr), nulls, n,
                                     static_cast<std::uint64_t *>(out->ptr), CurrentStream()), "qsx_bitmap_combine");
    }
  }
  return out;
}

// Null bits of the rows `tids` of one block's attribute -> dst (an output block's null bitmap).
void GatherBlockNulls(const StorageBlock &block, attribute_id attr, const void *tids, std::int64_t n, std::uint64_t *dst) {
  const std::uint64_t *seg = block.nullBitmap(attr);
  const std::int64_t zero = 0;
  CheckStatus(qsx_bitmap_gather_segmented(1, &seg, &zero, static_cast<const std::int32_t *>(tids), n, dst, CurrentStream()),
              "qsx_bitmap_gather_segmented");
}

// The null bits of the selected tuples of `block` follow the values of a projection: output attribute i takes the
// bits of input attribute selection[i] at the tuples set in `bitmap` (bulkInsertTuplesWithRemappedAttributes copies
// value and null bit together, storage/BasicColumnStoreTupleStorageSubBlock.cpp:339-425).
void ProjectNullBitmaps(const StorageBlock &block, const std::vector<attribute_id> &selection, const void *bitmap,
                        std::int64_t num_selected, StorageBlock *out) {
  const std::int64_t n = block.numTuples();
  std::unique_ptr<DeviceBuffer> tids;
  for (std::size_t i = 0; i < selection.size(); ++i) {
    if (selection[i] == kInvalidAttributeID) continue;          // an expression's value: no bitmap to carry over
    if (block.nullBitmap(selection[i]) == nullptr) continue;   // the output bitmap stays all-zero
    std::uint64_t *dst = out->nullBitmap(static_cast<attribute_id>(i));
    if (dst == nullptr) throw ExecutionError("projection of a nullable attribute into a non-nullable one", QSX_ERR_INVALID_ARGUMENT);
    if (num_selected == 0) continue;
    if (tids == nullptr) {
      tids.reset(new DeviceBuffer(static_cast<std::size_t>(n) * 4 + 16));
      const std::size_t ws_bytes = qsx_compact_workspace_bytes(n);
      DeviceBuffer ws(ws_bytes), count(8);
      CheckStatus(qsx_bitmap_to_tids(static_cast<const std::uint64_t *>(bitmap), n, 0, static_cast<std::int32_t *>(tids->ptr),
                                     static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()), "qsx_bitmap_to_tids");
      CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");   // ws / count are locals
    }
    GatherBlockNulls(block, selection[i], tids->ptr, num_selected, dst);
  }
  if (tids != nullptr) CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
}
}  // namespace

// ---------------------------------------------------------------------------
// Predicate
// ---------------------------------------------------------------------------
void *Predicate::getMatchesForBlock(const StorageBlock &block, std::int64_t *num_matches, const std::uint64_t *filter) const {
  const std::int64_t n = block.numTuples();
  const std::size_t words = static_cast<std::size_t>((n + 63) / 64);
  void *current = nullptr, *next = nullptr, *count = nullptr;
  CheckStatus(qsx_device_alloc(words * 8 + 8, &current), "qsx_device_alloc(bitmap)");
  CheckStatus(qsx_device_alloc(words * 8 + 8, &next), "qsx_device_alloc(bitmap)");
  CheckStatus(qsx_device_alloc(8, &count), "qsx_device_alloc(count)");
  bool first = true;
  bool recount = false;
  for (const ComparisonPredicate &term : conjuncts) {
    const Type &t = block.getRelation().getAttributeType(term.attribute);
    const std::uint64_t *in = first ? filter : static_cast<const std::uint64_t *>(current);
    // conjunctions chain the filter through their children (short-circuit)
    if (const CompressedAttribute *c = block.compressedAttribute(term.attribute)) {
      // CompressedTupleStorageSubBlock::getMatchesForPredicate (storage/CompressedTupleStorageSubBlock.cpp:160-250):
      // rewrite to a comparison on codes, scan the code stripe
      const PredicateTransformResult r = TransformPredicateOnCompressedAttribute(*c, t.id, term.comparison, term.literal);
      if (r.type == PredicateTransformResult::kAll || r.type == PredicateTransformResult::kNone) {
        // every code / no code: code >= 0 resp. code < 0
        CheckStatus(qsx_select_codes(c->code_width, c->codes, n, r.type == PredicateTransformResult::kAll ? QSX_CODE_GE : QSX_CODE_LT, 0, 0,
                                     in, static_cast<std::uint64_t *>(next), static_cast<std::int64_t *>(count), CurrentStream()),
                    "qsx_select_codes");
      } else if (term.attribute == block.sortColumn()) {
        // the sort column of a compressed store: the codes ascend, the matches are one range (the sort-column branches of
        // CompressedColumnStoreTupleStorageSubBlock.cpp:420-760)
        CheckStatus(qsx_select_codes_sorted(c->code_width, c->codes, n, r.comp, r.first_literal, r.second_literal, in,
                                            static_cast<std::uint64_t *>(next), static_cast<std::int64_t *>(count), CurrentStream()),
                    "qsx_select_codes_sorted");
      } else {
        CheckStatus(qsx_select_codes(c->code_width, c->codes, n, r.comp, r.first_literal, r.second_literal, in,
                                     static_cast<std::uint64_t *>(next), static_cast<std::int64_t *>(count), CurrentStream()),
                    "qsx_select_codes");
      }
    } else if (t.id == kChar) {
      // CHAR(n) OP string literal (AsciiStringUncheckedComparator, AsciiStringComparators.hpp:218-251)
      CheckStatus(qsx_select_cmp_char(block.stripe(term.attribute), t.width, n, static_cast<int>(term.comparison), term.literal.text.data(),
                                      static_cast<int>(term.literal.text.size()), in, static_cast<std::uint64_t *>(next),
                                      static_cast<std::int64_t *>(count), CurrentStream()), "qsx_select_cmp_char");
    } else if (term.attribute == block.sortColumn()) {
      // the block is sorted on this attribute: SortColumnPredicateEvaluator (storage/ColumnStoreUtil.cpp:40-280)
      CheckStatus(qsx_select_cmp_sorted(t.id, block.stripe(term.attribute), n, static_cast<int>(term.comparison), &term.literal.v, in,
                                        static_cast<std::uint64_t *>(next), static_cast<std::int64_t *>(count), CurrentStream()),
                  "qsx_select_cmp_sorted");
    } else {
      CheckStatus(qsx_select_cmp(t.id, block.stripe(term.attribute), n, static_cast<int>(term.comparison), &term.literal.v, in,
                                 static_cast<std::uint64_t *>(next), static_cast<std::int64_t *>(count), CurrentStream()),
                  "qsx_select_cmp");
    }
    if (block.nullBitmap(term.attribute) != nullptr && n > 0) {
      // a comparison with NULL is not true (LiteralComparators-inl.hpp:330-370: the nullable variants test the value
      // pointer first): the stripe holds an arbitrary value under a NULL, so its match is taken back
      CheckStatus(qsx_bitmap_combine(2, static_cast<const std::uint64_t *>(next), block.nullBitmap(term.attribute), n,
                                     static_cast<std::uint64_t *>(next), CurrentStream()), "qsx_bitmap_combine");
      recount = true;
    }
    std::swap(current, next);
    first = false;
  }
  if (recount) {
    CheckStatus(qsx_bitmap_count(static_cast<const std::uint64_t *>(current), n, static_cast<std::int64_t *>(count), CurrentStream()),
                "qsx_bitmap_count");
  }
  if (first) {  // empty conjunction: every tuple (of the filter) matches
    CheckStatus(qsx_memset_device(next, 0xFF, words * 8, CurrentStream()), "qsx_memset_device");
    CheckStatus(qsx_bitmap_combine(0, static_cast<const std::uint64_t *>(next),
                                   filter != nullptr ? filter : static_cast<const std::uint64_t *>(next), n,
                                   static_cast<std::uint64_t *>(current), CurrentStream()), "qsx_bitmap_combine");
    CheckStatus(qsx_bitmap_count(static_cast<const std::uint64_t *>(current), n, static_cast<std::int64_t *>(count),
                                 CurrentStream()), "qsx_bitmap_count");
  }
  *num_matches = ReadCount(count);
  qsx_device_free(next);
  qsx_device_free(count);
  return current;
}

// ---------------------------------------------------------------------------
// InsertDestination
// ---------------------------------------------------------------------------
BlockReference InsertDestination::getBlockForInsertion(std::int64_t capacity, block_id *id) {
  *id = storage_manager_->createBlock(relation_, capacity);
  return storage_manager_->getBlock(*id);
}
void InsertDestination::returnBlock(block_id id, std::int64_t num_tuples, partition_id input_partition) {
  if (isPartitionAware()) {
    repartitionBlock(id, num_tuples);
    return;
  }
  BlockReference block = storage_manager_->getBlock(id);
  block->setNumTuples(num_tuples);
  block->setFirstRow(storage_manager_->reserveRows(relation_->getID(), num_tuples));
  std::lock_guard<std::mutex> lock(mutex_);
  touched_.push_back(TouchedBlock{id, input_partition});
  if (relation_->hasPartitionScheme()) {
    relation_->addBlockToPartition(id, input_partition);   // the output keeps the input's partitioning (no repartition)
  } else {
    relation_->addBlock(id);
  }
}

// bulkInsertTuples of a PartitionAwareInsertDestination (storage/InsertDestination.hpp:560-660), on a whole block at once:
// K9 scatters the columns of 1 / 2 / 4 / 8 bytes and a row-number column by the partition of the partition attribute; wider
// columns (CHAR(n)) and the null bits follow through the scattered row numbers; the scattered block is then cut into one
// block per partition (views: no copy).
void InsertDestination::repartitionBlock(block_id id, std::int64_t num_tuples) {
  BlockReference src = storage_manager_->getBlock(id);
  src->setNumTuples(num_tuples);
  const std::size_t P = num_partitions_;
  const Type &key_type = relation_->getAttributeType(partition_attribute_);
  if (key_type.id != kInt && key_type.id != kLong) {
    throw ExecutionError("PartitionAwareInsertDestination: the partition attribute must be INT or LONG", QSX_ERR_UNSUPPORTED);
  }
  if (P > 64) throw ExecutionError("PartitionAwareInsertDestination: more than 64 partitions", QSX_ERR_UNSUPPORTED);
  if (num_tuples == 0) {
    storage_manager_->deleteBlockOrBlobFile(id);
    return;
  }
  block_id scattered_id;
  BlockReference scattered = getBlockForInsertion(num_tuples, &scattered_id);
  scattered->setNumTuples(num_tuples);
  bool need_rows = false;
  std::vector<const void *> cols;
  std::vector<void *> outs;
  std::vector<std::int32_t> widths;
  for (std::size_t a = 0; a < relation_->size(); ++a) {
    const Type &t = relation_->getAttributeType(static_cast<attribute_id>(a));
    if (t.nullable) need_rows = true;
    if (t.width == 1 || t.width == 2 || t.width == 4 || t.width == 8) {
      cols.push_back(src->stripe(static_cast<attribute_id>(a)));
      outs.push_back(scattered->stripe(static_cast<attribute_id>(a)));
      widths.push_back(t.width);
    } else {
      need_rows = true;
    }
  }
  std::unique_ptr<DeviceBuffer> rows, rows_scattered;
  if (need_rows) {
    // row numbers 0 .. n-1: the tuple ids of an all-ones TupleIdSequence (NOT of a zeroed one: trailing bits stay zero)
    const std::size_t words = static_cast<std::size_t>((num_tuples + 63) / 64) + 1;
    DeviceBuffer zero(words * 8), ones(words * 8), count(8);
    CheckStatus(qsx_memset_device(zero.ptr, 0, words * 8, CurrentStream()), "qsx_memset_device");
    CheckStatus(qsx_bitmap_combine(3, static_cast<const std::uint64_t *>(zero.ptr), nullptr, num_tuples, static_cast<std::uint64_t *>(ones.ptr),
                                   CurrentStream()), "qsx_bitmap_combine");
    rows.reset(new DeviceBuffer(static_cast<std::size_t>(num_tuples) * 4 + 8));
    rows_scattered.reset(new DeviceBuffer(static_cast<std::size_t>(num_tuples) * 4 + 8));
    const std::size_t tws = qsx_compact_workspace_bytes(num_tuples);
    DeviceBuffer tw(tws + 8);
    CheckStatus(qsx_bitmap_to_tids(static_cast<const std::uint64_t *>(ones.ptr), num_tuples, 0, static_cast<std::int32_t *>(rows->ptr),
                                   static_cast<std::int64_t *>(count.ptr), tw.ptr, tws, CurrentStream()), "qsx_bitmap_to_tids");
    cols.push_back(rows->ptr);
    outs.push_back(rows_scattered->ptr);
    widths.push_back(4);
    CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");   // (zero / ones / tw go out of scope)
  }
  const std::size_t ws_bytes = qsx_partition_workspace_bytes(num_tuples, static_cast<int>(P));
  DeviceBuffer ws(ws_bytes + 8), offsets_dev((P + 1) * 8);
  CheckStatus(qsx_partition_scatter(key_type.id, src->stripe(partition_attribute_), num_tuples, static_cast<int>(P), static_cast<int>(cols.size()),
                                    cols.data(), widths.data(), outs.data(), static_cast<std::int64_t *>(offsets_dev.ptr), ws.ptr, ws_bytes,
                                    CurrentStream()), "qsx_partition_scatter");
  std::vector<std::int64_t> offsets(P + 1);
  CheckStatus(qsx_copy_to_host(offsets.data(), offsets_dev.ptr, (P + 1) * 8, CurrentStream()), "qsx_copy_to_host");
  for (std::size_t a = 0; a < relation_->size(); ++a) {
    const Type &t = relation_->getAttributeType(static_cast<attribute_id>(a));
    if (t.width == 1 || t.width == 2 || t.width == 4 || t.width == 8) continue;
    CheckStatus(qsx_gather(t.width, src->stripe(static_cast<attribute_id>(a)), static_cast<const std::int32_t *>(rows_scattered->ptr), num_tuples,
                           scattered->stripe(static_cast<attribute_id>(a)), CurrentStream()), "qsx_gather");
  }
  std::vector<std::pair<block_id, partition_id>> made;
  for (std::size_t p = 0; p < P; ++p) {
    const std::int64_t first = offsets[p], rows_p = offsets[p + 1] - offsets[p];
    if (rows_p == 0) continue;
    const block_id view_id = storage_manager_->createViewBlock(scattered_id, first, rows_p);
    BlockReference view = storage_manager_->getBlock(view_id);
    for (std::size_t a = 0; a < relation_->size(); ++a) {
      std::uint64_t *dst = view->nullBitmap(static_cast<attribute_id>(a));
      if (dst == nullptr) continue;
      const std::uint64_t *bits = src->nullBitmap(static_cast<attribute_id>(a));
      const std::int64_t zero_row = 0;
      CheckStatus(qsx_bitmap_gather_segmented(1, &bits, &zero_row, static_cast<const std::int32_t *>(rows_scattered->ptr) + first, rows_p, dst,
                                              CurrentStream()), "qsx_bitmap_gather_segmented");
    }
    view->setFirstRow(storage_manager_->reserveRows(relation_->getID(), rows_p));
    made.emplace_back(view_id, p);
  }
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");   // before the source block goes
  storage_manager_->deleteBlockOrBlobFile(id);
  storage_manager_->deleteBlockOrBlobFile(scattered_id);      // (the views keep the scattered block alive)
  std::lock_guard<std::mutex> lock(mutex_);
  for (const auto &m : made) {
    touched_.push_back(TouchedBlock{m.first, m.second});
    relation_->addBlockToPartition(m.first, m.second);
  }
}
std::vector<block_id> InsertDestination::getTouchedBlocks() const {
  std::lock_guard<std::mutex> lock(mutex_);
  std::vector<block_id> ids;
  for (const TouchedBlock &t : touched_) ids.push_back(t.id);
  return ids;
}
std::vector<InsertDestination::TouchedBlock> InsertDestination::getTouchedBlocksWithPartitions() const {
  std::lock_guard<std::mutex> lock(mutex_);
  return touched_;
}

// ---------------------------------------------------------------------------
// AggregationOperationState
// ---------------------------------------------------------------------------
struct AggregationOperationState::Distinctify {
  std::size_t agg_index = 0;            // position in spec.aggregates
  std::vector<attribute_id> attrs;      // group-by..., argument (kInvalidAttributeID: the argument is `expression`)
  std::vector<Type> types;
  ScalarPtr expression;                 // DISTINCT over an arithmetic expression: evaluated per block into a DOUBLE column
  std::mutex mutex;                     // many AggregationWorkOrders append concurrently
  struct Chunk {
    std::vector<std::unique_ptr<DeviceBuffer>> cols;
    std::int64_t rows = 0;
  };
  std::vector<Chunk> chunks;
};

namespace {
qsx_agg_fn_t AggFn(AggregationID id) {
  switch (id) {
    case AggregationID::kCount: return QSX_AGG_COUNT_STAR;
    case AggregationID::kSum: return QSX_AGG_SUM;
    case AggregationID::kAvg: return QSX_AGG_AVG;
    case AggregationID::kMin: return QSX_AGG_MIN;
    default: return QSX_AGG_MAX;
  }
}
// Result type of an aggregate (AggregationHandle{Count,Sum,Avg,Min,Max}::getResultType).
Type AggResultType(AggregationID id, const Type &argument) {
  switch (id) {
    case AggregationID::kCount: return Type::Long();
    case AggregationID::kSum: return (argument.id == kInt || argument.id == kLong) ? Type::Long() : Type::Double();
    case AggregationID::kAvg: return Type::Double();
    default: return argument;
  }
}
}  // namespace

TypeID ScalarResultType(const ScalarPtr &scalar, const CatalogRelation &relation) {
  if (scalar == nullptr) throw ExecutionError("ScalarResultType: null scalar", QSX_ERR_INVALID_ARGUMENT);
  switch (scalar->kind) {
    case Scalar::kAttribute: {
      const TypeID t = relation.getAttributeType(scalar->attribute).id;
      if (t != kInt && t != kLong && t != kFloat && t != kDouble) {
        throw ExecutionError("arithmetic over a non-numeric attribute", QSX_ERR_UNSUPPORTED);
      }
      return t == kFloat ? kDouble : t;   // (FLOAT operands are evaluated in double)
    }
    case Scalar::kLiteral:
      return scalar->literal_type;
    default: {
      const TypeID l = ScalarResultType(scalar->left, relation), r = ScalarResultType(scalar->right, relation);
      if (l == kDouble || r == kDouble) return kDouble;
      return l == kLong || r == kLong ? kLong : kInt;
    }
  }
}

qsx_operand_t ExpressionFlattener::add(const ScalarPtr &scalar) {
  if (scalar == nullptr) throw ExecutionError("ExpressionFlattener: null scalar", QSX_ERR_INVALID_ARGUMENT);
  switch (scalar->kind) {
    case Scalar::kAttribute:
      return qsx_operand_t{QSX_OPD_COLUMN, column_of_(scalar->attribute)};
    case Scalar::kLiteral: {
      for (std::size_t i = 0; i < consts_.size(); ++i) {
        if (std::memcmp(&consts_[i], &scalar->literal, sizeof(double)) == 0) return qsx_operand_t{QSX_OPD_CONST, static_cast<std::int32_t>(i)};
      }
      if (consts_.size() >= QSX_MAX_CONSTS) throw ExecutionError("expression: too many distinct literals", QSX_ERR_UNSUPPORTED);
      consts_.push_back(scalar->literal);
      return qsx_operand_t{QSX_OPD_CONST, static_cast<std::int32_t>(consts_.size() - 1)};
    }
    default: {
      const qsx_operand_t a = add(scalar->left), b = add(scalar->right);
      const std::int32_t op = static_cast<std::int32_t>(scalar->operation);   // kAdd .. kDivide = QSX_EX_ADD .. QSX_EX_DIV
      for (const qsx_expr_instr_t &in : instrs_) {   // the same node again (shared subexpression): its temp
        if (in.op == op && in.a.kind == a.kind && in.a.index == a.index && in.b.kind == b.kind && in.b.index == b.index) {
          return qsx_operand_t{QSX_OPD_TEMP, in.dst};
        }
      }
      if (instrs_.size() >= QSX_MAX_INSTRS || instrs_.size() >= QSX_MAX_TEMPS) {
        throw ExecutionError("expression: more nodes than the kernel's program holds", QSX_ERR_UNSUPPORTED);
      }
      qsx_expr_instr_t in;
      in.op = op;
      in.dst = static_cast<std::int32_t>(instrs_.size());   // one temp per node
      in.a = a;
      in.b = b;
      instrs_.push_back(in);
      return qsx_operand_t{QSX_OPD_TEMP, in.dst};
    }
  }
}

AggregationOperationState::AggregationOperationState(const AggregationStateSpec &spec) : spec_(spec) {
  // the library must have been built from the header this file was compiled against
  if (qsx_abi_version() != QSX_ABI_VERSION || qsx_abi_sizeof_agg_config() != sizeof(qsx_agg_config_t)) {
    throw ExecutionError("libqsx.so and include/qsx.h disagree on the ABI version / qsx_agg_config_t", QSX_ERR_INVALID_ARGUMENT);
  }
  std::memset(&config_, 0, sizeof(config_));
  const CatalogRelation &rel = *spec.input_relation;
  auto column_of = [&](attribute_id attr) -> int {
    for (std::size_t i = 0; i < column_attr_.size(); ++i) {
      if (column_attr_[i] == attr) return static_cast<int>(i);
    }
    if (column_attr_.size() >= QSX_MAX_COLUMNS) throw ExecutionError("AggregationOperationState: too many columns", QSX_ERR_UNSUPPORTED);
    const Type &t = rel.getAttributeType(attr);
    config_.column_type[column_attr_.size()] = t.id;
    config_.column_width[column_attr_.size()] = t.width;
    config_.column_nullable[column_attr_.size()] = t.nullable ? 1 : 0;
    column_attr_.push_back(attr);
    return static_cast<int>(column_attr_.size() - 1);
  };
  config_.strategy = spec.group_by.empty() ? QSX_AGG_SINGLE_STATE : spec.strategy;
  config_.num_keys = static_cast<int>(spec.group_by.size());
  for (std::size_t k = 0; k < spec.group_by.size(); ++k) config_.key_column[k] = column_of(spec.group_by[k]);
  int num_main = 0;
  ExpressionFlattener flattener(column_of);
  for (std::size_t a = 0; a < spec_.aggregates.size(); ++a) {
    if (spec_.aggregates[a].argument_expression != nullptr && spec_.aggregates[a].argument_expression->kind == Scalar::kAttribute) {
      spec_.aggregates[a].argument = spec_.aggregates[a].argument_expression->attribute;   // ScalarAttribute: the plain form
      spec_.aggregates[a].argument_expression = nullptr;
    }
    const AggregateSpec &ag = spec_.aggregates[a];
    if (ag.argument_expression != nullptr && ag.is_distinct) {
      // DISTINCT over an arithmetic expression (Distinct.test:58-72 COUNT(DISTINCT x % y) is such a query): the distinctify
      // key is (group-by..., value of the expression); the value column is computed per block (qsx_eval_expression)
      if (spec.group_by.size() + 1 > QSX_MAX_KEYS) throw ExecutionError("DISTINCT aggregate: too many group-by attributes", QSX_ERR_UNSUPPORTED);
      std::unique_ptr<Distinctify> d(new Distinctify);
      d->agg_index = a;
      d->attrs = spec.group_by;
      d->attrs.push_back(kInvalidAttributeID);
      for (attribute_id attr : spec.group_by) d->types.push_back(rel.getAttributeType(attr));
      d->types.push_back(Type::Double());
      d->expression = ag.argument_expression;
      distinctify_.push_back(std::move(d));
      main_agg_.push_back(-1);
      continue;
    }
    if (ag.argument_expression != nullptr) {
      // an arithmetic expression as the aggregate's argument: part of the state's expression program
      if (ag.function == AggregationID::kCount) throw ExecutionError("COUNT over an expression: pass COUNT(*)", QSX_ERR_UNSUPPORTED);
      const qsx_operand_t value = flattener.add(ag.argument_expression);
      if (value.kind == QSX_OPD_CONST) throw ExecutionError("aggregate over a literal", QSX_ERR_UNSUPPORTED);
      config_.aggs[num_main].fn = AggFn(ag.function);
      config_.aggs[num_main].arg = value;
      main_agg_.push_back(num_main++);
      continue;
    }
    if (ag.is_distinct) {
      // "Initialize the corresponding distinctify hash table if this is a DISTINCT aggregation" (:172-207):
      // key types = group-by types + argument types
      if (ag.argument == kInvalidAttributeID) throw ExecutionError("DISTINCT aggregate without an argument", QSX_ERR_INVALID_ARGUMENT);
      if (spec.group_by.size() + 1 > QSX_MAX_KEYS) throw ExecutionError("DISTINCT aggregate: too many group-by attributes", QSX_ERR_UNSUPPORTED);
      std::unique_ptr<Distinctify> d(new Distinctify);
      d->agg_index = a;
      d->attrs = spec.group_by;
      d->attrs.push_back(ag.argument);
      for (attribute_id attr : d->attrs) d->types.push_back(rel.getAttributeType(attr));
      distinctify_.push_back(std::move(d));
      main_agg_.push_back(-1);
      continue;
    }
    config_.aggs[num_main].fn = AggFn(ag.function);
    if (ag.function != AggregationID::kCount) {
      config_.aggs[num_main].arg.kind = QSX_OPD_COLUMN;
      config_.aggs[num_main].arg.index = column_of(ag.argument);
    } else if (ag.argument != kInvalidAttributeID && rel.getAttributeType(ag.argument).nullable) {
      // COUNT(x) over a nullable x counts the non-NULL values (AggregationHandleCount<false, true>); over a
      // non-nullable x it is COUNT(*)
      config_.aggs[num_main].fn = QSX_AGG_COUNT;
      config_.aggs[num_main].arg.kind = QSX_OPD_COLUMN;
      config_.aggs[num_main].arg.index = column_of(ag.argument);
    }
    main_agg_.push_back(num_main++);
  }
  config_.num_aggs = num_main;
  config_.num_instrs = static_cast<int>(flattener.instrs().size());
  for (std::size_t k = 0; k < flattener.instrs().size(); ++k) config_.instrs[k] = flattener.instrs()[k];
  for (std::size_t k = 0; k < flattener.consts().size(); ++k) config_.consts[k] = flattener.consts()[k];
  if (spec.predicate != nullptr) {
    int in_state = 0;
    for (const ComparisonPredicate &term : spec.predicate->conjuncts) {
      const Type &t = rel.getAttributeType(term.attribute);
      if (t.id == kChar || term.rhs_attribute != kInvalidAttributeID || in_state == QSX_MAX_PRED_TERMS) {
        external_predicate_.conjuncts.push_back(term);   // string comparisons, attribute-vs-attribute, overflow
        continue;
      }
      config_.pred[in_state].column = column_of(term.attribute);
      config_.pred[in_state].op = static_cast<int>(term.comparison);
      std::memcpy(&config_.pred[in_state].literal, &term.literal.v, sizeof(term.literal.v));
      ++in_state;
    }
    config_.num_pred_terms = in_state;
  }
  config_.num_columns = static_cast<int>(column_attr_.size());
  config_.est_groups = spec.estimated_num_groups;
  config_.num_entries = spec.collision_free_num_entries;
  // all_distinct_ (:126-127, 620-628): no upsert into the final table per block, it is filled from the distinctify tables
  if (num_main > 0 || distinctify_.empty()) CheckStatus(qsx_agg_state_create(&config_, &state_), "qsx_agg_state_create");
}

AggregationOperationState::~AggregationOperationState() {
  if (state_ != nullptr) qsx_agg_state_destroy(state_);
  if (coded_state_ != nullptr) qsx_agg_state_destroy(coded_state_);
}

void AggregationOperationState::aggregateBlock(const StorageBlock &block, const std::uint64_t *lip_filter) {
  const std::int64_t n = block.numTuples();
  if (!distinctify_.empty() && n > 0) {
    // insertValueAccessorIntoDistinctifyHashTable per DISTINCT aggregate (:522-528, 600-628), on the tuples that pass
    // the state's predicate and the LIP filters: the block's distinct (group-by..., argument) tuples are appended
    std::int64_t matches = n;
    void *selected = nullptr;
    if (spec_.predicate != nullptr) selected = spec_.predicate->getMatchesForBlock(block, &matches, lip_filter);
    const std::uint64_t *filter = selected != nullptr ? static_cast<const std::uint64_t *>(selected) : lip_filter;
    const std::size_t ws_bytes = qsx_sort_workspace_bytes(n);
    DeviceBuffer ws(ws_bytes), tids(static_cast<std::size_t>(n) * 4 + 16), count(8);
    for (auto &d : distinctify_) {
      const void *cols[QSX_MAX_KEYS];
      std::int32_t types[QSX_MAX_KEYS];
      std::unique_ptr<DeviceBuffer> expression_values;
      std::vector<attribute_id> nullable_sources;   // attributes whose NULLs keep a tuple out of the table
      for (std::size_t c = 0; c < d->attrs.size(); ++c) {
        types[c] = d->types[c].id;
        if (d->attrs[c] != kInvalidAttributeID) {
          cols[c] = block.stripe(d->attrs[c]);
          nullable_sources.push_back(d->attrs[c]);
          continue;
        }
        std::vector<attribute_id> attrs;
        ExpressionFlattener flattener([&](attribute_id a) {
          for (std::size_t k = 0; k < attrs.size(); ++k) if (attrs[k] == a) return static_cast<int>(k);
          attrs.push_back(a);
          return static_cast<int>(attrs.size() - 1);
        });
        const qsx_operand_t result = flattener.add(d->expression);
        const void *in_cols[QSX_MAX_COLUMNS];
        std::int32_t in_types[QSX_MAX_COLUMNS];
        if (attrs.size() > QSX_MAX_COLUMNS) throw ExecutionError("DISTINCT: expression over too many attributes", QSX_ERR_UNSUPPORTED);
        for (std::size_t k = 0; k < attrs.size(); ++k) {
          in_cols[k] = block.stripe(attrs[k]);
          in_types[k] = block.getRelation().getAttributeType(attrs[k]).id;
          nullable_sources.push_back(attrs[k]);   // a NULL operand makes the value NULL
        }
        double consts[QSX_MAX_CONSTS] = {};
        for (std::size_t k = 0; k < flattener.consts().size(); ++k) consts[k] = flattener.consts()[k];
        expression_values.reset(new DeviceBuffer(static_cast<std::size_t>(n) * 8 + 8));
        CheckStatus(qsx_eval_expression(static_cast<int>(attrs.size()), in_cols, in_types, static_cast<int>(flattener.instrs().size()),
                                        flattener.instrs().data(), consts, result, n, static_cast<double *>(expression_values->ptr),
                                        CurrentStream()), "qsx_eval_expression");
        cols[c] = expression_values->ptr;
      }
      // the distinctify table is keyed by (group-by..., argument): a tuple with a NULL in any of them is not inserted
      // (PackedPayloadHashTable.hpp:861-867)
      const std::unique_ptr<DeviceBuffer> not_null = NotNullFilter(block, nullable_sources, filter);
      const std::uint64_t *distinct_filter = not_null != nullptr ? static_cast<const std::uint64_t *>(not_null->ptr) : filter;
      CheckStatus(qsx_distinct_rows(static_cast<int>(d->attrs.size()), cols, types, n, distinct_filter, static_cast<std::int32_t *>(tids.ptr),
                                    static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()), "qsx_distinct_rows");
      Distinctify::Chunk chunk;
      chunk.rows = ReadCount(count.ptr);
      if (chunk.rows == 0) continue;
      for (std::size_t c = 0; c < d->attrs.size(); ++c) {
        chunk.cols.emplace_back(new DeviceBuffer(static_cast<std::size_t>(chunk.rows) * d->types[c].width + 16));
        CheckStatus(qsx_gather(d->types[c].width, cols[c], static_cast<const std::int32_t *>(tids.ptr), chunk.rows,
                               chunk.cols.back()->ptr, CurrentStream()), "qsx_gather");
      }
      CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
      std::lock_guard<std::mutex> lock(d->mutex);
      d->chunks.push_back(std::move(chunk));
    }
    qsx_device_free(selected);
  }
  if (state_ == nullptr) return;
  // the conjuncts the kernel does not evaluate: a TupleIdSequence like the one a SelectOperator computes, used as the filter
  struct OwnedBitmap {
    void *ptr = nullptr;
    ~OwnedBitmap() { if (ptr != nullptr) qsx_device_free(ptr); }
  } external_matches;
  if (!external_predicate_.conjuncts.empty() && n > 0) {
    std::int64_t matches = 0;
    external_matches.ptr = external_predicate_.getMatchesForBlock(block, &matches, lip_filter);
    lip_filter = static_cast<const std::uint64_t *>(external_matches.ptr);
  }
  // A block with compressed operand attributes whose values have not been materialised: aggregate on the codes.
  // (Key and predicate columns of the state take the value path here: stripe() decodes them once per block.)
  // null bitmaps of the nullable operand attributes (a block may hold none: loadBlock without bitmaps)
  const std::uint64_t *nulls[QSX_MAX_COLUMNS] = {};
  bool any_nulls = false;
  for (std::size_t i = 0; i < column_attr_.size(); ++i) {
    nulls[i] = config_.column_nullable[i] != 0 ? block.nullBitmap(column_attr_[i]) : nullptr;
    any_nulls = any_nulls || nulls[i] != nullptr;
  }
  int code_width[QSX_MAX_COLUMNS] = {};
  bool any_coded = false;
  for (std::size_t i = 0; i < column_attr_.size() && !any_nulls; ++i) {
    const CompressedAttribute *ca = block.compressedAttribute(column_attr_[i]);
    const int type = config_.column_type[i];
    if (ca != nullptr && type != kChar && !block.valuesMaterialized(column_attr_[i])) {
      code_width[i] = ca->code_width;
      any_coded = true;
    }
  }
  if (any_coded && n > 0) {
    bool use_coded = false;
    {
      std::lock_guard<std::mutex> lock(coded_mutex_);
      if (coded_state_ == nullptr && !coded_merged_) {
        coded_config_ = config_;
        for (std::size_t i = 0; i < column_attr_.size(); ++i) coded_config_.column_code_width[i] = code_width[i];
        CheckStatus(qsx_agg_state_create(&coded_config_, &coded_state_), "qsx_agg_state_create");
      }
      if (coded_state_ != nullptr && !coded_merged_) {
        use_coded = true;
        for (std::size_t i = 0; i < column_attr_.size(); ++i) use_coded = use_coded && coded_config_.column_code_width[i] == code_width[i];
      }
    }
    if (use_coded) {
      const void *cols[QSX_MAX_COLUMNS];
      const void *dicts[QSX_MAX_COLUMNS];
      for (std::size_t i = 0; i < column_attr_.size(); ++i) {
        const CompressedAttribute *ca = code_width[i] != 0 ? block.compressedAttribute(column_attr_[i]) : nullptr;
        cols[i] = ca != nullptr ? ca->codes : block.stripe(column_attr_[i]);
        dicts[i] = ca != nullptr && ca->kind == CompressedAttribute::kDictionary ? ca->dictionary : nullptr;
      }
      CheckStatus(qsx_agg_update_coded(coded_state_, cols, dicts, n, lip_filter, CurrentStream()), "qsx_agg_update_coded");
      ++coded_blocks_;
      CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
      return;
    }
  }
  const void *cols[QSX_MAX_COLUMNS];
  for (std::size_t i = 0; i < column_attr_.size(); ++i) cols[i] = block.stripe(column_attr_[i]);
  if (any_nulls) {
    CheckStatus(qsx_agg_update_nullable(state_, cols, nulls, n, lip_filter, CurrentStream()), "qsx_agg_update_nullable");
  } else {
    CheckStatus(qsx_agg_update(state_, cols, n, lip_filter, CurrentStream()), "qsx_agg_update");
  }
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
}

void AggregationOperationState::aggregateBlocks(const std::vector<BlockReference> &blocks,
                                                const std::vector<const std::uint64_t *> &lip_filters) {
  std::vector<std::int64_t> rows;
  std::vector<const void *> cols;
  std::vector<const std::uint64_t *> filters;
  bool any_filter = false;
  std::vector<std::int64_t> coded_rows;
  std::vector<const void *> coded_cols, coded_dicts;
  std::vector<const std::uint64_t *> coded_filters;
  bool any_coded_filter = false;
  const bool state_allows = state_ != nullptr && distinctify_.empty() && external_predicate_.conjuncts.empty();
  for (std::size_t i = 0; i < blocks.size(); ++i) {
    const StorageBlock &block = *blocks[i];
    const std::uint64_t *filter = i < lip_filters.size() ? lip_filters[i] : nullptr;
    bool in_run = state_allows && block.numTuples() > 0;
    int code_width[QSX_MAX_COLUMNS] = {};
    bool any_coded = false;
    for (std::size_t c = 0; c < column_attr_.size() && in_run; ++c) {
      // null bitmaps travel with single-block calls (qsx_agg_update_nullable)
      if (block.nullBitmap(column_attr_[c]) != nullptr) in_run = false;
      const CompressedAttribute *ca = block.compressedAttribute(column_attr_[c]);
      if (ca != nullptr && config_.column_type[c] != kChar && !block.valuesMaterialized(column_attr_[c])) {
        code_width[c] = ca->code_width;          // aggregated on its codes, like aggregateBlock does
        any_coded = true;
      }
    }
    if (in_run && any_coded) {
      // compressed blocks: one run through the state over code stripes (created by the first such block, which fixes the code
      // widths; a block that compressed differently goes block by block)
      bool use_coded = false;
      {
        std::lock_guard<std::mutex> lock(coded_mutex_);
        if (coded_state_ == nullptr && !coded_merged_) {
          coded_config_ = config_;
          for (std::size_t c = 0; c < column_attr_.size(); ++c) coded_config_.column_code_width[c] = code_width[c];
          CheckStatus(qsx_agg_state_create(&coded_config_, &coded_state_), "qsx_agg_state_create");
        }
        if (coded_state_ != nullptr && !coded_merged_) {
          use_coded = true;
          for (std::size_t c = 0; c < column_attr_.size(); ++c) use_coded = use_coded && coded_config_.column_code_width[c] == code_width[c];
        }
      }
      if (!use_coded) {
        aggregateBlock(block, filter);
        continue;
      }
      coded_rows.push_back(block.numTuples());
      for (std::size_t c = 0; c < column_attr_.size(); ++c) {
        const CompressedAttribute *ca = code_width[c] != 0 ? block.compressedAttribute(column_attr_[c]) : nullptr;
        coded_cols.push_back(ca != nullptr ? ca->codes : block.stripe(column_attr_[c]));
        coded_dicts.push_back(ca != nullptr && ca->kind == CompressedAttribute::kDictionary ? ca->dictionary : nullptr);
      }
      coded_filters.push_back(filter);
      any_coded_filter = any_coded_filter || filter != nullptr;
      continue;
    }
    if (!in_run) {
      aggregateBlock(block, filter);
      continue;
    }
    rows.push_back(block.numTuples());
    for (std::size_t c = 0; c < column_attr_.size(); ++c) cols.push_back(block.stripe(column_attr_[c]));
    filters.push_back(filter);
    any_filter = any_filter || filter != nullptr;
  }
  if (!coded_rows.empty()) {
    CheckStatus(qsx_agg_update_coded_blocks(coded_state_, static_cast<int>(coded_rows.size()), coded_rows.data(), coded_cols.data(),
                                            coded_dicts.data(), any_coded_filter ? coded_filters.data() : nullptr, CurrentStream()),
                "qsx_agg_update_coded_blocks");
    coded_blocks_ += static_cast<int>(coded_rows.size());
    CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  }
  if (rows.empty()) return;
  CheckStatus(qsx_agg_update_blocks(state_, static_cast<int>(rows.size()), rows.data(), cols.data(), any_filter ? filters.data() : nullptr,
                                    CurrentStream()), "qsx_agg_update_blocks");
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
}

// finalizeAggregate with DISTINCT aggregates: every distinctify table is reduced to its distinct tuples, which are
// aggregated once each into a table keyed like the final one (aggregateOnDistinctifyHashTableFor{Single,GroupBy},
// AggregationOperationState.cpp:652-670, 720-760); the per-aggregate results are then lined up on the group key.
void AggregationOperationState::finalizeWithDistinct(InsertDestination *dest) {
  const CatalogRelation &rel = *spec_.input_relation;
  const int nk = config_.num_keys;
  struct ResultSet {
    std::vector<std::unique_ptr<DeviceBuffer>> keys, vals;
    std::int64_t rows = 0;
    std::unique_ptr<DeviceBuffer> order;      // row numbers in ascending key order
  };
  auto finalize_into = [&](qsx_agg_state_t *state, const qsx_agg_config_t &cfg, const std::vector<Type> &val_types, ResultSet *out) {
    std::int64_t groups = 0;
    CheckStatus(qsx_agg_num_groups(state, &groups, CurrentStream()), "qsx_agg_num_groups");
    const std::int64_t cap = groups > 0 ? groups : 1;
    void *key_cols[QSX_MAX_KEYS];
    void *val_cols[QSX_MAX_AGGS];
    for (int k = 0; k < cfg.num_keys; ++k) {
      out->keys.emplace_back(new DeviceBuffer(static_cast<std::size_t>(cap) * cfg.column_width[cfg.key_column[k]] + 16));
      key_cols[k] = out->keys.back()->ptr;
    }
    for (int a = 0; a < cfg.num_aggs; ++a) {
      out->vals.emplace_back(new DeviceBuffer(static_cast<std::size_t>(cap) * val_types[a].width + 16));
      val_cols[a] = out->vals.back()->ptr;
    }
    DeviceBuffer rows(8);
    CheckStatus(qsx_agg_finalize(state, 0, 1, key_cols, val_cols, nullptr, cap, static_cast<std::int64_t *>(rows.ptr), CurrentStream()),
                "qsx_agg_finalize");
    out->rows = ReadCount(rows.ptr);
    if (out->rows == QSX_GROUPS_HASH_COLLISION) throw ExecutionError("qsx_agg_finalize: wide group-by key", QSX_ERR_HASH_COLLISION);
    if (cfg.num_keys > 0 && out->rows > 0) {   // ascending key order: the common order of all result sets
      const void *cols[QSX_MAX_KEYS];
      std::int32_t types[QSX_MAX_KEYS];
      for (int k = 0; k < cfg.num_keys; ++k) {
        cols[k] = out->keys[k]->ptr;
        types[k] = cfg.column_type[cfg.key_column[k]];
        if (types[k] == kChar && cfg.column_width[cfg.key_column[k]] != 1) {
          throw ExecutionError("DISTINCT aggregate: CHAR group-by keys wider than one byte", QSX_ERR_UNSUPPORTED);
        }
      }
      const std::size_t ws_bytes = qsx_sort_workspace_bytes(out->rows);
      DeviceBuffer ws(ws_bytes);
      out->order.reset(new DeviceBuffer(static_cast<std::size_t>(out->rows) * 4 + 16));
      CheckStatus(qsx_sort_permutation(cfg.num_keys, cols, types, nullptr, out->rows, static_cast<std::int32_t *>(out->order->ptr), ws.ptr,
                                       ws_bytes, CurrentStream()), "qsx_sort_permutation");
    }
  };

  // the non-DISTINCT aggregates
  ResultSet main;
  std::vector<Type> main_types;
  if (state_ != nullptr) {
    for (std::size_t a = 0; a < spec_.aggregates.size(); ++a) {
      if (main_agg_[a] < 0) continue;
      const AggregateSpec &ag = spec_.aggregates[a];
      main_types.push_back(AggResultType(ag.function, ag.argument_expression != nullptr ? Type::Double()   // expressions evaluate in DOUBLE
                                                      : ag.argument == kInvalidAttributeID ? Type::Long() : rel.getAttributeType(ag.argument)));
    }
    finalize_into(state_, config_, main_types, &main);
  }
  // one result set per DISTINCT aggregate
  std::vector<ResultSet> distinct(distinctify_.size());
  for (std::size_t i = 0; i < distinctify_.size(); ++i) {
    Distinctify &d = *distinctify_[i];
    const AggregateSpec &ag = spec_.aggregates[d.agg_index];
    const std::size_t ncols = d.attrs.size();
    std::int64_t total = 0;
    for (const auto &chunk : d.chunks) total += chunk.rows;
    // all block-level tuples side by side, then distinct over the whole input
    std::vector<std::unique_ptr<DeviceBuffer>> all, tuples;
    for (std::size_t c = 0; c < ncols; ++c) {
      all.emplace_back(new DeviceBuffer(static_cast<std::size_t>(total) * d.types[c].width + 16));
      char *at = static_cast<char *>(all.back()->ptr);
      for (const auto &chunk : d.chunks) {
        const std::size_t bytes = static_cast<std::size_t>(chunk.rows) * d.types[c].width;
        CheckStatus(qsx_copy_on_device(at, chunk.cols[c]->ptr, bytes, CurrentStream()), "qsx_copy_on_device");
        at += bytes;
      }
    }
    std::int64_t rows = 0;
    if (total > 0) {
      const void *cols[QSX_MAX_KEYS];
      std::int32_t types[QSX_MAX_KEYS];
      for (std::size_t c = 0; c < ncols; ++c) { cols[c] = all[c]->ptr; types[c] = d.types[c].id; }
      const std::size_t ws_bytes = qsx_sort_workspace_bytes(total);
      DeviceBuffer ws(ws_bytes), tids(static_cast<std::size_t>(total) * 4 + 16), count(8);
      CheckStatus(qsx_distinct_rows(static_cast<int>(ncols), cols, types, total, nullptr, static_cast<std::int32_t *>(tids.ptr),
                                    static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()), "qsx_distinct_rows");
      rows = ReadCount(count.ptr);
      for (std::size_t c = 0; c < ncols; ++c) {
        tuples.emplace_back(new DeviceBuffer(static_cast<std::size_t>(rows) * d.types[c].width + 16));
        CheckStatus(qsx_gather(d.types[c].width, cols[c], static_cast<const std::int32_t *>(tids.ptr), rows, tuples.back()->ptr,
                               CurrentStream()), "qsx_gather");
      }
    }
    // the aggregate over the distinct tuples, keyed like the final table
    qsx_agg_config_t cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.strategy = config_.strategy;
    cfg.num_columns = static_cast<int>(ncols);
    for (std::size_t c = 0; c < ncols; ++c) {
      cfg.column_type[c] = d.types[c].id;
      cfg.column_width[c] = d.types[c].width;
    }
    cfg.num_keys = nk;
    for (int k = 0; k < nk; ++k) cfg.key_column[k] = k;
    cfg.num_aggs = 1;
    cfg.aggs[0].fn = AggFn(ag.function);             // COUNT(DISTINCT x) = COUNT(*) over the distinct tuples (no NULLs)
    if (ag.function != AggregationID::kCount) {
      cfg.aggs[0].arg.kind = QSX_OPD_COLUMN;
      cfg.aggs[0].arg.index = nk;
    }
    cfg.est_groups = config_.est_groups;
    cfg.num_entries = config_.num_entries;
    qsx_agg_state_t *state = nullptr;
    CheckStatus(qsx_agg_state_create(&cfg, &state), "qsx_agg_state_create");
    try {
      if (rows > 0) {
        const void *cols[QSX_MAX_KEYS];
        for (std::size_t c = 0; c < ncols; ++c) cols[c] = tuples[c]->ptr;
        CheckStatus(qsx_agg_update(state, cols, rows, nullptr, CurrentStream()), "qsx_agg_update");
      }
      finalize_into(state, cfg, {AggResultType(ag.function, d.types.back())}, &distinct[i]);
    } catch (...) {
      qsx_agg_state_destroy(state);
      throw;
    }
    CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
    qsx_agg_state_destroy(state);
  }
  // every result set holds the same groups (each group has at least one tuple in every table)
  const ResultSet &first = state_ != nullptr ? main : distinct.front();
  for (const ResultSet &r : distinct) {
    if (r.rows != first.rows) throw ExecutionError("DISTINCT aggregate: group sets differ", QSX_ERR_INVALID_ARGUMENT);
  }
  block_id id;
  BlockReference out = dest->getBlockForInsertion(first.rows > 0 ? first.rows : 1, &id);
  auto emit = [&](const ResultSet &r, const void *src, int width, attribute_id out_attr) {
    if (r.rows == 0) return;
    if (r.order != nullptr) {
      CheckStatus(qsx_gather(width, src, static_cast<const std::int32_t *>(r.order->ptr), r.rows, out->stripe(out_attr), CurrentStream()),
                  "qsx_gather");
    } else {
      CheckStatus(qsx_copy_on_device(out->stripe(out_attr), src, static_cast<std::size_t>(r.rows) * width, CurrentStream()),
                  "qsx_copy_on_device");
    }
  };
  for (int k = 0; k < nk; ++k) emit(first, first.keys[k]->ptr, config_.column_width[config_.key_column[k]], k);
  std::size_t next_distinct = 0;
  for (std::size_t a = 0; a < spec_.aggregates.size(); ++a) {
    const attribute_id out_attr = static_cast<attribute_id>(nk + a);
    if (main_agg_[a] >= 0) {
      emit(main, main.vals[main_agg_[a]]->ptr, main_types[main_agg_[a]].width, out_attr);
    } else {
      const ResultSet &r = distinct[next_distinct];
      emit(r, r.vals[0]->ptr, AggResultType(spec_.aggregates[a].function, distinctify_[next_distinct]->types.back()).width, out_attr);
      ++next_distinct;
    }
  }
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  dest->returnBlock(id, first.rows);
}

void AggregationOperationState::buildExistenceMap(const StorageBlock &block, attribute_id build_attribute, const Type &type) {
  if (type.id != kInt && type.id != kLong) {   // LOG(FATAL) "Build attribute type not supported" (:203-206)
    throw ExecutionError("BuildAggregationExistenceMapOperator: build attribute must be INT or LONG", QSX_ERR_UNSUPPORTED);
  }
  CheckStatus(qsx_agg_mark_existence(state_, type.id, block.stripe(build_attribute), block.numTuples(), nullptr, CurrentStream()),
              "qsx_agg_mark_existence");
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
}

void AggregationOperationState::finalizeAggregate(std::size_t partition, std::size_t num_partitions, InsertDestination *dest) {
  {   // the state fed by compressed blocks joins the other one (same image layout: mergeFrom semantics)
    std::lock_guard<std::mutex> lock(coded_mutex_);
    if (coded_state_ != nullptr && !coded_merged_) {
      CheckStatus(qsx_agg_merge(state_, coded_state_, CurrentStream()), "qsx_agg_merge");
      CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
    }
    coded_merged_ = true;
  }
  if (!distinctify_.empty()) {
    // the distinctify tables are drained by one work order; the others of a partitioned finalize have nothing to emit
    if (partition == 0) finalizeWithDistinct(dest);
    return;
  }
  std::int64_t groups = 0;
  CheckStatus(qsx_agg_num_groups(state_, &groups, CurrentStream()), "qsx_agg_num_groups");
  block_id id;
  BlockReference out = dest->getBlockForInsertion(groups > 0 ? groups : 1, &id);
  void *key_cols[QSX_MAX_KEYS];
  void *val_cols[QSX_MAX_AGGS];
  std::uint8_t *null_cols[QSX_MAX_AGGS] = {};
  std::vector<std::unique_ptr<DeviceBuffer>> null_flags;
  for (int k = 0; k < config_.num_keys; ++k) key_cols[k] = out->stripe(k);
  for (int a = 0; a < config_.num_aggs; ++a) {
    val_cols[a] = out->stripe(config_.num_keys + a);
    // result types of SUM / AVG / MIN / MAX are nullable (AggregationHandleSum::getResultType ...->getNullableVersion()):
    // an output attribute declared nullable receives the NULL flags as its null bitmap
    if (out->nullBitmap(static_cast<attribute_id>(config_.num_keys + a)) != nullptr) {
      null_flags.emplace_back(new DeviceBuffer(static_cast<std::size_t>(out->capacity()) + 16));
      null_cols[a] = static_cast<std::uint8_t *>(null_flags.back()->ptr);
    }
  }
  DeviceBuffer rows(8);
  CheckStatus(qsx_agg_finalize(state_, static_cast<int>(partition), static_cast<int>(num_partitions), key_cols, val_cols,
                               null_flags.empty() ? nullptr : null_cols, out->capacity(), static_cast<std::int64_t *>(rows.ptr),
                               CurrentStream()),
              "qsx_agg_finalize");
  const std::int64_t written = ReadCount(rows.ptr);
  // a key wider than 8 bytes is grouped by its 64-bit hash and verified: two keys under one hash void the result
  if (written == QSX_GROUPS_HASH_COLLISION) throw ExecutionError("qsx_agg_finalize: wide group-by key", QSX_ERR_HASH_COLLISION);
  if (written > out->capacity()) throw ExecutionError("qsx_agg_finalize: more groups than the output block holds", QSX_ERR_CAPACITY);
  for (int a = 0; a < config_.num_aggs && written > 0; ++a) {
    if (null_cols[a] == nullptr) continue;
    // byte flags -> TupleIdSequence-ordered bitmap: a scan of 1-byte "codes" for flag >= 1
    CheckStatus(qsx_select_codes(1, null_cols[a], written, QSX_CODE_GE, 1, 0, nullptr,
                                 out->nullBitmap(static_cast<attribute_id>(config_.num_keys + a)), static_cast<std::int64_t *>(rows.ptr),
                                 CurrentStream()), "qsx_select_codes(null flags)");
  }
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  dest->returnBlock(id, written);
}

// ---------------------------------------------------------------------------
// QueryContext
// ---------------------------------------------------------------------------
QueryContext::~QueryContext() {
  for (auto &parts : join_tables_) {
    for (qsx_join_table_t *t : parts) qsx_join_table_destroy(t);
  }
  for (qsx_lip_filter_t *f : lip_filters_) qsx_lip_filter_destroy(f);
}
QueryContext::lip_filter_id QueryContext::addLIPFilter(qsx_lip_kind_t kind, std::int64_t cardinality, std::int64_t min_value,
                                                       bool is_anti) {
  qsx_lip_filter_t *f = nullptr;
  CheckStatus(qsx_lip_filter_create(kind, cardinality, min_value, is_anti ? 1 : 0, &f), "qsx_lip_filter_create");
  lip_filters_.push_back(f);
  return static_cast<lip_filter_id>(lip_filters_.size() - 1);
}
void QueryContext::destroyLIPFilter(lip_filter_id id) {
  qsx_lip_filter_destroy(lip_filters_.at(id));
  lip_filters_.at(id) = nullptr;
}
QueryContext::lip_deployment_id QueryContext::addLIPDeployment(LIPFilterDeployment deployment) {
  lip_deployments_.push_back(std::move(deployment));
  return static_cast<lip_deployment_id>(lip_deployments_.size() - 1);
}

// ---------------------------------------------------------------------------
// LIP filter builder / prober
// ---------------------------------------------------------------------------
LIPFilterBuilder::LIPFilterBuilder(const QueryContext::LIPFilterDeployment &deployment, const QueryContext &query_context) {
  for (const auto &e : deployment.build_entries) entries_.emplace_back(query_context.getLIPFilterMutable(e.lip_filter), e.attribute);
}
void LIPFilterBuilder::insertValueAccessor(const StorageBlock &block, const std::uint64_t *filter) const {
  for (const auto &e : entries_) {
    // a NULL is never inserted (SingleIdentityHashFilter.hpp:115-126, BitVectorExactFilter.hpp:115-128)
    std::unique_ptr<DeviceBuffer> not_null = NotNullFilter(block, {e.second}, filter);
    CheckStatus(qsx_lip_build(e.first, block.getRelation().getAttributeType(e.second).id, block.stripe(e.second),
                              block.numTuples(), not_null != nullptr ? static_cast<const std::uint64_t *>(not_null->ptr) : filter,
                              CurrentStream()), "qsx_lip_build");
    if (not_null != nullptr) CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  }
}
bool LIPFilterBuilder::insertBlocks(const std::vector<BlockReference> &blocks) const {
  for (const auto &e : entries_) {
    for (const BlockReference &b : blocks) {
      if (b->nullBitmap(e.second) != nullptr) return false;   // (a compressed attribute is read through stripe(): decoded once)
    }
  }
  std::vector<std::int64_t> rows;
  for (const BlockReference &b : blocks) rows.push_back(b->numTuples());
  std::vector<const void *> keys(blocks.size());
  for (const auto &e : entries_) {
    for (std::size_t b = 0; b < blocks.size(); ++b) keys[b] = blocks[b]->stripe(e.second);
    CheckStatus(qsx_lip_build_blocks(e.first, blocks.front()->getRelation().getAttributeType(e.second).id,
                                     static_cast<std::int64_t>(blocks.size()), rows.data(), keys.data(), nullptr, CurrentStream()),
                "qsx_lip_build_blocks");
  }
  return true;
}
LIPFilterAdaptiveProber::LIPFilterAdaptiveProber(const QueryContext::LIPFilterDeployment &deployment,
                                                 const QueryContext &query_context) {
  for (const auto &e : deployment.probe_entries) entries_.emplace_back(query_context.getLIPFilterMutable(e.lip_filter), e.attribute);
}
void *LIPFilterAdaptiveProber::filterValueAccessor(const StorageBlock &block, const std::uint64_t *filter,
                                                   std::int64_t *num_hits) const {
  const std::int64_t n = block.numTuples();
  const std::size_t bytes = static_cast<std::size_t>((n + 63) / 64) * 8 + 8;
  void *current = nullptr, *next = nullptr;
  CheckStatus(qsx_device_alloc(bytes, &current), "qsx_device_alloc(bitmap)");
  CheckStatus(qsx_device_alloc(bytes, &next), "qsx_device_alloc(bitmap)");
  DeviceBuffer count(8);
  const std::uint64_t *in = filter;
  for (const auto &e : entries_) {
    // a NULL never passes a filter (SingleIdentityHashFilter.hpp:133-152)
    std::unique_ptr<DeviceBuffer> not_null = NotNullFilter(block, {e.second}, in);
    if (not_null != nullptr) in = static_cast<const std::uint64_t *>(not_null->ptr);
    CheckStatus(qsx_lip_probe(e.first, block.getRelation().getAttributeType(e.second).id, block.stripe(e.second), n, in,
                              static_cast<std::uint64_t *>(next), static_cast<std::int64_t *>(count.ptr), CurrentStream()),
                "qsx_lip_probe");
    if (not_null != nullptr) CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
    std::swap(current, next);
    in = static_cast<const std::uint64_t *>(current);
  }
  if (num_hits != nullptr) *num_hits = ReadCount(count.ptr);
  qsx_device_free(next);
  return current;
}
bool LIPFilterAdaptiveProber::filterBlocks(const std::vector<BlockReference> &blocks, void **storage,
                                           std::vector<const std::uint64_t *> *bitmaps, std::int64_t *num_hits) const {
  *storage = nullptr;
  for (const auto &e : entries_) {
    for (const BlockReference &b : blocks) {
      if (b->nullBitmap(e.second) != nullptr) return false;   // (a compressed attribute is read through stripe(): decoded once)
    }
  }
  const std::size_t nb = blocks.size();
  std::vector<std::int64_t> rows;
  std::size_t words = 0;
  for (const BlockReference &b : blocks) {
    rows.push_back(b->numTuples());
    words += static_cast<std::size_t>((b->numTuples() + 63) / 64) + 1;
  }
  // two sets of per-block bitmaps in one allocation; the filters ping-pong between them and the result ends in the first
  CheckStatus(qsx_device_alloc(2 * words * 8 + 8, storage), "qsx_device_alloc(bitmaps)");
  std::vector<std::uint64_t *> cur(nb), nxt(nb);
  std::size_t at = 0;
  for (std::size_t b = 0; b < nb; ++b) {
    cur[b] = static_cast<std::uint64_t *>(*storage) + at;
    nxt[b] = static_cast<std::uint64_t *>(*storage) + words + at;
    at += static_cast<std::size_t>((rows[b] + 63) / 64) + 1;
  }
  std::vector<const void *> keys(nb);
  DeviceBuffer count(8);
  bool first = true;
  for (const auto &e : entries_) {
    for (std::size_t b = 0; b < nb; ++b) keys[b] = blocks[b]->stripe(e.second);
    CheckStatus(qsx_lip_probe_blocks(e.first, blocks.front()->getRelation().getAttributeType(e.second).id, static_cast<std::int64_t>(nb),
                                     rows.data(), keys.data(), first ? nullptr : reinterpret_cast<const std::uint64_t *const *>(cur.data()),
                                     nxt.data(), static_cast<std::int64_t *>(count.ptr), CurrentStream()), "qsx_lip_probe_blocks");
    std::swap(cur, nxt);
    first = false;
  }
  if (num_hits != nullptr) {
    *num_hits = 0;
    if (first) {
      for (std::int64_t r : rows) *num_hits += r;
    } else {
      *num_hits = ReadCount(count.ptr);
    }
  }
  if (first) {   // no filter attached: every tuple
    for (std::size_t b = 0; b < nb; ++b) {
      CheckStatus(qsx_memset_device(nxt[b], 0xFF, static_cast<std::size_t>((rows[b] + 63) / 64) * 8, CurrentStream()), "qsx_memset_device");
      if (rows[b] > 0) {
        CheckStatus(qsx_bitmap_combine(0, nxt[b], nxt[b], rows[b], cur[b], CurrentStream()), "qsx_bitmap_combine");   // clears the tail bits
      }
    }
  }
  bitmaps->assign(cur.begin(), cur.end());
  return true;
}
LIPFilterBuilder *CreateLIPFilterBuilderHelper(QueryContext::lip_deployment_id id, const QueryContext *query_context) {
  const QueryContext::LIPFilterDeployment *d = query_context->getLIPDeployment(id);
  return d == nullptr || d->build_entries.empty() ? nullptr : new LIPFilterBuilder(*d, *query_context);
}
LIPFilterAdaptiveProber *CreateLIPFilterAdaptiveProberHelper(QueryContext::lip_deployment_id id, const QueryContext *query_context) {
  const QueryContext::LIPFilterDeployment *d = query_context->getLIPDeployment(id);
  return d == nullptr || d->probe_entries.empty() ? nullptr : new LIPFilterAdaptiveProber(*d, *query_context);
}
QueryContext::predicate_id QueryContext::addPredicate(Predicate p) {
  predicates_.push_back(std::move(p));
  return static_cast<predicate_id>(predicates_.size() - 1);
}
QueryContext::scalar_group_id QueryContext::addScalarGroup(std::vector<attribute_id> attrs) {
  scalar_groups_.push_back(std::move(attrs));
  return static_cast<scalar_group_id>(scalar_groups_.size() - 1);
}
QueryContext::join_hash_table_id QueryContext::addJoinHashTable(TypeID key_type, std::int64_t estimated_entries,
                                                                std::size_t num_partitions,
                                                                const ExactKeyRange *exact_key_range) {
  std::vector<qsx_join_table_t *> parts(num_partitions, nullptr);
  if (key_type == kChar) key_type = kLong;   // a CHAR(n <= 8) key travels as the LONG qsx_join_key_pack_char makes of it
  for (std::size_t p = 0; p < num_partitions; ++p) {
    if (exact_key_range != nullptr) {
      // every partition addresses the whole range: the single-node partition function is not a stride of the key
      CheckStatus(qsx_join_table_create_dense(key_type, exact_key_range->min_value, exact_key_range->max_value, 1,
                                              estimated_entries, &parts[p]),
                  "qsx_join_table_create_dense");
    } else {
      CheckStatus(qsx_join_table_create(key_type, estimated_entries, &parts[p]), "qsx_join_table_create");
    }
  }
  join_tables_.push_back(std::move(parts));
  return static_cast<join_hash_table_id>(join_tables_.size() - 1);
}
void QueryContext::destroyJoinHashTable(join_hash_table_id id, partition_id part) {
  qsx_join_table_destroy(join_tables_.at(id).at(part));
  join_tables_.at(id).at(part) = nullptr;
}
QueryContext::aggregation_state_id QueryContext::addAggregationState(const AggregationStateSpec &spec,
                                                                     std::size_t num_partitions) {
  std::vector<std::unique_ptr<AggregationOperationState>> parts;
  for (std::size_t p = 0; p < num_partitions; ++p) parts.emplace_back(new AggregationOperationState(spec));
  agg_states_.push_back(std::move(parts));
  return static_cast<aggregation_state_id>(agg_states_.size() - 1);
}
QueryContext::insert_destination_id QueryContext::addInsertDestination(CatalogRelation *relation,
                                                                       StorageManager *storage_manager) {
  destinations_.emplace_back(new InsertDestination(relation, storage_manager));
  return static_cast<insert_destination_id>(destinations_.size() - 1);
}
QueryContext::insert_destination_id QueryContext::addPartitionAwareInsertDestination(CatalogRelation *relation,
                                                                                     StorageManager *storage_manager) {
  if (!relation->hasPartitionScheme()) {
    throw ExecutionError("addPartitionAwareInsertDestination: the output relation has no partition scheme", QSX_ERR_INVALID_ARGUMENT);
  }
  destinations_.emplace_back(new InsertDestination(relation, storage_manager, relation->getNumPartitions(), relation->getPartitionAttribute()));
  return static_cast<insert_destination_id>(destinations_.size() - 1);
}

// has_repartition of an operator (RelationalOperator.hpp:311-320) and the kind of its InsertDestination must agree: a
// repartitioning operator whose destination would drop the partition scheme is a plan error, never silently accepted.
void CheckRepartition(const char *op, bool has_repartition, const InsertDestination *dest) {
  if (dest == nullptr || has_repartition == dest->isPartitionAware()) return;
  throw ExecutionError(std::string(op) + (has_repartition ? ": has_repartition needs a PartitionAwareInsertDestination (QueryContext::addPartitionAwareInsertDestination)"
                                                          : ": a PartitionAwareInsertDestination needs has_repartition = true"),
                       QSX_ERR_INVALID_ARGUMENT);
}

// ---------------------------------------------------------------------------
// WorkOrdersContainer
// ---------------------------------------------------------------------------
void WorkOrdersContainer::addNormalWorkOrder(WorkOrder *workorder, std::size_t operator_index) {
  std::lock_guard<std::mutex> lock(mutex_);
  queues_.at(operator_index).emplace_back(workorder);
}
bool WorkOrdersContainer::hasNormalWorkOrder(std::size_t operator_index) const {
  std::lock_guard<std::mutex> lock(mutex_);
  return !queues_.at(operator_index).empty();
}
WorkOrder *WorkOrdersContainer::getNormalWorkOrder(std::size_t operator_index) {
  std::lock_guard<std::mutex> lock(mutex_);
  auto &q = queues_.at(operator_index);
  if (q.empty()) return nullptr;
  WorkOrder *wo = q.front().release();
  q.pop_front();
  return wo;
}
std::size_t WorkOrdersContainer::getNumNormalWorkOrders(std::size_t operator_index) const {
  std::lock_guard<std::mutex> lock(mutex_);
  return queues_.at(operator_index).size();
}

// ---------------------------------------------------------------------------
// Select
// ---------------------------------------------------------------------------
SelectOperator::SelectOperator(std::size_t query_id, const CatalogRelation &input_relation, bool has_repartition,
                               const CatalogRelation &output_relation,
                               QueryContext::insert_destination_id output_destination_index,
                               QueryContext::predicate_id predicate_index, std::vector<attribute_id> &&selection,
                               bool input_relation_is_stored, bool on_gpu)
    : RelationalOperator(query_id, 1, has_repartition), input_relation_(input_relation),
      output_relation_(output_relation), output_destination_index_(output_destination_index),
      predicate_index_(predicate_index), simple_selection_(std::move(selection)),
      input_relation_is_stored_(input_relation_is_stored), on_gpu_(on_gpu) {
  if (input_relation_is_stored) input_relation_block_ids_ = input_relation.getBlocksSnapshot();
}

SelectOperator::SelectOperator(std::size_t query_id, const CatalogRelation &input_relation, bool has_repartition,
                               const CatalogRelation &output_relation,
                               QueryContext::insert_destination_id output_destination_index,
                               QueryContext::predicate_id predicate_index, std::vector<ScalarPtr> &&selection,
                               bool input_relation_is_stored)
    : RelationalOperator(query_id, 1, has_repartition), input_relation_(input_relation),
      output_relation_(output_relation), output_destination_index_(output_destination_index),
      predicate_index_(predicate_index), selection_(std::move(selection)),
      input_relation_is_stored_(input_relation_is_stored), on_gpu_(true) {
  if (selection_.size() != output_relation.size()) {
    throw ExecutionError("SelectOperator: one Scalar per output attribute", QSX_ERR_INVALID_ARGUMENT);
  }
  for (std::size_t i = 0; i < selection_.size(); ++i) {
    if (selection_[i] == nullptr) throw ExecutionError("SelectOperator: null Scalar", QSX_ERR_INVALID_ARGUMENT);
    // (an expression's output attribute must have its result type: checked per work order, against the block's relation)
  }
  if (input_relation_is_stored) input_relation_block_ids_ = input_relation.getBlocksSnapshot();
}

bool SelectOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                      StorageManager *storage_manager, const tmb::client_id, tmb::MessageBus *) {
  // SelectOperator.cpp:47-107: one work order per input block; streaming inputs
  // generate incrementally and finish when done_feeding_input_relation_.
  const Predicate *predicate = query_context->getPredicate(predicate_index_);
  InsertDestination *dest = query_context->getInsertDestination(output_destination_index_);
  CheckRepartition("SelectOperator", has_repartition_, dest);
  std::lock_guard<std::mutex> lock(mutex_);
  while (num_workorders_generated_ < input_relation_block_ids_.size()) {
    // every block that has arrived, in runs of blocks_per_work_order_ (1: the reference's one work order per block)
    const std::size_t take = on_gpu_ ? std::min(blocks_per_work_order_, input_relation_block_ids_.size() - num_workorders_generated_) : 1;
    if (take > 1) {
      std::vector<block_id> run(input_relation_block_ids_.begin() + static_cast<std::ptrdiff_t>(num_workorders_generated_),
                                input_relation_block_ids_.begin() + static_cast<std::ptrdiff_t>(num_workorders_generated_ + take));
      container->addNormalWorkOrder(new SelectWorkOrder(query_id_, std::move(run), predicate, simple_selection_, dest, storage_manager,
                                                        CreateLIPFilterAdaptiveProberHelper(lip_deployment_index_, query_context),
                                                        selection_.empty() ? nullptr : &selection_),
                                    op_index_);
    } else {
      container->addNormalWorkOrder(new SelectWorkOrder(query_id_, input_relation_block_ids_[num_workorders_generated_],
                                                        predicate, simple_selection_, dest, storage_manager, on_gpu_,
                                                        CreateLIPFilterAdaptiveProberHelper(lip_deployment_index_, query_context),
                                                        selection_.empty() ? nullptr : &selection_),
                                    op_index_);
    }
    num_workorders_generated_ += take;
  }
  return input_relation_is_stored_ || done_feeding_input_relation_;
}

void SelectWorkOrder::execute() {
  if (!on_gpu_) {
    executeOnHost();
    return;
  }
  if (run_block_ids_.empty()) {
    executeBlock(input_block_id_);
    return;
  }
  if (executeRun()) return;
  for (block_id id : run_block_ids_) executeBlock(id);
}

// A run of blocks as one unit: every predicate term is one launch over all blocks (per-block bitmaps chained through the
// terms like a conjunction), then the selected tuples of the run, block after block, are compacted into ONE output block
// — what consecutive SelectWorkOrders do to an InsertDestination's current block (InsertDestination.cpp:222-260).
bool SelectWorkOrder::executeRun() {
  const bool has_terms = predicate_ != nullptr && !predicate_->conjuncts.empty();
  if (!has_terms && lip_filter_adaptive_prober_ == nullptr) return false;   // a plain copy: block by block
  static const Predicate no_terms;
  const Predicate &predicate = has_terms ? *predicate_ : no_terms;
  std::vector<attribute_id> selection;
  if (selection_ != nullptr && !selection_->empty()) {
    for (const ScalarPtr &scalar : *selection_) {
      if (scalar->kind != Scalar::kAttribute) return false;
      selection.push_back(scalar->attribute);
    }
  } else {
    selection = simple_selection_;
  }
  if (selection.size() > QSX_MAX_COLUMNS) return false;
  std::vector<BlockReference> blocks;
  std::vector<std::int64_t> rows;
  std::int64_t total_rows = 0;
  std::size_t bitmap_words = 0;
  const StorageBlock *reference_block = nullptr;   // the first non-empty block: what the others have to agree with
  for (block_id id : run_block_ids_) {
    blocks.push_back(storage_manager_->getBlock(id));
    const StorageBlock &b = *blocks.back();
    for (const ComparisonPredicate &term : predicate.conjuncts) {
      const Type &t = b.getRelation().getAttributeType(term.attribute);
      if (term.rhs_attribute != kInvalidAttributeID || b.nullBitmap(term.attribute) != nullptr) return false;
      if (b.numTuples() == 0) continue;                 // (an empty block has neither codes nor an order to agree on)
      // a term on the blocks' sort column is a per-block binary search (also on the code stripe of a compressed sort column), a
      // term on a compressed attribute a scan of the code stripes with the comparison rewritten per block
      if (t.id == kChar && b.compressedAttribute(term.attribute) == nullptr && term.attribute == b.sortColumn()) return false;
      if (reference_block == nullptr) reference_block = &b;
      const StorageBlock &f = *reference_block;
      if ((term.attribute == b.sortColumn()) != (term.attribute == f.sortColumn())) return false;
      const CompressedAttribute *cb = b.compressedAttribute(term.attribute), *cf = f.compressedAttribute(term.attribute);
      if ((cb != nullptr) != (cf != nullptr) || (cb != nullptr && cb->code_width != cf->code_width)) return false;
    }
    for (attribute_id a : selection) {
      if (b.nullBitmap(a) != nullptr) return false;   // (projected values of a compressed attribute: stripe() decodes once)
    }
    rows.push_back(b.numTuples());
    total_rows += b.numTuples();
    bitmap_words += static_cast<std::size_t>((b.numTuples() + 63) / 64) + 1;
  }
  const std::size_t nb = blocks.size();
  // two sets of per-block bitmaps in two allocations, the terms ping-pong between them
  DeviceBuffer set_a(bitmap_words * 8 + 8), set_b(bitmap_words * 8 + 8), counts(nb * 8 + 8);
  std::vector<std::uint64_t *> cur(nb), nxt(nb);
  std::size_t at = 0;
  for (std::size_t b = 0; b < nb; ++b) {
    cur[b] = static_cast<std::uint64_t *>(set_a.ptr) + at;
    nxt[b] = static_cast<std::uint64_t *>(set_b.ptr) + at;
    at += static_cast<std::size_t>((rows[b] + 63) / 64) + 1;
  }
  std::vector<const void *> stripes(nb);
  // SelectOperator.cpp:161-195: predicate matches, then the LIP filters on what is left — here the filters run first and
  // the predicate only looks at their survivors (the same conjunction, as in the single-block form)
  struct OwnedStorage {
    void *ptr = nullptr;
    ~OwnedStorage() { qsx_device_free(ptr); }
  } lip_storage;
  std::vector<const std::uint64_t *> lip_bitmaps;
  std::int64_t lip_hits = 0;
  if (lip_filter_adaptive_prober_ != nullptr &&
      !lip_filter_adaptive_prober_->filterBlocks(blocks, &lip_storage.ptr, &lip_bitmaps, has_terms ? nullptr : &lip_hits)) {
    return false;
  }
  bool first = true;
  for (const ComparisonPredicate &term : predicate.conjuncts) {
    const Type &t = blocks.front()->getRelation().getAttributeType(term.attribute);
    const StorageBlock &ref = reference_block != nullptr ? *reference_block : *blocks.front();
    if (ref.compressedAttribute(term.attribute) == nullptr) {   // (a compressed sort column is searched on its codes)
      for (std::size_t b = 0; b < nb; ++b) stripes[b] = blocks[b]->stripe(term.attribute);
    }
    const std::uint64_t *const *in = first ? (lip_bitmaps.empty() ? nullptr : lip_bitmaps.data())
                                           : reinterpret_cast<const std::uint64_t *const *>(cur.data());
    const bool on_sort_column = term.attribute == ref.sortColumn();
    if (on_sort_column && ref.compressedAttribute(term.attribute) != nullptr) {
      // the sort column of compressed blocks: the comparison rewritten on every block's own codes
      // (CompressedTupleStorageSubBlock::getMatchesForPredicate), then one search per block on the code stripes
      std::vector<std::int32_t> ops(nb);
      std::vector<std::uint32_t> firsts(nb), seconds(nb);
      for (std::size_t b = 0; b < nb; ++b) {
        const CompressedAttribute *c = blocks[b]->compressedAttribute(term.attribute);
        if (c == nullptr) {   // an empty block
          stripes[b] = nullptr;
          ops[b] = QSX_CODE_LT;
          firsts[b] = seconds[b] = 0;
          continue;
        }
        const PredicateTransformResult r = TransformPredicateOnCompressedAttribute(*c, t.id, term.comparison, term.literal);
        stripes[b] = c->codes;
        if (r.type == PredicateTransformResult::kAll || r.type == PredicateTransformResult::kNone) {
          ops[b] = r.type == PredicateTransformResult::kAll ? QSX_CODE_GE : QSX_CODE_LT;   // every code / no code
          firsts[b] = seconds[b] = 0;
        } else {
          ops[b] = r.comp;
          firsts[b] = r.first_literal;
          seconds[b] = r.second_literal;
        }
      }
      CheckStatus(qsx_select_codes_sorted_blocks(ref.compressedAttribute(term.attribute)->code_width, static_cast<std::int64_t>(nb),
                                                 rows.data(), stripes.data(), ops.data(), firsts.data(), seconds.data(), in, nxt.data(),
                                                 static_cast<std::int64_t *>(counts.ptr), CurrentStream()), "qsx_select_codes_sorted_blocks");
    } else if (ref.compressedAttribute(term.attribute) != nullptr) {
      // a compressed attribute: every block's code stripe scanned with the comparison rewritten on that block's codes
      std::vector<std::int32_t> ops(nb);
      std::vector<std::uint32_t> firsts(nb), seconds(nb);
      for (std::size_t b = 0; b < nb; ++b) {
        const CompressedAttribute *c = blocks[b]->compressedAttribute(term.attribute);
        if (c == nullptr) {   // an empty block
          stripes[b] = nullptr;
          ops[b] = QSX_CODE_LT;
          firsts[b] = seconds[b] = 0;
          continue;
        }
        const PredicateTransformResult r = TransformPredicateOnCompressedAttribute(*c, t.id, term.comparison, term.literal);
        stripes[b] = c->codes;
        if (r.type == PredicateTransformResult::kAll || r.type == PredicateTransformResult::kNone) {
          ops[b] = r.type == PredicateTransformResult::kAll ? QSX_CODE_GE : QSX_CODE_LT;
          firsts[b] = seconds[b] = 0;
        } else {
          ops[b] = r.comp;
          firsts[b] = r.first_literal;
          seconds[b] = r.second_literal;
        }
      }
      CheckStatus(qsx_select_codes_blocks(ref.compressedAttribute(term.attribute)->code_width, static_cast<std::int64_t>(nb), rows.data(),
                                          stripes.data(), ops.data(), firsts.data(), seconds.data(), in, nxt.data(),
                                          static_cast<std::int64_t *>(counts.ptr), CurrentStream()), "qsx_select_codes_blocks");
    } else if (t.id == kChar) {
      // CHAR(n) OP string literal on plain stripes (AsciiStringUncheckedComparator, AsciiStringComparators.hpp:218-251)
      CheckStatus(qsx_select_cmp_char_blocks(t.width, static_cast<std::int64_t>(nb), rows.data(), stripes.data(), static_cast<int>(term.comparison),
                                             term.literal.text.data(), static_cast<int>(term.literal.text.size()), in, nxt.data(),
                                             static_cast<std::int64_t *>(counts.ptr), CurrentStream()), "qsx_select_cmp_char_blocks");
    } else if (on_sort_column) {
      // SortColumnPredicateEvaluator (storage/ColumnStoreUtil.cpp:40-280), one search per block
      CheckStatus(qsx_select_cmp_sorted_blocks(t.id, static_cast<std::int64_t>(nb), rows.data(), stripes.data(), static_cast<int>(term.comparison),
                                               &term.literal.v, in, nxt.data(), static_cast<std::int64_t *>(counts.ptr), CurrentStream()),
                  "qsx_select_cmp_sorted_blocks");
    } else {
      CheckStatus(qsx_select_cmp_blocks(t.id, static_cast<std::int64_t>(nb), rows.data(), stripes.data(), static_cast<int>(term.comparison),
                                        &term.literal.v, in, nxt.data(), static_cast<std::int64_t *>(counts.ptr), CurrentStream()),
                  "qsx_select_cmp_blocks");
    }
    std::swap(cur, nxt);
    first = false;
  }
  std::int64_t matches = lip_hits;
  const std::uint64_t *const *selected = lip_bitmaps.empty() ? nullptr : lip_bitmaps.data();   // only LIP filters: their bitmaps
  if (has_terms) {
    std::vector<std::int64_t> block_matches(nb);
    CheckStatus(qsx_copy_to_host(block_matches.data(), counts.ptr, nb * 8, CurrentStream()), "qsx_copy_to_host");
    CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
    matches = 0;
    for (std::int64_t m : block_matches) matches += m;
    selected = reinterpret_cast<const std::uint64_t *const *>(cur.data());
  }
  block_id out_id;
  BlockReference out = output_destination_->getBlockForInsertion(matches > 0 ? matches : 1, &out_id);
  std::vector<const void *> src(nb * selection.size());
  std::vector<void *> dst;
  std::vector<std::int32_t> widths;
  for (std::size_t i = 0; i < selection.size(); ++i) {
    dst.push_back(out->stripe(static_cast<attribute_id>(i)));
    widths.push_back(blocks.front()->getRelation().getAttributeType(selection[i]).width);
    for (std::size_t b = 0; b < nb; ++b) src[b * selection.size() + i] = blocks[b]->stripe(selection[i]);
  }
  const std::size_t ws_bytes = qsx_compact_blocks_workspace_bytes(static_cast<std::int64_t>(nb), rows.data());
  DeviceBuffer ws(ws_bytes + 8), count(8);
  CheckStatus(qsx_compact_gather_blocks(static_cast<int>(selection.size()), widths.data(), static_cast<std::int64_t>(nb), rows.data(),
                                        src.data(), selected, nullptr, dst.data(),
                                        nullptr, static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()),
              "qsx_compact_gather_blocks");
  const std::int64_t written = ReadCount(count.ptr);   // synchronises the work order, like the reference's execute()
  output_destination_->returnBlock(out_id, written, getPartitionId());
  return true;
}

void SelectWorkOrder::executeBlock(block_id input_block_id) {
  BlockReference block = storage_manager_->getBlock(input_block_id);
  const std::int64_t n = block->numTuples();
  std::int64_t matches = 0;
  Predicate all;
  // SelectOperator.cpp:161-195: predicate matches, then the LIP filters on what is left; the filters run
  // first here and the predicate only evaluates their survivors (same conjunction)
  void *lip = nullptr;
  if (lip_filter_adaptive_prober_ != nullptr) lip = lip_filter_adaptive_prober_->filterValueAccessor(*block, nullptr, nullptr);
  void *bitmap = (predicate_ != nullptr ? predicate_ : &all)
                     ->getMatchesForBlock(*block, &matches, static_cast<const std::uint64_t *>(lip));  // getMatchesForPredicate
  qsx_device_free(lip);
  block_id out_id;
  BlockReference out = output_destination_->getBlockForInsertion(matches > 0 ? matches : 1, &out_id);
  // block->selectSimple(simple_selection_, matches, output_destination_) (StorageBlock.cpp:390-399), or block->select(
  // selection_, ...) (:363-388): every Scalar's values for the block, then the matching rows of each
  std::vector<const void *> src;
  std::vector<void *> dst;
  std::vector<std::int32_t> widths;
  std::vector<std::unique_ptr<DeviceBuffer>> expression_values;
  std::vector<attribute_id> null_sources;   // per output attribute: the input attribute whose null bitmap it inherits
  if (selection_ != nullptr && !selection_->empty()) {
    for (std::size_t i = 0; i < selection_->size(); ++i) {
      const ScalarPtr &scalar = (*selection_)[i];
      dst.push_back(out->stripe(static_cast<attribute_id>(i)));
      if (scalar->kind == Scalar::kAttribute) {
        src.push_back(block->stripe(scalar->attribute));
        widths.push_back(block->getRelation().getAttributeType(scalar->attribute).width);
        null_sources.push_back(scalar->attribute);
        continue;
      }
      // ScalarBinaryExpression / ScalarLiteral: one fused pass over the operand stripes (qsx_eval_expression)
      std::vector<attribute_id> attrs;
      ExpressionFlattener flattener([&](attribute_id a) {
        for (std::size_t c = 0; c < attrs.size(); ++c) if (attrs[c] == a) return static_cast<int>(c);
        attrs.push_back(a);
        return static_cast<int>(attrs.size() - 1);
      });
      const qsx_operand_t result = flattener.add(scalar);
      const void *cols[QSX_MAX_COLUMNS];
      std::int32_t types[QSX_MAX_COLUMNS];
      if (attrs.size() > QSX_MAX_COLUMNS) throw ExecutionError("SelectWorkOrder: expression over too many attributes", QSX_ERR_UNSUPPORTED);
      for (std::size_t c = 0; c < attrs.size(); ++c) {
        cols[c] = block->stripe(attrs[c]);
        types[c] = block->getRelation().getAttributeType(attrs[c]).id;
      }
      double consts[QSX_MAX_CONSTS] = {};
      for (std::size_t c = 0; c < flattener.consts().size(); ++c) consts[c] = flattener.consts()[c];
      expression_values.emplace_back(new DeviceBuffer(static_cast<std::size_t>(n > 0 ? n : 1) * 8));
      const TypeID value_type = ScalarResultType(scalar, block->getRelation());
      const int value_width = value_type == kInt ? 4 : 8;
      if (out->getRelation().getAttributeType(static_cast<attribute_id>(i)).width != value_width) {
        throw ExecutionError("SelectWorkOrder: the output attribute of an expression must have the expression's type "
                             "(INT op INT is an INT, with a LONG a LONG, with a FLOAT / DOUBLE a DOUBLE)", QSX_ERR_INVALID_ARGUMENT);
      }
      if (value_type == kDouble) {
        CheckStatus(qsx_eval_expression(static_cast<int>(attrs.size()), cols, types, static_cast<int>(flattener.instrs().size()),
                                        flattener.instrs().data(), consts, result, n, static_cast<double *>(expression_values.back()->ptr),
                                        CurrentStream()), "qsx_eval_expression");
      } else {
        // integer operands: integer arithmetic (ArithmeticBinaryOperators.hpp:203-340 for INT / LONG)
        std::int64_t int_consts[QSX_MAX_CONSTS] = {};
        for (std::size_t c = 0; c < flattener.consts().size(); ++c) int_consts[c] = static_cast<std::int64_t>(flattener.consts()[c]);
        CheckStatus(qsx_eval_expression_long(static_cast<int>(attrs.size()), cols, types, static_cast<int>(flattener.instrs().size()),
                                             flattener.instrs().data(), int_consts, result, n, value_width, expression_values.back()->ptr,
                                             CurrentStream()), "qsx_eval_expression_long");
      }
      src.push_back(expression_values.back()->ptr);
      widths.push_back(value_width);
      null_sources.push_back(kInvalidAttributeID);
    }
  } else {
    for (std::size_t i = 0; i < simple_selection_.size(); ++i) {
      src.push_back(block->stripe(simple_selection_[i]));
      dst.push_back(out->stripe(static_cast<attribute_id>(i)));
      widths.push_back(block->getRelation().getAttributeType(simple_selection_[i]).width);
      null_sources.push_back(simple_selection_[i]);
    }
  }
  const std::size_t ws_bytes = qsx_compact_workspace_bytes(n);
  DeviceBuffer ws(ws_bytes), count(8);
  CheckStatus(qsx_compact_gather(static_cast<int>(src.size()), src.data(), widths.data(),
                                 static_cast<const std::uint64_t *>(bitmap), n, dst.data(),
                                 static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()),
              "qsx_compact_gather");
  const std::int64_t written = ReadCount(count.ptr);  // synchronises the work order, like the reference's execute()
  ProjectNullBitmaps(*block, null_sources, bitmap, written, out.get());
  qsx_device_free(bitmap);
  output_destination_->returnBlock(out_id, written, getPartitionId());
}

// CPU work order of BASELINE config 1: the same plumbing with the loops on the
// host (blocks in host memory).  Not a fallback: only built by operators that
// were constructed with on_gpu = false.
void SelectWorkOrder::executeOnHost() {
  BlockReference block = storage_manager_->getBlock(input_block_id_);
  const std::int64_t n = block->numTuples();
  std::vector<bool> match(static_cast<std::size_t>(n), true);
  if (predicate_ != nullptr) {
    for (const ComparisonPredicate &term : predicate_->conjuncts) {
      const Type &t = block->getRelation().getAttributeType(term.attribute);
      const char *base = static_cast<const char *>(block->stripe(term.attribute));
      for (std::int64_t i = 0; i < n; ++i) {
        if (!match[i]) continue;
        double a, b;
        std::int64_t ia = 0, ib = 0;
        bool is_int = true;
        switch (t.id) {
          case kInt: ia = reinterpret_cast<const std::int32_t *>(base)[i]; ib = term.literal.v.i32; break;
          case kLong: ia = reinterpret_cast<const std::int64_t *>(base)[i]; ib = term.literal.v.i64; break;
          case kFloat: is_int = false; a = reinterpret_cast<const float *>(base)[i]; b = term.literal.v.f32; break;
          default: is_int = false; a = reinterpret_cast<const double *>(base)[i]; b = term.literal.v.f64; break;
        }
        bool r;
        switch (term.comparison) {
          case ComparisonID::kEqual: r = is_int ? ia == ib : a == b; break;
          case ComparisonID::kNotEqual: r = is_int ? ia != ib : a != b; break;
          case ComparisonID::kLess: r = is_int ? ia < ib : a < b; break;
          case ComparisonID::kLessOrEqual: r = is_int ? ia <= ib : a <= b; break;
          case ComparisonID::kGreater: r = is_int ? ia > ib : a > b; break;
          default: r = is_int ? ia >= ib : a >= b; break;
        }
        match[i] = r;
      }
    }
  }
  std::int64_t matches = 0;
  for (std::int64_t i = 0; i < n; ++i) matches += match[i];
  block_id out_id;
  BlockReference out = output_destination_->getBlockForInsertion(matches > 0 ? matches : 1, &out_id);
  for (std::size_t c = 0; c < simple_selection_.size(); ++c) {
    const int w = block->getRelation().getAttributeType(simple_selection_[c]).width;
    const char *src = static_cast<const char *>(block->stripe(simple_selection_[c]));
    char *dst = static_cast<char *>(out->stripe(static_cast<attribute_id>(c)));
    std::int64_t o = 0;
    for (std::int64_t i = 0; i < n; ++i) {
      if (match[i]) std::memcpy(dst + (o++) * w, src + i * w, w);
    }
  }
  output_destination_->returnBlock(out_id, matches, getPartitionId());
}

// ---------------------------------------------------------------------------
// BuildHash
// ---------------------------------------------------------------------------
BuildHashOperator::BuildHashOperator(std::size_t query_id, const CatalogRelation &input_relation,
                                     bool input_relation_is_stored, const std::vector<attribute_id> &join_key_attributes,
                                     bool, std::size_t num_partitions, QueryContext::join_hash_table_id hash_table_index,
                                     QueryContext::predicate_id build_predicate_index)
    : RelationalOperator(query_id, num_partitions), input_relation_(input_relation),
      input_relation_is_stored_(input_relation_is_stored), join_key_attributes_(join_key_attributes),
      is_broadcast_join_(num_partitions > 1u && !input_relation.hasPartitionScheme()),
      hash_table_index_(hash_table_index), build_predicate_index_(build_predicate_index), input_(num_partitions) {
  if (join_key_attributes.empty() || join_key_attributes.size() > QSX_MAX_KEYS) {
    throw ExecutionError("BuildHashOperator: 1 to 4 INT/LONG join key attributes are on the GPU path", QSX_ERR_UNSUPPORTED);
  }
  if (input_relation_is_stored) {
    // BuildHashOperator.hpp:105-119: per-partition blocks, or every block for every partition (broadcast)
    if (input_relation.hasPartitionScheme() && input_relation.getNumPartitions() != num_partitions) {
      throw ExecutionError("BuildHashOperator: num_partitions differs from the input relation's partition scheme", QSX_ERR_INVALID_ARGUMENT);
    }
    for (partition_id part = 0; part < num_partitions; ++part) {
      input_.ids[part] = is_broadcast_join_ ? input_relation.getBlocksSnapshot() : input_relation.getBlocksInPartition(part);
    }
  }
}

bool BuildHashOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                         StorageManager *storage_manager, const tmb::client_id, tmb::MessageBus *) {
  const Predicate *predicate = query_context->getPredicate(build_predicate_index_);
  std::lock_guard<std::mutex> lock(mutex_);
  if (!started_) {
    query_context->setJoinHashTableBuildKeyAttributes(hash_table_index_, join_key_attributes_);
    started_ = true;
  }
  for (partition_id part = 0; part < num_partitions_; ++part) {   // BuildHashOperator.cpp:82-110
    qsx_join_table_t *table = query_context->getJoinHashTable(hash_table_index_, part);
    while (input_.generated[part] < input_.ids[part].size()) {
      const std::size_t take = std::min(blocks_per_work_order_, input_.ids[part].size() - input_.generated[part]);
      BuildHashWorkOrder *order = new BuildHashWorkOrder(query_id_, input_relation_, join_key_attributes_,
                                                         input_.ids[part][input_.generated[part]], predicate, table,
                                                         storage_manager, part,
                                                         CreateLIPFilterBuilderHelper(lip_deployment_index_, query_context));
      if (take > 1) {
        order->setRun(std::vector<block_id>(input_.ids[part].begin() + static_cast<std::ptrdiff_t>(input_.generated[part]),
                                            input_.ids[part].begin() + static_cast<std::ptrdiff_t>(input_.generated[part] + take)));
      }
      container->addNormalWorkOrder(order, op_index_);
      input_.generated[part] += take;
    }
  }
  return input_relation_is_stored_ || done_feeding_input_relation_;
}

namespace {
// The single key column the join table sees for one block: the attribute's stripe, or — for a
// composite key — the LONG fold of the components (qsx_join_key_pack; the QueryContext creates
// the table of a composite key with key type kLong).  `exact` is false when the fold is the
// reference's composite hash and joined pairs still need their components compared.
// A CHAR(n <= 8) key attribute of `block` as a LONG stripe (qsx_join_key_pack_char); wider strings are not join keys here.
std::unique_ptr<DeviceBuffer> CharKeyAsLong(const StorageBlock &block, attribute_id a) {
  const Type &t = block.getRelation().getAttributeType(a);
  if (t.width > 8) throw ExecutionError("join key CHAR(n): n > 8 is not supported", QSX_ERR_UNSUPPORTED);
  std::unique_ptr<DeviceBuffer> out(new DeviceBuffer(static_cast<std::size_t>(block.numTuples()) * 8 + 8));
  CheckStatus(qsx_join_key_pack_char(block.stripe(a), t.width, block.numTuples(), static_cast<std::int64_t *>(out->ptr), CurrentStream()),
              "qsx_join_key_pack_char");
  return out;
}

struct JoinKeys {
  const void *ptr = nullptr;
  bool exact = true;
  std::unique_ptr<DeviceBuffer> packed;
  std::vector<std::unique_ptr<DeviceBuffer>> char_keys;
  JoinKeys(const StorageBlock &block, const std::vector<attribute_id> &attrs) {
    if (attrs.size() == 1) {
      if (block.getRelation().getAttributeType(attrs.front()).id == kChar) {
        char_keys.push_back(CharKeyAsLong(block, attrs.front()));
        ptr = char_keys.back()->ptr;
        return;
      }
      ptr = block.stripe(attrs.front());
      return;
    }
    std::vector<const void *> cols;
    std::vector<std::int32_t> types;
    for (attribute_id a : attrs) {
      if (block.getRelation().getAttributeType(a).id == kChar) {
        char_keys.push_back(CharKeyAsLong(block, a));
        cols.push_back(char_keys.back()->ptr);
        types.push_back(kLong);
        continue;
      }
      cols.push_back(block.stripe(a));
      types.push_back(block.getRelation().getAttributeType(a).id);
    }
    packed.reset(new DeviceBuffer(static_cast<std::size_t>(block.numTuples()) * 8 + 8));
    int is_exact = 0;
    CheckStatus(qsx_join_key_pack(static_cast<int>(cols.size()), cols.data(), types.data(), block.numTuples(),
                                  static_cast<std::int64_t *>(packed->ptr), &is_exact, CurrentStream()),
                "qsx_join_key_pack");
    ptr = packed->ptr;
    exact = is_exact != 0;
  }
};
}  // namespace

namespace {
// The key stripes of a run of blocks as the join table sees them: the attribute's stripes, or — composite key — one stripe of
// packed keys for the whole run (qsx_join_key_pack_blocks), block b's keys at row first_rows[b] of it.
struct RunJoinKeys {
  std::vector<const void *> ptr;          // per block
  bool exact = true;
  std::unique_ptr<DeviceBuffer> packed;
  std::vector<std::unique_ptr<DeviceBuffer>> char_keys;
  RunJoinKeys(const std::vector<BlockReference> &blocks, const std::vector<attribute_id> &attrs, const std::vector<std::int64_t> &rows) {
    const CatalogRelation &relation = blocks.front()->getRelation();
    if (attrs.size() == 1) {
      for (const BlockReference &b : blocks) {
        if (relation.getAttributeType(attrs.front()).id == kChar) {
          char_keys.push_back(CharKeyAsLong(*b, attrs.front()));
          ptr.push_back(char_keys.back()->ptr);
        } else {
          ptr.push_back(b->stripe(attrs.front()));
        }
      }
      return;
    }
    std::vector<const void *> cols;
    std::vector<std::int32_t> types;
    for (attribute_id a : attrs) types.push_back(relation.getAttributeType(a).id == kChar ? static_cast<std::int32_t>(kLong) : relation.getAttributeType(a).id);
    std::int64_t total = 0;
    for (std::size_t b = 0; b < blocks.size(); ++b) {
      for (attribute_id a : attrs) {
        if (relation.getAttributeType(a).id == kChar) {
          char_keys.push_back(CharKeyAsLong(*blocks[b], a));
          cols.push_back(char_keys.back()->ptr);
        } else {
          cols.push_back(blocks[b]->stripe(a));
        }
      }
      total += rows[b];
    }
    packed.reset(new DeviceBuffer(static_cast<std::size_t>(total) * 8 + 8));
    int is_exact = 0;
    CheckStatus(qsx_join_key_pack_blocks(static_cast<int>(attrs.size()), types.data(), static_cast<std::int64_t>(blocks.size()), rows.data(),
                                         cols.data(), static_cast<std::int64_t *>(packed->ptr), &is_exact, CurrentStream()),
                "qsx_join_key_pack_blocks");
    exact = is_exact != 0;
    std::int64_t at = 0;
    for (std::size_t b = 0; b < blocks.size(); ++b) {
      ptr.push_back(static_cast<const std::int64_t *>(packed->ptr) + at);
      at += rows[b];
    }
  }
};
}  // namespace

void BuildHashWorkOrder::execute() {
  if (run_block_ids_.empty()) {
    executeBlock(build_block_id_);
    return;
  }
  if (executeRun()) return;
  for (block_id id : run_block_ids_) executeBlock(id);
}

bool BuildHashWorkOrder::executeRun() {
  if (predicate_ != nullptr) return false;
  std::vector<BlockReference> blocks;
  std::vector<std::int64_t> rows;
  std::vector<std::int32_t> bases;
  for (block_id id : run_block_ids_) {
    blocks.push_back(storage_manager_->getBlock(id));
    const StorageBlock &b = *blocks.back();
    for (attribute_id a : join_key_attributes_) {
      if (b.nullBitmap(a) != nullptr) return false;   // (compressed key: stripe() decodes once)
    }
    if (b.firstRow() + b.numTuples() > INT32_MAX) return false;
    rows.push_back(b.numTuples());
    bases.push_back(static_cast<std::int32_t>(b.firstRow()));   // the stored reference: relation-global row number
  }
  if (lip_filter_builder_ != nullptr && !lip_filter_builder_->insertBlocks(blocks)) return false;   // BuildHashOperator.cpp:187-190
  const RunJoinKeys run_keys(blocks, join_key_attributes_, rows);
  const std::vector<const void *> &keys = run_keys.ptr;
  CheckStatus(qsx_join_build_blocks(hash_table_, static_cast<std::int64_t>(blocks.size()), rows.data(), keys.data(), bases.data(), nullptr,
                                    CurrentStream()), "qsx_join_build_blocks");
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  return true;
}

void BuildHashWorkOrder::executeBlock(block_id build_block_id) {
  BlockReference block = storage_manager_->getBlock(build_block_id);
  void *bitmap = nullptr;
  if (predicate_ != nullptr) {
    std::int64_t matches = 0;
    bitmap = predicate_->getMatchesForBlock(*block, &matches);
  }
  // hash_table_->putValueAccessor[CompositeKey](accessor, key_attr(s), nullable, &TupleReferenceGenerator) (:192-203);
  // the stored reference is the relation-global row number of the tuple.
  if (lip_filter_builder_ != nullptr) {
    lip_filter_builder_->insertValueAccessor(*block, static_cast<const std::uint64_t *>(bitmap));  // :187-190
  }
  JoinKeys keys(*block, join_key_attributes_);
  // check_for_null_keys: tuples with a NULL key component are not inserted (HashTable.hpp:1409-1418, 1505-1518)
  std::unique_ptr<DeviceBuffer> not_null = NotNullFilter(*block, join_key_attributes_, static_cast<const std::uint64_t *>(bitmap));
  const std::uint64_t *build_filter = not_null != nullptr ? static_cast<const std::uint64_t *>(not_null->ptr)
                                                          : static_cast<const std::uint64_t *>(bitmap);
  CheckStatus(qsx_join_build(hash_table_, keys.ptr, block->numTuples(),
                             static_cast<std::int32_t>(block->firstRow()), build_filter,
                             CurrentStream()), "qsx_join_build");
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  qsx_device_free(bitmap);
}

// ---------------------------------------------------------------------------
// HashJoin
// ---------------------------------------------------------------------------
HashJoinOperator::HashJoinOperator(std::size_t query_id, const CatalogRelation &build_relation,
                                   const CatalogRelation &probe_relation, bool probe_relation_is_stored,
                                   const std::vector<attribute_id> &join_key_attributes, bool, std::size_t num_partitions,
                                   bool has_repartition, const CatalogRelation &output_relation,
                                   QueryContext::insert_destination_id output_destination_index,
                                   QueryContext::join_hash_table_id hash_table_index,
                                   QueryContext::predicate_id residual_predicate_index,
                                   QueryContext::scalar_group_id selection_index,
                                   const std::vector<bool> *is_selection_on_build, JoinType join_type)
    : RelationalOperator(query_id, num_partitions, has_repartition), build_relation_(build_relation),
      probe_relation_(probe_relation), probe_relation_is_stored_(probe_relation_is_stored),
      join_key_attributes_(join_key_attributes), output_relation_(output_relation),
      output_destination_index_(output_destination_index), hash_table_index_(hash_table_index),
      residual_predicate_index_(residual_predicate_index), selection_index_(selection_index), join_type_(join_type),
      probe_(num_partitions) {
  if (join_key_attributes.empty() || join_key_attributes.size() > QSX_MAX_KEYS) {
    throw ExecutionError("HashJoinOperator: 1 to 4 INT/LONG join key attributes are on the GPU path", QSX_ERR_UNSUPPORTED);
  }
  if (join_type == JoinType::kLeftOuterJoin && residual_predicate_index != QueryContext::kInvalidPredicateId) {
    // as in the reference: HashOuterJoinWorkOrder takes no residual predicate (HashJoinOperator.hpp:571-640)
    throw ExecutionError("HashJoinOperator: outer joins take no residual predicate", QSX_ERR_UNSUPPORTED);
  }
  if (is_selection_on_build != nullptr) is_selection_on_build_ = *is_selection_on_build;
  if (probe_relation_is_stored) {
    if (num_partitions > 1 && probe_relation.getNumPartitions() != num_partitions) {
      throw ExecutionError("HashJoinOperator: num_partitions differs from the probe relation's partition scheme", QSX_ERR_INVALID_ARGUMENT);
    }
    for (partition_id part = 0; part < num_partitions; ++part) {
      probe_.ids[part] = num_partitions > 1 ? probe_relation.getBlocksInPartition(part) : probe_relation.getBlocksSnapshot();
    }
  }
}

bool HashJoinOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                        StorageManager *storage_manager, const tmb::client_id, tmb::MessageBus *) {
  const std::vector<attribute_id> &selection = query_context->getScalarGroup(selection_index_);
  if (is_selection_on_build_.empty()) is_selection_on_build_.assign(selection.size(), false);
  InsertDestination *dest = query_context->getInsertDestination(output_destination_index_);
  CheckRepartition("HashJoinOperator", has_repartition_, dest);
  std::lock_guard<std::mutex> lock(mutex_);
  if (!started_) {
    // the build operator is a blocking dependency: it has published its key attributes by now
    build_key_attributes_ = query_context->getJoinHashTableBuildKeyAttributes(hash_table_index_);
    if (build_key_attributes_.size() != join_key_attributes_.size()) {
      throw ExecutionError("HashJoinOperator: build and probe sides have different numbers of key attributes", QSX_ERR_INVALID_ARGUMENT);
    }
    started_ = true;
  }
  for (partition_id part = 0; part < num_partitions_; ++part) {   // HashJoinOperator.cpp:220-250
    qsx_join_table_t *table = query_context->getJoinHashTable(hash_table_index_, part);
    while (probe_.generated[part] < probe_.ids[part].size()) {
      // every probe block that has arrived, in runs of blocks_per_work_order_ (1: one work order per block)
      const std::size_t take = std::min(blocks_per_work_order_, probe_.ids[part].size() - probe_.generated[part]);
      HashInnerJoinWorkOrder *order =
          new HashInnerJoinWorkOrder(query_id_, build_relation_, probe_relation_, join_key_attributes_, build_key_attributes_,
                                     probe_.ids[part][probe_.generated[part]],
                                     query_context->getPredicate(residual_predicate_index_), selection, is_selection_on_build_,
                                     join_type_, table, dest, storage_manager, part,
                                     CreateLIPFilterAdaptiveProberHelper(lip_deployment_index_, query_context));
      if (take > 1) {
        order->setRun(std::vector<block_id>(probe_.ids[part].begin() + static_cast<std::ptrdiff_t>(probe_.generated[part]),
                                            probe_.ids[part].begin() + static_cast<std::ptrdiff_t>(probe_.generated[part] + take)));
      }
      container->addNormalWorkOrder(order, op_index_);
      probe_.generated[part] += take;
    }
  }
  return probe_relation_is_stored_ || done_feeding_input_relation_;
}

namespace {
// The build relation as gather segments (one per build block; the reference loops over build
// blocks instead, HashJoinOperator.cpp:494-540).
struct BuildSegments {
  std::vector<BlockReference> refs;
  std::vector<std::int64_t> first_rows;
  BuildSegments(const CatalogRelation &build_relation, StorageManager *storage_manager) {
    for (block_id b : build_relation.getBlocksSnapshot()) refs.push_back(storage_manager->getBlock(b));
    std::sort(refs.begin(), refs.end(),
              [](const BlockReference &a, const BlockReference &b) { return a->firstRow() < b->firstRow(); });
    for (const BlockReference &b : refs) first_rows.push_back(b->firstRow());
  }
  void gather(attribute_id attr, int width, const void *build_tids, std::int64_t n, void *dst) const {
    std::vector<const void *> segs;
    for (const BlockReference &b : refs) segs.push_back(b->stripe(attr));
    CheckStatus(qsx_gather_segmented(width, static_cast<int>(segs.size()), segs.data(), first_rows.data(),
                                     static_cast<const std::int32_t *>(build_tids), n, dst, CurrentStream()),
                "qsx_gather_segmented");
  }
  // null bits of a build attribute for the joined pairs (negative tid = outer-join padding = NULL)
  void gatherNulls(attribute_id attr, const void *build_tids, std::int64_t n, std::uint64_t *dst) const {
    std::vector<const std::uint64_t *> segs;
    for (const BlockReference &b : refs) segs.push_back(b->nullBitmap(attr));
    CheckStatus(qsx_bitmap_gather_segmented(static_cast<int>(segs.size()), segs.data(), first_rows.data(),
                                            static_cast<const std::int32_t *>(build_tids), n, dst, CurrentStream()),
                "qsx_bitmap_gather_segmented");
  }
};

// Joined pairs of one probe block, on device.
struct JoinedPairs {
  std::unique_ptr<DeviceBuffer> probe_tids, build_tids;
  std::int64_t count = 0;
};

// Keep the pairs set in `bitmap` (order preserving).
void CompactPairs(JoinedPairs *pairs, const void *bitmap) {
  const std::int64_t n = pairs->count;
  std::unique_ptr<DeviceBuffer> p(new DeviceBuffer(static_cast<std::size_t>(n) * 4 + 8)), b(new DeviceBuffer(static_cast<std::size_t>(n) * 4 + 8));
  const void *src[2] = {pairs->probe_tids->ptr, pairs->build_tids->ptr};
  void *dst[2] = {p->ptr, b->ptr};
  const std::int32_t widths[2] = {4, 4};
  const std::size_t ws_bytes = qsx_compact_workspace_bytes(n);
  DeviceBuffer ws(ws_bytes), count(8);
  CheckStatus(qsx_compact_gather(2, src, widths, static_cast<const std::uint64_t *>(bitmap), n, dst,
                                 static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()),
              "qsx_compact_gather(pairs)");
  pairs->count = ReadCount(count.ptr);
  pairs->probe_tids = std::move(p);
  pairs->build_tids = std::move(b);
}
}  // namespace

bool HashInnerJoinWorkOrder::prefersExclusiveDevice() const {
  constexpr std::int64_t kRows = 4ll << 20;   // (below that a probe is over before the other streams have drained)
  std::int64_t rows = 0;
  for (block_id id : run_block_ids_) {
    rows += storage_manager_->getBlock(id)->numTuples();
    if (rows >= kRows) return true;
  }
  return false;
}

void HashInnerJoinWorkOrder::execute() {
  if (run_block_ids_.empty()) {
    executeBlock(block_id_);
    return;
  }
  if (executeRun()) return;
  for (block_id id : run_block_ids_) executeBlock(id);
}

// A run of probe blocks as one unit: one counting and one pair-emitting launch over all blocks (probe tuple ids are
// run-global row numbers), then every output attribute is one segmented gather — the probe side from the run's own
// stripes, the build side from the build relation's blocks — into ONE output block.
bool HashInnerJoinWorkOrder::executeRun() {
  using JoinType = HashJoinOperator::JoinType;
  const bool existence = join_type_ == JoinType::kLeftSemiJoin || join_type_ == JoinType::kLeftAntiJoin;
  const bool outer = join_type_ == JoinType::kLeftOuterJoin;
  if (join_type_ != JoinType::kInnerJoin && !existence && !outer) return false;
  if (outer && residual_predicate_ != nullptr) return false;   // (HashOuterJoinWorkOrder takes none either)
  // semi / anti with a residual predicate: the pairs of the run, the residual on them, then the probe tuples that kept (semi)
  // or never had (anti) a pair — HashSemiJoinWorkOrder / HashAntiJoinWorkOrder::executeWithResidualPredicate (:680-793, :880-1000)
  const bool existence_by_pairs = existence && residual_predicate_ != nullptr;
  int key_bits = 0;
  for (attribute_id a : join_key_attributes_) {   // (a CHAR(n <= 8) component travels as a LONG)
    key_bits += probe_relation_.getAttributeType(a).id == kChar ? 64 : probe_relation_.getAttributeType(a).width * 8;
  }
  const bool hashed_key = join_key_attributes_.size() > 1 && key_bits > 64;   // the fold is a hash: pairs need their components compared
  if (hashed_key && existence) return false;
  std::vector<BlockReference> blocks;
  std::vector<std::int64_t> rows, first_rows;
  std::int64_t total_rows = 0;
  for (block_id id : run_block_ids_) {
    blocks.push_back(storage_manager_->getBlock(id));
    const StorageBlock &b = *blocks.back();
    for (attribute_id a : join_key_attributes_) {
      if (b.nullBitmap(a) != nullptr) return false;   // (compressed key: stripe() decodes once)
    }
    for (std::size_t i = 0; i < selection_.size(); ++i) {
      if (!is_selection_on_build_[i] && b.nullBitmap(selection_[i]) != nullptr) return false;
    }
    rows.push_back(b.numTuples());
    first_rows.push_back(total_rows);
    total_rows += b.numTuples();
    // (the pairs of a semi / anti join end up as ONE bitmap over the run's tuple ids: every block starts at a word boundary,
    // so that its part of the bitmap is the bitmap of the block)
    if (existence_by_pairs || outer) total_rows = (total_rows + 63) / 64 * 64;
  }
  if (total_rows > INT32_MAX || blocks.size() > 16384) return false;
  const std::int64_t nb = static_cast<std::int64_t>(blocks.size());
  // The terms evaluated on the pair list (component equalities of a hashed composite key, residual conjuncts): the run form
  // gathers their operands with widths of 1 / 2 / 4 / 8 bytes and compares numeric types — a nullable or CHAR(n) operand
  // sends the whole work order to the block-by-block form BEFORE any device work is done for the run.
  {
    auto operand_ok = [&](attribute_id attr, bool on_build, bool against_literal) {
      const Type &t = (on_build ? build_relation_ : probe_relation_).getAttributeType(attr);
      // (a CHAR(n) attribute against a literal is compared by qsx_select_cmp_char; attribute against attribute is numeric)
      return !t.nullable && t.id != kVarChar && (t.id != kChar || against_literal);
    };
    if (hashed_key) {
      for (std::size_t k = 0; k < join_key_attributes_.size(); ++k) {
        if (!operand_ok(join_key_attributes_[k], false, false) || !operand_ok(build_key_attributes_[k], true, false)) return false;
      }
    }
    if (residual_predicate_ != nullptr) {
      for (const ComparisonPredicate &term : residual_predicate_->conjuncts) {
        const bool literal = term.rhs_attribute == kInvalidAttributeID;
        if (!operand_ok(term.attribute, term.on_build_side, literal)) return false;
        if (!literal && !operand_ok(term.rhs_attribute, term.rhs_on_build_side, false)) return false;
      }
    }
  }
  const RunJoinKeys run_keys(blocks, join_key_attributes_, rows);   // composite key: one launch packs the run's keys
  const std::vector<const void *> &keys = run_keys.ptr;
  // existence_map of the LIPFilterAdaptiveProber (:462-470): the probe tuples this work order looks up
  struct OwnedStorage {
    void *ptr = nullptr;
    ~OwnedStorage() { qsx_device_free(ptr); }
  } lip_storage;
  std::vector<const std::uint64_t *> lip_bitmaps;
  // (anti join with a residual under a LIP filter: the tuples the filter rejects must not come back through the complement;
  // that combination stays block by block)
  if (existence_by_pairs && join_type_ == JoinType::kLeftAntiJoin && lip_filter_adaptive_prober_ != nullptr) return false;
  if (lip_filter_adaptive_prober_ != nullptr && !lip_filter_adaptive_prober_->filterBlocks(blocks, &lip_storage.ptr, &lip_bitmaps)) return false;
  const std::uint64_t *const *lookup = lip_bitmaps.empty() ? nullptr : lip_bitmaps.data();
  DeviceBuffer count(8);
  if (existence && !existence_by_pairs) {
    // HashSemiJoinWorkOrder / HashAntiJoinWorkOrder without residual (:795-816, :860-877): the probe tuples with / without a
    // match, projected on the probe attributes — one existence probe and one compaction over the run
    std::size_t words = 0;
    for (std::int64_t r : rows) words += static_cast<std::size_t>((r + 63) / 64) + 1;
    DeviceBuffer bitmap_storage(words * 8 + 8);
    std::vector<std::uint64_t *> bitmaps(blocks.size());
    std::size_t at = 0;
    for (std::size_t b = 0; b < blocks.size(); ++b) {
      bitmaps[b] = static_cast<std::uint64_t *>(bitmap_storage.ptr) + at;
      at += static_cast<std::size_t>((rows[b] + 63) / 64) + 1;
    }
    CheckStatus(qsx_join_probe_exists_blocks(hash_table_, nb, rows.data(), keys.data(), lookup, join_type_ == JoinType::kLeftAntiJoin ? 1 : 0,
                                             bitmaps.data(), static_cast<std::int64_t *>(count.ptr), CurrentStream()),
                "qsx_join_probe_exists_blocks");
    const std::int64_t selected = ReadCount(count.ptr);
    block_id out_id;
    BlockReference out = output_destination_->getBlockForInsertion(selected > 0 ? selected : 1, &out_id);
    std::vector<const void *> src(blocks.size() * selection_.size());
    std::vector<void *> dst;
    std::vector<std::int32_t> widths;
    for (std::size_t i = 0; i < selection_.size(); ++i) {
      dst.push_back(out->stripe(static_cast<attribute_id>(i)));
      widths.push_back(probe_relation_.getAttributeType(selection_[i]).width);
      for (std::size_t b = 0; b < blocks.size(); ++b) src[b * selection_.size() + i] = blocks[b]->stripe(selection_[i]);
    }
    const std::size_t ws_bytes = qsx_compact_blocks_workspace_bytes(nb, rows.data());
    DeviceBuffer ws(ws_bytes + 8);
    CheckStatus(qsx_compact_gather_blocks(static_cast<int>(selection_.size()), widths.data(), nb, rows.data(), src.data(),
                                          reinterpret_cast<const std::uint64_t *const *>(bitmaps.data()), nullptr, dst.data(), nullptr,
                                          static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()),
                "qsx_compact_gather_blocks");
    const std::int64_t written = ReadCount(count.ptr);
    output_destination_->returnBlock(out_id, written, getPartitionId());
    return true;
  }
  BuildSegments build(build_relation_, storage_manager_);
  // Nothing is evaluated on the pairs (an exact key, no residual predicate) and every output attribute is a plain value of
  // 1 / 2 / 4 / 8 bytes: the probe writes the output tuples itself (qsx_join_probe_project_blocks) into a block with room for
  // one match per probe tuple.  More matches than that (duplicate build keys) and the work order takes the pair list below.
  bool projectable = !outer && run_keys.exact && residual_predicate_ == nullptr && !selection_.empty() &&
                     selection_.size() <= QSX_MAX_PROJECTED && total_rows > 0;
  for (std::size_t i = 0; i < selection_.size() && projectable; ++i) {
    const Type &t = (is_selection_on_build_[i] ? build_relation_ : probe_relation_).getAttributeType(selection_[i]);
    projectable = !t.nullable && (t.width == 1 || t.width == 2 || t.width == 4 || t.width == 8);
  }
  if (projectable) {
    const std::size_t nc = selection_.size(), nseg = build.refs.size();
    qsx_join_projection_t proj{};
    proj.num_columns = static_cast<std::int32_t>(nc);
    std::vector<const void *> probe_stripes(blocks.size() * nc, nullptr), build_stripes(nseg * nc, nullptr);
    std::vector<void *> out_columns(nc);
    block_id out_id;
    BlockReference out = output_destination_->getBlockForInsertion(total_rows, &out_id);
    for (std::size_t i = 0; i < nc; ++i) {
      const bool on_build = is_selection_on_build_[i];
      proj.width[i] = (on_build ? build_relation_ : probe_relation_).getAttributeType(selection_[i]).width;
      proj.on_build[i] = on_build ? 1 : 0;
      out_columns[i] = out->stripe(static_cast<attribute_id>(i));
      if (on_build) {
        for (std::size_t sg = 0; sg < nseg; ++sg) build_stripes[sg * nc + i] = build.refs[sg]->stripe(selection_[i]);
      } else {
        for (std::size_t b = 0; b < blocks.size(); ++b) probe_stripes[b * nc + i] = blocks[b]->stripe(selection_[i]);
      }
    }
    proj.probe_stripes = probe_stripes.data();
    proj.num_build_segments = static_cast<std::int32_t>(nseg);
    proj.build_first_tids = build.first_rows.data();
    proj.build_stripes = build_stripes.data();
    proj.out_columns = out_columns.data();
    CheckStatus(qsx_join_probe_project_blocks(hash_table_, nb, rows.data(), keys.data(), lookup, &proj, total_rows,
                                              static_cast<std::int64_t *>(count.ptr), CurrentStream()),
                "qsx_join_probe_project_blocks");
    const std::int64_t matches = ReadCount(count.ptr);   // (synchronises the stream: the block's tuples are written)
    if (matches <= total_rows) {
      output_destination_->returnBlock(out_id, matches, getPartitionId());
      return true;
    }
    out = BlockReference();
    storage_manager_->deleteBlockOrBlobFile(out_id);   // never returned to the destination: nobody else knows the block
  }
  // No counting pass: the pair lists get room for one match per probe tuple — what a foreign-key probe of a primary-key
  // build side produces at most (the reference sizes from the same uniqueness fact, impliesUniqueAttributes).  The probe
  // counts every match it finds, also those that did not fit: a build side with duplicate keys makes this work order probe
  // once more with the exact capacity.
  JoinedPairs pairs;
  std::int64_t room = total_rows > 0 ? total_rows : 1;
  std::vector<std::int32_t> base_tids;   // semi / anti by pairs: the word-aligned first tuple id of every block (first_rows)
  if (existence_by_pairs || outer) base_tids.assign(first_rows.begin(), first_rows.end());
  for (int attempt = 0; attempt < 2; ++attempt) {
    pairs.probe_tids.reset(new DeviceBuffer(static_cast<std::size_t>(room) * 4 + 8));
    pairs.build_tids.reset(new DeviceBuffer(static_cast<std::size_t>(room) * 4 + 8));
    CheckStatus(qsx_join_probe_blocks(hash_table_, nb, rows.data(), keys.data(), base_tids.empty() ? nullptr : base_tids.data(), lookup,
                                      static_cast<std::int32_t *>(pairs.probe_tids->ptr),
                                      static_cast<std::int32_t *>(pairs.build_tids->ptr), room, static_cast<std::int64_t *>(count.ptr),
                                      CurrentStream()), "qsx_join_probe_blocks");
    pairs.count = ReadCount(count.ptr);
    if (pairs.count <= room) break;
    if (attempt == 1) throw ExecutionError("HashJoinOperator: the match count changed between two probes of one run", QSX_ERR_CAPACITY);
    room = pairs.count;
  }
  std::vector<const void *> segments(blocks.size());
  std::vector<ComparisonPredicate> terms;
  if (!run_keys.exact) {   // compositeKeyCollisionCheck (SeparateChainingHashTable.hpp:1046): equal folds, equal components?
    for (std::size_t k = 0; k < join_key_attributes_.size(); ++k) {
      terms.push_back(ComparisonPredicate::Attributes(join_key_attributes_[k], false, ComparisonID::kEqual, build_key_attributes_[k], true));
    }
  }
  if (residual_predicate_ != nullptr) terms.insert(terms.end(), residual_predicate_->conjuncts.begin(), residual_predicate_->conjuncts.end());
  if (!terms.empty() && pairs.count > 0) {
    // matchesForJoinedTuples (:510-524) on the pairs of the run: each term's operands gathered by the pair lists (the probe
    // side through the run's own stripes), compared, chained through the filter bitmap like a conjunction
    const std::int64_t m = pairs.count;
    const std::size_t pair_bitmap_bytes = static_cast<std::size_t>((m + 63) / 64) * 8 + 8;
    std::size_t widest = 8;
    for (const ComparisonPredicate &term : terms) {
      widest = std::max<std::size_t>(widest, (term.on_build_side ? build_relation_ : probe_relation_).getAttributeType(term.attribute).width);
    }
    DeviceBuffer current(pair_bitmap_bytes), next(pair_bitmap_bytes), lhs(static_cast<std::size_t>(m) * widest + 8), rhs(static_cast<std::size_t>(m) * 8 + 8);
    void *cur = current.ptr, *nxt = next.ptr;
    bool first = true;
    auto gather_side = [&](attribute_id attr, bool on_build, void *dst) -> Type {
      const Type t = (on_build ? build_relation_ : probe_relation_).getAttributeType(attr);
      if (on_build) {
        build.gather(attr, t.width, pairs.build_tids->ptr, m, dst);
      } else {
        for (std::size_t b = 0; b < blocks.size(); ++b) segments[b] = blocks[b]->stripe(attr);
        CheckStatus(qsx_gather_segmented(t.width, static_cast<int>(segments.size()), segments.data(), first_rows.data(),
                                         static_cast<const std::int32_t *>(pairs.probe_tids->ptr), m, dst, CurrentStream()),
                    "qsx_gather_segmented");
      }
      return t;
    };
    for (const ComparisonPredicate &term : terms) {
      const Type t = gather_side(term.attribute, term.on_build_side, lhs.ptr);   // (operand types were vetted above)
      if (term.rhs_attribute != kInvalidAttributeID) {
        const Type rt = gather_side(term.rhs_attribute, term.rhs_on_build_side, rhs.ptr);
        if (rt.id != t.id) throw ExecutionError("join predicate compares attributes of different types", QSX_ERR_UNSUPPORTED);
        CheckStatus(qsx_select_cmp_columns(t.id, lhs.ptr, rhs.ptr, m, static_cast<int>(term.comparison),
                                           first ? nullptr : static_cast<const std::uint64_t *>(cur), static_cast<std::uint64_t *>(nxt), nullptr,
                                           CurrentStream()), "qsx_select_cmp_columns");
      } else if (t.id == kChar) {
        CheckStatus(qsx_select_cmp_char(lhs.ptr, t.width, m, static_cast<int>(term.comparison), term.literal.text.data(),
                                        static_cast<int>(term.literal.text.size()), first ? nullptr : static_cast<const std::uint64_t *>(cur),
                                        static_cast<std::uint64_t *>(nxt), nullptr, CurrentStream()), "qsx_select_cmp_char");
      } else {
        CheckStatus(qsx_select_cmp(t.id, lhs.ptr, m, static_cast<int>(term.comparison), &term.literal.v,
                                   first ? nullptr : static_cast<const std::uint64_t *>(cur), static_cast<std::uint64_t *>(nxt), nullptr,
                                   CurrentStream()), "qsx_select_cmp");
      }
      std::swap(cur, nxt);
      first = false;
    }
    if (!first) CompactPairs(&pairs, cur);
  }
  if (existence_by_pairs) {
    // the probe tuples that kept a pair, as one bitmap over the run's (word-aligned) tuple ids; anti: its complement, block by
    // block (the complement of a block's part ends at the block's last tuple); then one compaction over the run
    const std::size_t words = static_cast<std::size_t>(total_rows / 64) + 1;
    DeviceBuffer bitmap(words * 8 + 8);
    CheckStatus(qsx_tids_to_bitmap(static_cast<const std::int32_t *>(pairs.probe_tids->ptr), pairs.count, 0, total_rows,
                                   static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_tids_to_bitmap");
    std::vector<std::uint64_t *> bitmaps(blocks.size());
    std::int64_t upper = 0;   // tuples the compaction can select at most
    for (std::size_t b = 0; b < blocks.size(); ++b) {
      bitmaps[b] = static_cast<std::uint64_t *>(bitmap.ptr) + first_rows[b] / 64;
      if (join_type_ == JoinType::kLeftAntiJoin && rows[b] > 0) {
        CheckStatus(qsx_bitmap_combine(3, bitmaps[b], nullptr, rows[b], bitmaps[b], CurrentStream()), "qsx_bitmap_combine");
      }
      upper += rows[b];
    }
    if (join_type_ == JoinType::kLeftSemiJoin) upper = std::min<std::int64_t>(upper, pairs.count);
    block_id out_id;
    BlockReference out = output_destination_->getBlockForInsertion(upper > 0 ? upper : 1, &out_id);
    std::vector<const void *> src(blocks.size() * selection_.size());
    std::vector<void *> dst;
    std::vector<std::int32_t> widths;
    for (std::size_t i = 0; i < selection_.size(); ++i) {
      dst.push_back(out->stripe(static_cast<attribute_id>(i)));
      widths.push_back(probe_relation_.getAttributeType(selection_[i]).width);
      for (std::size_t b = 0; b < blocks.size(); ++b) src[b * selection_.size() + i] = blocks[b]->stripe(selection_[i]);
    }
    const std::size_t ws_bytes = qsx_compact_blocks_workspace_bytes(nb, rows.data());
    DeviceBuffer ws(ws_bytes + 8);
    CheckStatus(qsx_compact_gather_blocks(static_cast<int>(selection_.size()), widths.data(), nb, rows.data(), src.data(),
                                          reinterpret_cast<const std::uint64_t *const *>(bitmaps.data()), nullptr, dst.data(), nullptr,
                                          static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()),
                "qsx_compact_gather_blocks");
    const std::int64_t written = ReadCount(count.ptr);
    output_destination_->returnBlock(out_id, written, getPartitionId());
    return true;
  }
  if (outer) {
    // HashOuterJoinWorkOrder (:1026-1099) over the run: the matched pairs, then the probe tuples without one (the complement of
    // the matched tuples' bitmap, block by block so that it ends at each block's last tuple, AND the LIP filter's survivors)
    // with NULL build-side attributes
    const std::int64_t matches = pairs.count;
    const std::size_t words = static_cast<std::size_t>(total_rows / 64) + 1;
    DeviceBuffer bitmap(words * 8 + 8);
    CheckStatus(qsx_tids_to_bitmap(static_cast<const std::int32_t *>(pairs.probe_tids->ptr), matches, 0, total_rows,
                                   static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_tids_to_bitmap");
    for (std::size_t b = 0; b < blocks.size(); ++b) {
      if (rows[b] == 0) continue;
      std::uint64_t *part = static_cast<std::uint64_t *>(bitmap.ptr) + first_rows[b] / 64;
      CheckStatus(qsx_bitmap_combine(3, part, nullptr, rows[b], part, CurrentStream()), "qsx_bitmap_combine");
      if (lookup != nullptr && lookup[b] != nullptr) {
        CheckStatus(qsx_bitmap_combine(0, part, lookup[b], rows[b], part, CurrentStream()), "qsx_bitmap_combine");
      }
    }
    DeviceBuffer unmatched_tids(static_cast<std::size_t>(total_rows) * 4 + 8);
    const std::size_t ws_bytes = qsx_compact_workspace_bytes(total_rows);
    DeviceBuffer ws(ws_bytes + 8);
    CheckStatus(qsx_bitmap_to_tids(static_cast<const std::uint64_t *>(bitmap.ptr), total_rows, 0, static_cast<std::int32_t *>(unmatched_tids.ptr),
                                   static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()), "qsx_bitmap_to_tids");
    const std::int64_t unmatched = ReadCount(count.ptr);
    const std::int64_t total = matches + unmatched;
    block_id out_id;
    BlockReference out = output_destination_->getBlockForInsertion(total > 0 ? total : 1, &out_id);
    std::unique_ptr<DeviceBuffer> padded_build_tids;   // the pairs' build tuple ids, then -1 for every unmatched probe tuple
    for (std::size_t i = 0; i < selection_.size(); ++i) {
      char *dst = static_cast<char *>(out->stripe(static_cast<attribute_id>(i)));
      const bool on_build = is_selection_on_build_[i];
      const Type &t = (on_build ? build_relation_ : probe_relation_).getAttributeType(selection_[i]);
      char *tail = dst + static_cast<std::size_t>(matches) * t.width;
      if (on_build) {
        std::uint64_t *nulls = out->nullBitmap(static_cast<attribute_id>(i));
        if (nulls == nullptr) throw ExecutionError("outer join output attribute taken from the build side must be nullable", QSX_ERR_INVALID_ARGUMENT);
        if (matches > 0) build.gather(selection_[i], t.width, pairs.build_tids->ptr, matches, dst);
        if (unmatched > 0) CheckStatus(qsx_memset_device(tail, 0, static_cast<std::size_t>(unmatched) * t.width, CurrentStream()), "qsx_memset_device");
        if (total > 0) {
          if (padded_build_tids == nullptr) {
            padded_build_tids.reset(new DeviceBuffer(static_cast<std::size_t>(total) * 4 + 8));
            if (matches > 0) {
              CheckStatus(qsx_copy_on_device(padded_build_tids->ptr, pairs.build_tids->ptr, static_cast<std::size_t>(matches) * 4, CurrentStream()),
                          "qsx_copy_on_device");
            }
            if (unmatched > 0) {
              CheckStatus(qsx_memset_device(static_cast<char *>(padded_build_tids->ptr) + static_cast<std::size_t>(matches) * 4, 0xFF,
                                            static_cast<std::size_t>(unmatched) * 4, CurrentStream()), "qsx_memset_device");
            }
          }
          build.gatherNulls(selection_[i], padded_build_tids->ptr, total, nulls);
        }
      } else {
        for (std::size_t b = 0; b < blocks.size(); ++b) segments[b] = blocks[b]->stripe(selection_[i]);
        if (matches > 0) {
          CheckStatus(qsx_gather_segmented(t.width, static_cast<int>(segments.size()), segments.data(), first_rows.data(),
                                           static_cast<const std::int32_t *>(pairs.probe_tids->ptr), matches, dst, CurrentStream()),
                      "qsx_gather_segmented");
        }
        if (unmatched > 0) {
          CheckStatus(qsx_gather_segmented(t.width, static_cast<int>(segments.size()), segments.data(), first_rows.data(),
                                           static_cast<const std::int32_t *>(unmatched_tids.ptr), unmatched, tail, CurrentStream()),
                      "qsx_gather_segmented");
        }
      }
    }
    CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
    output_destination_->returnBlock(out_id, total, getPartitionId());
    return true;
  }
  const std::int64_t matches = pairs.count;
  DeviceBuffer &probe_tids = *pairs.probe_tids, &build_tids = *pairs.build_tids;
  block_id out_id;
  BlockReference out = output_destination_->getBlockForInsertion(matches > 0 ? matches : 1, &out_id);
  for (std::size_t i = 0; i < selection_.size(); ++i) {
    void *dst = out->stripe(static_cast<attribute_id>(i));
    const bool on_build = is_selection_on_build_[i];
    const Type &t = (on_build ? build_relation_ : probe_relation_).getAttributeType(selection_[i]);
    if (on_build) {
      build.gather(selection_[i], t.width, build_tids.ptr, matches, dst);
      if (t.nullable && matches > 0) {
        std::uint64_t *nulls = out->nullBitmap(static_cast<attribute_id>(i));
        if (nulls == nullptr) throw ExecutionError("join output of a nullable attribute must be nullable", QSX_ERR_INVALID_ARGUMENT);
        build.gatherNulls(selection_[i], build_tids.ptr, matches, nulls);
      }
    } else {
      for (std::size_t b = 0; b < blocks.size(); ++b) segments[b] = blocks[b]->stripe(selection_[i]);
      CheckStatus(qsx_gather_segmented(t.width, static_cast<int>(segments.size()), segments.data(), first_rows.data(),
                                       static_cast<const std::int32_t *>(probe_tids.ptr), matches, dst, CurrentStream()),
                  "qsx_gather_segmented");
    }
  }
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  output_destination_->returnBlock(out_id, matches, getPartitionId());
  return true;
}

void HashInnerJoinWorkOrder::executeBlock(block_id probe_block_id) {
  using JoinType = HashJoinOperator::JoinType;
  BlockReference probe = storage_manager_->getBlock(probe_block_id);
  const std::int64_t n = probe->numTuples();
  JoinKeys keys(*probe, join_key_attributes_);
  DeviceBuffer count(8);
  const std::size_t bitmap_bytes = static_cast<std::size_t>((n + 63) / 64) * 8 + 8;
  // existence_map of the LIPFilterAdaptiveProber (:462-470): the probe tuples this work order looks at
  struct LipBitmap {
    void *ptr = nullptr;
    ~LipBitmap() { qsx_device_free(ptr); }
  } lip_holder;
  if (lip_filter_adaptive_prober_ != nullptr) lip_holder.ptr = lip_filter_adaptive_prober_->filterValueAccessor(*probe, nullptr, nullptr);
  const std::uint64_t *lip = static_cast<const std::uint64_t *>(lip_holder.ptr);
  // check_for_null_keys: a probe tuple with a NULL key component is not looked up (HashTable.hpp:2158-2160, 1855-1865):
  // it matches nothing — out of an inner / semi join, NULL-padded in an outer join, kept by an anti join.
  // `lookup` = the tuples that are looked up, `lip` stays the set of tuples this work order is about.
  const std::unique_ptr<DeviceBuffer> not_null_keys = NotNullFilter(*probe, join_key_attributes_, lip);
  const std::uint64_t *lookup = not_null_keys != nullptr ? static_cast<const std::uint64_t *>(not_null_keys->ptr) : lip;
  const bool pairs_needed = join_type_ == JoinType::kInnerJoin || join_type_ == JoinType::kLeftOuterJoin ||
                            residual_predicate_ != nullptr || !keys.exact;

  JoinedPairs pairs;
  std::unique_ptr<BuildSegments> build;
  if (pairs_needed) {
    // hash_table_.getAllFromValueAccessor[CompositeKey](accessor, key(s), nullable, &collector) (:480-485)
    CheckStatus(qsx_join_probe_count(hash_table_, keys.ptr, n, lookup, static_cast<std::int64_t *>(count.ptr), CurrentStream()),
                "qsx_join_probe_count");
    pairs.count = ReadCount(count.ptr);
    pairs.probe_tids.reset(new DeviceBuffer(static_cast<std::size_t>(pairs.count) * 4 + 8));
    pairs.build_tids.reset(new DeviceBuffer(static_cast<std::size_t>(pairs.count) * 4 + 8));
    CheckStatus(qsx_join_probe(hash_table_, keys.ptr, n, /*probe_base_tid=*/0, lookup,
                               static_cast<std::int32_t *>(pairs.probe_tids->ptr), static_cast<std::int32_t *>(pairs.build_tids->ptr),
                               pairs.count, static_cast<std::int64_t *>(count.ptr), CurrentStream()), "qsx_join_probe");
    build.reset(new BuildSegments(build_relation_, storage_manager_));

    // Terms evaluated on the pairs: the component equalities of a hashed composite key
    // (compositeKeyCollisionCheck, SeparateChainingHashTable.hpp:1046) and the residual predicate
    // (matchesForJoinedTuples, :510-524), chained through the filter bitmap like a conjunction.
    std::vector<ComparisonPredicate> terms;
    if (!keys.exact) {
      for (std::size_t k = 0; k < join_key_attributes_.size(); ++k) {
        terms.push_back(ComparisonPredicate::Attributes(join_key_attributes_[k], false, ComparisonID::kEqual,
                                                        build_key_attributes_[k], true));
      }
    }
    if (residual_predicate_ != nullptr) {
      terms.insert(terms.end(), residual_predicate_->conjuncts.begin(), residual_predicate_->conjuncts.end());
    }
    if (!terms.empty() && pairs.count > 0) {
      const std::int64_t m = pairs.count;
      const std::size_t pair_bitmap_bytes = static_cast<std::size_t>((m + 63) / 64) * 8 + 8;
      std::size_t widest = 8;
      for (const ComparisonPredicate &term : terms) {
        widest = std::max<std::size_t>(widest, (term.on_build_side ? build_relation_ : probe_relation_).getAttributeType(term.attribute).width);
      }
      DeviceBuffer current(pair_bitmap_bytes), next(pair_bitmap_bytes), lhs(static_cast<std::size_t>(m) * widest + 8),
          rhs(static_cast<std::size_t>(m) * 8 + 8);
      void *cur = current.ptr, *nxt = next.ptr;
      bool first = true;
      auto gather_side = [&](attribute_id attr, bool on_build, void *dst) -> Type {
        const Type t = (on_build ? build_relation_ : probe_relation_).getAttributeType(attr);
        if (on_build) {
          build->gather(attr, t.width, pairs.build_tids->ptr, m, dst);
        } else {
          CheckStatus(qsx_gather(t.width, probe->stripe(attr), static_cast<const std::int32_t *>(pairs.probe_tids->ptr), m, dst,
                                 CurrentStream()), "qsx_gather");
        }
        return t;
      };
      for (const ComparisonPredicate &term : terms) {
        const Type t = gather_side(term.attribute, term.on_build_side, lhs.ptr);
        if (term.rhs_attribute != kInvalidAttributeID) {
          if (t.id == kChar || t.id == kVarChar) {
            throw ExecutionError("join predicate compares two string attributes (only string = literal is supported)", QSX_ERR_UNSUPPORTED);
          }
          const Type rt = gather_side(term.rhs_attribute, term.rhs_on_build_side, rhs.ptr);
          if (rt.id != t.id) throw ExecutionError("join predicate compares attributes of different types", QSX_ERR_UNSUPPORTED);
          CheckStatus(qsx_select_cmp_columns(t.id, lhs.ptr, rhs.ptr, m, static_cast<int>(term.comparison),
                                             first ? nullptr : static_cast<const std::uint64_t *>(cur),
                                             static_cast<std::uint64_t *>(nxt), nullptr, CurrentStream()),
                      "qsx_select_cmp_columns");
        } else if (t.id == kChar) {
          CheckStatus(qsx_select_cmp_char(lhs.ptr, t.width, m, static_cast<int>(term.comparison), term.literal.text.data(),
                                          static_cast<int>(term.literal.text.size()), first ? nullptr : static_cast<const std::uint64_t *>(cur),
                                          static_cast<std::uint64_t *>(nxt), nullptr, CurrentStream()), "qsx_select_cmp_char");
        } else {
          CheckStatus(qsx_select_cmp(t.id, lhs.ptr, m, static_cast<int>(term.comparison), &term.literal.v,
                                     first ? nullptr : static_cast<const std::uint64_t *>(cur),
                                     static_cast<std::uint64_t *>(nxt), nullptr, CurrentStream()), "qsx_select_cmp");
        }
        std::swap(cur, nxt);
        first = false;
      }
      CompactPairs(&pairs, cur);
    }
  }

  if (join_type_ == JoinType::kLeftSemiJoin || join_type_ == JoinType::kLeftAntiJoin) {
    // HashSemiJoinWorkOrder / HashAntiJoinWorkOrder (:680-877, :880-1000): the probe tuples with
    // (semi) / without (anti) a surviving match, projected on the probe attributes.
    const bool anti = join_type_ == JoinType::kLeftAntiJoin;
    DeviceBuffer bitmap(bitmap_bytes);
    if (pairs_needed) {
      CheckStatus(qsx_tids_to_bitmap(static_cast<const std::int32_t *>(pairs.probe_tids->ptr), pairs.count, 0, n,
                                     static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_tids_to_bitmap");
      if (anti) {
        CheckStatus(qsx_bitmap_combine(3, static_cast<const std::uint64_t *>(bitmap.ptr), nullptr, n,
                                       static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_bitmap_combine");
        if (lip != nullptr) {
          CheckStatus(qsx_bitmap_combine(0, static_cast<const std::uint64_t *>(bitmap.ptr), lip, n,
                                         static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_bitmap_combine");
        }
      }
      CheckStatus(qsx_bitmap_count(static_cast<const std::uint64_t *>(bitmap.ptr), n, static_cast<std::int64_t *>(count.ptr),
                                   CurrentStream()), "qsx_bitmap_count");
    } else if (anti && lookup != lip) {
      // NULL keys are not looked up and therefore survive the anti join: tuples \ (looked-up tuples with a match)
      CheckStatus(qsx_join_probe_exists(hash_table_, keys.ptr, n, lookup, 0, static_cast<std::uint64_t *>(bitmap.ptr),
                                        static_cast<std::int64_t *>(count.ptr), CurrentStream()), "qsx_join_probe_exists");
      CheckStatus(qsx_bitmap_combine(3, static_cast<const std::uint64_t *>(bitmap.ptr), nullptr, n,
                                     static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_bitmap_combine");
      if (lip != nullptr) {
        CheckStatus(qsx_bitmap_combine(0, static_cast<const std::uint64_t *>(bitmap.ptr), lip, n,
                                       static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_bitmap_combine");
      }
      CheckStatus(qsx_bitmap_count(static_cast<const std::uint64_t *>(bitmap.ptr), n, static_cast<std::int64_t *>(count.ptr),
                                   CurrentStream()), "qsx_bitmap_count");
    } else {
      CheckStatus(qsx_join_probe_exists(hash_table_, keys.ptr, n, lookup, anti ? 1 : 0,
                                        static_cast<std::uint64_t *>(bitmap.ptr), static_cast<std::int64_t *>(count.ptr),
                                        CurrentStream()), "qsx_join_probe_exists");
    }
    const std::int64_t matches = ReadCount(count.ptr);
    block_id out_id;
    BlockReference out = output_destination_->getBlockForInsertion(matches > 0 ? matches : 1, &out_id);
    std::vector<const void *> src;
    std::vector<void *> dst;
    std::vector<std::int32_t> widths;
    for (std::size_t i = 0; i < selection_.size(); ++i) {
      src.push_back(probe->stripe(selection_[i]));
      dst.push_back(out->stripe(static_cast<attribute_id>(i)));
      widths.push_back(probe_relation_.getAttributeType(selection_[i]).width);
    }
    const std::size_t ws_bytes = qsx_compact_workspace_bytes(n);
    DeviceBuffer ws(ws_bytes);
    CheckStatus(qsx_compact_gather(static_cast<int>(src.size()), src.data(), widths.data(),
                                   static_cast<const std::uint64_t *>(bitmap.ptr), n, dst.data(),
                                   static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()),
                "qsx_compact_gather");
    const std::int64_t written = ReadCount(count.ptr);
    ProjectNullBitmaps(*probe, selection_, bitmap.ptr, written, out.get());
    output_destination_->returnBlock(out_id, written, getPartitionId());
    return;
  }

  // Inner / left outer: matched pairs first, then (outer) the probe tuples without a match with
  // NULL build-side attributes (HashOuterJoinWorkOrder, :1026-1099).
  const std::int64_t matches = pairs.count;
  std::int64_t unmatched = 0;
  std::unique_ptr<DeviceBuffer> unmatched_tids;
  if (join_type_ == JoinType::kLeftOuterJoin) {
    DeviceBuffer bitmap(bitmap_bytes);
    CheckStatus(qsx_tids_to_bitmap(static_cast<const std::int32_t *>(pairs.probe_tids->ptr), matches, 0, n,
                                   static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_tids_to_bitmap");
    CheckStatus(qsx_bitmap_combine(3, static_cast<const std::uint64_t *>(bitmap.ptr), nullptr, n,
                                   static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_bitmap_combine");
    if (lip != nullptr) {
      CheckStatus(qsx_bitmap_combine(0, static_cast<const std::uint64_t *>(bitmap.ptr), lip, n,
                                     static_cast<std::uint64_t *>(bitmap.ptr), CurrentStream()), "qsx_bitmap_combine");
    }
    unmatched_tids.reset(new DeviceBuffer(static_cast<std::size_t>(n) * 4 + 8));
    const std::size_t ws_bytes = qsx_compact_workspace_bytes(n);
    DeviceBuffer ws(ws_bytes);
    CheckStatus(qsx_bitmap_to_tids(static_cast<const std::uint64_t *>(bitmap.ptr), n, 0, static_cast<std::int32_t *>(unmatched_tids->ptr),
                                   static_cast<std::int64_t *>(count.ptr), ws.ptr, ws_bytes, CurrentStream()), "qsx_bitmap_to_tids");
    unmatched = ReadCount(count.ptr);
  }
  const std::int64_t total = matches + unmatched;
  block_id out_id;
  BlockReference out = output_destination_->getBlockForInsertion(total > 0 ? total : 1, &out_id);
  // Row numbers of all `total` output rows per side, for the null bits: pairs first, then (outer) the unmatched probe
  // tuples next to build tid -1 = NULL padding.  Only materialised when some output attribute can be NULL.
  std::unique_ptr<DeviceBuffer> all_probe_tids, all_build_tids;
  auto all_tids = [&](bool on_build) -> const void * {
    std::unique_ptr<DeviceBuffer> &buf = on_build ? all_build_tids : all_probe_tids;
    if (unmatched == 0) return on_build ? pairs.build_tids->ptr : pairs.probe_tids->ptr;
    if (buf == nullptr) {
      buf.reset(new DeviceBuffer(static_cast<std::size_t>(total) * 4 + 8));
      char *tail = static_cast<char *>(buf->ptr) + static_cast<std::size_t>(matches) * 4;
      CheckStatus(qsx_copy_on_device(buf->ptr, on_build ? pairs.build_tids->ptr : pairs.probe_tids->ptr,
                                     static_cast<std::size_t>(matches) * 4, CurrentStream()), "qsx_copy_on_device");
      if (on_build) {
        CheckStatus(qsx_memset_device(tail, 0xFF, static_cast<std::size_t>(unmatched) * 4, CurrentStream()), "qsx_memset_device");
      } else {
        CheckStatus(qsx_copy_on_device(tail, unmatched_tids->ptr, static_cast<std::size_t>(unmatched) * 4, CurrentStream()),
                    "qsx_copy_on_device");
      }
    }
    return buf->ptr;
  };
  for (std::size_t i = 0; i < selection_.size(); ++i) {
    // Scalar::getAllValuesForJoin (:529-536)
    char *dst = static_cast<char *>(out->stripe(static_cast<attribute_id>(i)));
    const bool on_build = is_selection_on_build_[i];
    const int width = (on_build ? build_relation_ : probe_relation_).getAttributeType(selection_[i]).width;
    if (on_build) {
      build->gather(selection_[i], width, pairs.build_tids->ptr, matches, dst);
    } else {
      CheckStatus(qsx_gather(width, probe->stripe(selection_[i]), static_cast<const std::int32_t *>(pairs.probe_tids->ptr), matches,
                             dst, CurrentStream()), "qsx_gather");
    }
    if (unmatched > 0) {
      char *tail = dst + static_cast<std::size_t>(matches) * width;
      if (on_build) {
        // result->fillWithNulls() (:1077-1080): zero bytes here, the null bits of rows [matches, total) below
        CheckStatus(qsx_memset_device(tail, 0, static_cast<std::size_t>(unmatched) * width, CurrentStream()), "qsx_memset_device");
        if (out->nullBitmap(static_cast<attribute_id>(i)) == nullptr) {
          throw ExecutionError("outer join output attribute taken from the build side must be nullable", QSX_ERR_INVALID_ARGUMENT);
        }
      } else {
        CheckStatus(qsx_gather(width, probe->stripe(selection_[i]), static_cast<const std::int32_t *>(unmatched_tids->ptr), unmatched,
                               tail, CurrentStream()), "qsx_gather");
      }
    }
    // null bits of the output attribute: the source attribute's bits at the joined rows, 1 under the outer join's padding
    const bool source_nullable = (on_build ? build_relation_ : probe_relation_).getAttributeType(selection_[i]).nullable;
    if (total > 0 && (source_nullable || (on_build && unmatched > 0))) {
      std::uint64_t *nulls = out->nullBitmap(static_cast<attribute_id>(i));
      if (nulls == nullptr) throw ExecutionError("join output of a nullable attribute must be nullable", QSX_ERR_INVALID_ARGUMENT);
      if (on_build) {
        build->gatherNulls(selection_[i], all_tids(true), total, nulls);
      } else {
        GatherBlockNulls(*probe, selection_[i], all_tids(false), total, nulls);
      }
    }
  }
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  output_destination_->returnBlock(out_id, total, getPartitionId());  // output_destination_->bulkInsertTuples(&temp_result) (:539)
}

namespace {
class DestroyHashWorkOrder : public WorkOrder {
 public:
  DestroyHashWorkOrder(std::size_t query_id, QueryContext::join_hash_table_id id, QueryContext *ctx, partition_id part)
      : WorkOrder(query_id, part), id_(id), ctx_(ctx) {}
  void execute() override { ctx_->destroyJoinHashTable(id_, partition_id_); }  // DestroyHashOperator.cpp:70-72
 private:
  QueryContext::join_hash_table_id id_;
  QueryContext *ctx_;
};
class AggregationWorkOrder : public WorkOrder {
 public:
  AggregationWorkOrder(std::size_t query_id, block_id input_block_id, AggregationOperationState *state,
                       StorageManager *storage_manager, partition_id part = 0, LIPFilterAdaptiveProber *prober = nullptr)
      : WorkOrder(query_id, part), input_block_id_(input_block_id), state_(state), storage_manager_(storage_manager),
        lip_filter_adaptive_prober_(prober) {}
  // a run of blocks (AggregationOperator::setBlocksPerWorkOrder)
  AggregationWorkOrder(std::size_t query_id, std::vector<block_id> input_block_ids, AggregationOperationState *state,
                       StorageManager *storage_manager, partition_id part, LIPFilterAdaptiveProber *prober)
      : WorkOrder(query_id, part), input_block_id_(input_block_ids.front()), more_block_ids_(input_block_ids.begin() + 1, input_block_ids.end()),
        state_(state), storage_manager_(storage_manager), lip_filter_adaptive_prober_(prober) {}
  void execute() override {  // AggregationOperator.cpp:124-126
    if (!more_block_ids_.empty()) {
      std::vector<BlockReference> blocks{storage_manager_->getBlock(input_block_id_)};
      for (block_id id : more_block_ids_) blocks.push_back(storage_manager_->getBlock(id));
      std::vector<const std::uint64_t *> filters(blocks.size(), nullptr);
      std::vector<void *> owned;
      if (lip_filter_adaptive_prober_ != nullptr) {
        void *storage = nullptr;
        if (lip_filter_adaptive_prober_->filterBlocks(blocks, &storage, &filters)) {   // one launch per filter over the run
          owned.push_back(storage);
        } else {
          filters.assign(blocks.size(), nullptr);
          for (std::size_t i = 0; i < blocks.size(); ++i) {
            owned.push_back(lip_filter_adaptive_prober_->filterValueAccessor(*blocks[i], nullptr, nullptr));
            filters[i] = static_cast<const std::uint64_t *>(owned.back());
          }
        }
      }
      state_->aggregateBlocks(blocks, filters);
      for (void *p : owned) qsx_device_free(p);
      return;
    }
    BlockReference block = storage_manager_->getBlock(input_block_id_);
    void *lip = nullptr;
    if (lip_filter_adaptive_prober_ != nullptr) lip = lip_filter_adaptive_prober_->filterValueAccessor(*block, nullptr, nullptr);
    state_->aggregateBlock(*block, static_cast<const std::uint64_t *>(lip));
    qsx_device_free(lip);
  }
 private:
  block_id input_block_id_;
  std::vector<block_id> more_block_ids_;
  AggregationOperationState *state_;
  StorageManager *storage_manager_;
  std::unique_ptr<LIPFilterAdaptiveProber> lip_filter_adaptive_prober_;
};
class BuildAggregationExistenceMapWorkOrder : public WorkOrder {
 public:
  BuildAggregationExistenceMapWorkOrder(std::size_t query_id, const CatalogRelation &input_relation, partition_id part,
                                        block_id build_block_id, attribute_id build_attribute, AggregationOperationState *state,
                                        StorageManager *storage_manager)
      : WorkOrder(query_id, part), input_relation_(input_relation), build_block_id_(build_block_id),
        build_attribute_(build_attribute), state_(state), storage_manager_(storage_manager) {}
  void execute() override {   // BuildAggregationExistenceMapOperator.cpp:177-208
    BlockReference block = storage_manager_->getBlock(build_block_id_);
    state_->buildExistenceMap(*block, build_attribute_, input_relation_.getAttributeType(build_attribute_));
  }
 private:
  const CatalogRelation &input_relation_;
  block_id build_block_id_;
  attribute_id build_attribute_;
  AggregationOperationState *state_;
  StorageManager *storage_manager_;
};
class FinalizeAggregationWorkOrder : public WorkOrder {
 public:
  FinalizeAggregationWorkOrder(std::size_t query_id, std::size_t part, std::size_t num_parts,
                               AggregationOperationState *state, InsertDestination *dest)
      : WorkOrder(query_id), part_(part), num_parts_(num_parts), state_(state), dest_(dest) {}
  void execute() override { state_->finalizeAggregate(part_, num_parts_, dest_); }  // FinalizeAggregationOperator.cpp:99-101
 private:
  std::size_t part_, num_parts_;
  AggregationOperationState *state_;
  InsertDestination *dest_;
};
class DestroyAggregationStateWorkOrder : public WorkOrder {
 public:
  DestroyAggregationStateWorkOrder(std::size_t query_id, QueryContext::aggregation_state_id id, QueryContext *ctx,
                                   partition_id part)
      : WorkOrder(query_id, part), id_(id), ctx_(ctx) {}
  void execute() override { ctx_->destroyAggregationState(id_, partition_id_); }
 private:
  QueryContext::aggregation_state_id id_;
  QueryContext *ctx_;
};
}  // namespace

bool DestroyHashOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context, StorageManager *,
                                           const tmb::client_id, tmb::MessageBus *) {
  if (!work_generated_) {
    work_generated_ = true;
    for (partition_id part = 0; part < num_partitions_; ++part) {   // DestroyHashOperator.cpp:40-50
      container->addNormalWorkOrder(new DestroyHashWorkOrder(query_id_, hash_table_index_, query_context, part), op_index_);
    }
  }
  return true;
}

// ---------------------------------------------------------------------------
// Aggregation
// ---------------------------------------------------------------------------
AggregationOperator::AggregationOperator(std::size_t query_id, const CatalogRelation &input_relation,
                                         bool input_relation_is_stored, QueryContext::aggregation_state_id aggr_state_index,
                                         std::size_t num_partitions)
    : RelationalOperator(query_id, num_partitions), input_relation_(input_relation),
      input_relation_is_stored_(input_relation_is_stored), aggr_state_index_(aggr_state_index), input_(num_partitions) {
  if (input_relation_is_stored) {
    if (num_partitions > 1 && input_relation.getNumPartitions() != num_partitions) {
      throw ExecutionError("AggregationOperator: num_partitions differs from the input relation's partition scheme", QSX_ERR_INVALID_ARGUMENT);
    }
    for (partition_id part = 0; part < num_partitions; ++part) {
      input_.ids[part] = num_partitions > 1 ? input_relation.getBlocksInPartition(part) : input_relation.getBlocksSnapshot();
    }
  }
}

bool AggregationOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                           StorageManager *storage_manager, const tmb::client_id, tmb::MessageBus *) {
  std::lock_guard<std::mutex> lock(mutex_);
  for (partition_id part = 0; part < num_partitions_; ++part) {   // AggregationOperator.cpp:49-61
    AggregationOperationState *state = query_context->getAggregationState(aggr_state_index_, part);
    while (input_.generated[part] < input_.ids[part].size()) {
      // every block that has arrived, in runs of blocks_per_work_order_ (1: the reference's one work order per block)
      const std::size_t take = std::min(blocks_per_work_order_, input_.ids[part].size() - input_.generated[part]);
      if (take > 1) {
        std::vector<block_id> run(input_.ids[part].begin() + static_cast<std::ptrdiff_t>(input_.generated[part]),
                                  input_.ids[part].begin() + static_cast<std::ptrdiff_t>(input_.generated[part] + take));
        container->addNormalWorkOrder(new AggregationWorkOrder(query_id_, std::move(run), state, storage_manager, part,
                                                               CreateLIPFilterAdaptiveProberHelper(lip_deployment_index_, query_context)),
                                      op_index_);
      } else {
        container->addNormalWorkOrder(new AggregationWorkOrder(query_id_, input_.ids[part][input_.generated[part]], state,
                                                               storage_manager, part,
                                                               CreateLIPFilterAdaptiveProberHelper(lip_deployment_index_, query_context)),
                                      op_index_);
      }
      input_.generated[part] += take;
    }
  }
  return input_relation_is_stored_ || done_feeding_input_relation_;
}

BuildAggregationExistenceMapOperator::BuildAggregationExistenceMapOperator(
    std::size_t query_id, const CatalogRelation &input_relation, attribute_id build_attribute, bool input_relation_is_stored,
    QueryContext::aggregation_state_id aggr_state_index, std::size_t num_partitions)
    : RelationalOperator(query_id, num_partitions), input_relation_(input_relation), build_attribute_(build_attribute),
      input_relation_is_stored_(input_relation_is_stored), aggr_state_index_(aggr_state_index), input_(num_partitions) {
  if (input_relation_is_stored) {
    for (partition_id part = 0; part < num_partitions; ++part) {
      input_.ids[part] = num_partitions > 1 ? input_relation.getBlocksInPartition(part) : input_relation.getBlocksSnapshot();
    }
  }
}

bool BuildAggregationExistenceMapOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                                            StorageManager *storage_manager, const tmb::client_id,
                                                            tmb::MessageBus *) {
  std::lock_guard<std::mutex> lock(mutex_);
  for (partition_id part = 0; part < num_partitions_; ++part) {   // BuildAggregationExistenceMapOperator.cpp:82-128
    AggregationOperationState *state = query_context->getAggregationState(aggr_state_index_, part);
    while (input_.generated[part] < input_.ids[part].size()) {
      container->addNormalWorkOrder(new BuildAggregationExistenceMapWorkOrder(query_id_, input_relation_, part,
                                                                              input_.ids[part][input_.generated[part]],
                                                                              build_attribute_, state, storage_manager),
                                    op_index_);
      ++input_.generated[part];
    }
  }
  return input_relation_is_stored_ || done_feeding_input_relation_;
}

bool FinalizeAggregationOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                                   StorageManager *, const tmb::client_id, tmb::MessageBus *) {
  if (!started_) {
    started_ = true;
    InsertDestination *dest = query_context->getInsertDestination(output_destination_index_);
    CheckRepartition("FinalizeAggregationOperator", has_repartition_, dest);
    // num_partitions x aggr_state_num_partitions work orders (FinalizeAggregationOperator.cpp:48-66)
    for (partition_id part = 0; part < num_partitions_; ++part) {
      AggregationOperationState *state = query_context->getAggregationState(aggr_state_index_, part);
      for (std::size_t p = 0; p < aggr_state_num_partitions_; ++p) {
        container->addNormalWorkOrder(new FinalizeAggregationWorkOrder(query_id_, p, aggr_state_num_partitions_, state, dest),
                                      op_index_);
      }
    }
  }
  return true;
}

bool DestroyAggregationStateOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                                       StorageManager *, const tmb::client_id, tmb::MessageBus *) {
  if (!work_generated_) {
    work_generated_ = true;
    for (partition_id part = 0; part < num_partitions_; ++part) {
      container->addNormalWorkOrder(new DestroyAggregationStateWorkOrder(query_id_, aggr_state_index_, query_context, part), op_index_);
    }
  }
  return true;
}

// ---------------------------------------------------------------------------
// ORDER BY
// ---------------------------------------------------------------------------
namespace {
// Sorts the concatenation of `blocks` by `config` and writes the first `limit` tuples (0 = all) into one output block.
void SortBlocksInto(const std::vector<BlockReference> &blocks, const CatalogRelation &relation,
                    const QueryContext::SortConfiguration &config, std::size_t limit, InsertDestination *dest) {
  std::int64_t n = 0;
  for (const BlockReference &b : blocks) n += b->numTuples();
  const std::int64_t out_rows = limit != 0 && static_cast<std::int64_t>(limit) < n ? static_cast<std::int64_t>(limit) : n;
  block_id out_id;
  BlockReference out = dest->getBlockForInsertion(out_rows > 0 ? out_rows : 1, &out_id);
  if (n == 0) {
    dest->returnBlock(out_id, 0);
    return;
  }
  // one contiguous stripe per attribute (a single input block is used in place)
  std::vector<std::unique_ptr<DeviceBuffer>> owned;
  std::vector<const void *> stripes(relation.size(), nullptr);
  for (std::size_t a = 0; a < relation.size(); ++a) {
    const int width = relation.getAttributeType(static_cast<attribute_id>(a)).width;
    if (blocks.size() == 1) {
      stripes[a] = blocks.front()->stripe(static_cast<attribute_id>(a));
      continue;
    }
    owned.emplace_back(new DeviceBuffer(static_cast<std::size_t>(n) * width + 16));
    char *at = static_cast<char *>(owned.back()->ptr);
    for (const BlockReference &b : blocks) {
      const std::size_t bytes = static_cast<std::size_t>(b->numTuples()) * width;
      CheckStatus(qsx_copy_on_device(at, b->stripe(static_cast<attribute_id>(a)), bytes, CurrentStream()), "qsx_copy_on_device");
      at += bytes;
    }
    stripes[a] = owned.back()->ptr;
  }
  std::vector<const void *> key_cols;
  std::vector<std::int32_t> key_types, descending;
  for (std::size_t k = 0; k < config.order_by.size(); ++k) {
    key_cols.push_back(stripes.at(config.order_by[k]));
    key_types.push_back(relation.getAttributeType(config.order_by[k]).id);
    descending.push_back(config.ordering.at(k) ? 0 : 1);
  }
  const std::size_t ws_bytes = qsx_sort_workspace_bytes(n);
  DeviceBuffer ws(ws_bytes), tids(static_cast<std::size_t>(n) * 4 + 16);
  if (out_rows < n) {   // top_k: only the leading rows are wanted
    CheckStatus(qsx_sort_top_k(static_cast<int>(key_cols.size()), key_cols.data(), key_types.data(), descending.data(), n, out_rows,
                               static_cast<std::int32_t *>(tids.ptr), ws.ptr, ws_bytes, CurrentStream()),
                "qsx_sort_top_k");
  } else {
    CheckStatus(qsx_sort_permutation(static_cast<int>(key_cols.size()), key_cols.data(), key_types.data(), descending.data(), n,
                                     static_cast<std::int32_t *>(tids.ptr), ws.ptr, ws_bytes, CurrentStream()),
                "qsx_sort_permutation");
  }
  for (std::size_t a = 0; a < relation.size(); ++a) {
    CheckStatus(qsx_gather(relation.getAttributeType(static_cast<attribute_id>(a)).width, stripes[a],
                           static_cast<const std::int32_t *>(tids.ptr), out_rows, out->stripe(static_cast<attribute_id>(a)),
                           CurrentStream()), "qsx_gather");
  }
  CheckStatus(qsx_stream_synchronize(CurrentStream()), "qsx_stream_synchronize");
  dest->returnBlock(out_id, out_rows);
}

class SortWorkOrder : public WorkOrder {
 public:
  SortWorkOrder(std::size_t query_id, std::vector<block_id> blocks, const CatalogRelation &relation,
                const QueryContext::SortConfiguration &config, std::size_t limit, InsertDestination *dest,
                StorageManager *storage_manager)
      : WorkOrder(query_id), blocks_(std::move(blocks)), relation_(relation), config_(config), limit_(limit), dest_(dest),
        storage_manager_(storage_manager) {}
  void execute() override {   // SortRunGenerationOperator.cpp:88-105 / SortMergeRunOperator.cpp:150-200
    std::vector<BlockReference> refs;
    for (block_id b : blocks_) refs.push_back(storage_manager_->getBlock(b));
    SortBlocksInto(refs, relation_, config_, limit_, dest_);
  }
 private:
  std::vector<block_id> blocks_;
  const CatalogRelation &relation_;
  const QueryContext::SortConfiguration &config_;
  const std::size_t limit_;
  InsertDestination *dest_;
  StorageManager *storage_manager_;
};
}  // namespace

bool SortRunGenerationOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                                 StorageManager *storage_manager, const tmb::client_id, tmb::MessageBus *) {
  const QueryContext::SortConfiguration &config = query_context->getSortConfig(sort_config_index_);
  InsertDestination *dest = query_context->getInsertDestination(output_destination_index_);
  std::lock_guard<std::mutex> lock(mutex_);
  while (num_workorders_generated_ < input_relation_block_ids_.size()) {   // one sorted run per input block
    container->addNormalWorkOrder(new SortWorkOrder(query_id_, {input_relation_block_ids_[num_workorders_generated_]}, input_relation_,
                                                    config, 0, dest, storage_manager), op_index_);
    ++num_workorders_generated_;
  }
  return input_relation_is_stored_ || done_feeding_input_relation_;
}

bool SortMergeRunOperator::getAllWorkOrders(WorkOrdersContainer *container, QueryContext *query_context,
                                            StorageManager *storage_manager, const tmb::client_id, tmb::MessageBus *) {
  std::lock_guard<std::mutex> lock(mutex_);
  if (!input_relation_is_stored_ && !done_feeding_input_relation_) return false;   // every run must have arrived
  if (!work_generated_) {
    work_generated_ = true;
    container->addNormalWorkOrder(new SortWorkOrder(query_id_, input_relation_block_ids_, input_relation_,
                                                    query_context->getSortConfig(sort_config_index_), top_k_,
                                                    query_context->getInsertDestination(output_destination_index_), storage_manager),
                                  op_index_);
  }
  return true;
}

// ---------------------------------------------------------------------------
// QueryPlan / Foreman / Worker
// ---------------------------------------------------------------------------
std::size_t QueryPlan::addRelationalOperator(RelationalOperator *op) {
  operators_.emplace_back(op);
  deps_.emplace_back();
  op->setOperatorIndex(operators_.size() - 1);
  return operators_.size() - 1;
}
void QueryPlan::addDirectDependency(std::size_t consumer, std::size_t producer, bool is_pipeline_breaker) {
  deps_.at(consumer).push_back(Edge{producer, is_pipeline_breaker});
}

ForemanSingleNode::ForemanSingleNode(QueryPlan *plan, QueryContext *query_context, StorageManager *storage_manager,
                                     std::size_t num_workers)
    : plan_(plan), query_context_(query_context), storage_manager_(storage_manager),
      num_workers_(num_workers ? num_workers : 1), outstanding_(plan->size(), 0), executing_(plan->size(), 0) {}

namespace {
// The Worker threads of the process (query_execution/Worker.hpp: created once at start-up, they outlive every query).
// Thread i keeps its HIP stream and with it everything that is cached per (thread, stream): the scratch arenas and staging
// buffers inside libqsx.so, this layer's cache of scratch allocations — a query does not pay for them again (creating
// four threads, streams and their first pinned / device buffers cost ~8 ms per query when the Foreman did it per run()).
// The threads are never joined: they sleep on their queues when the process exits.
class WorkerThreads {
 public:
  // The pool of one device (-1: no usable GPU, every work order fails with QSX_ERR_NO_DEVICE anyway).  One process per GPU is
  // the rule and then there is one pool; a process that drives several GPUs gets Worker threads — and streams — per device.
  static WorkerThreads &instance(int device) {
    static std::mutex pools_mutex;
    static std::map<int, WorkerThreads *> *pools = new std::map<int, WorkerThreads *>;
    std::lock_guard<std::mutex> lock(pools_mutex);
    WorkerThreads *&pool = (*pools)[device];
    if (pool == nullptr) pool = new WorkerThreads(device);
    return *pool;
  }
  // The device that is current in the calling thread: the one the caller's blocks, tables and states live on.
  static int callersDevice() {
    int device = -1;
    if (qsx_device_count() == 0 || qsx_current_device(&device) != QSX_OK) device = -1;
    return device;
  }
  // fn(i) on worker thread i for every i < n; returns when all have returned.
  void run(std::size_t n, const std::function<void(std::size_t)> &fn) {
    std::mutex done_mutex;
    std::condition_variable done_cv;
    std::size_t remaining = n;
    std::vector<Slot *> mine;   // (taken under the lock: another Foreman may be growing slots_ right now)
    {
      std::lock_guard<std::mutex> lock(mutex_);
      while (slots_.size() < n) {
        slots_.emplace_back(new Slot);
        Slot *slot = slots_.back().get();
        std::thread([slot, device = device_]() { threadMain(slot, device); }).detach();
      }
      for (std::size_t i = 0; i < n; ++i) mine.push_back(slots_[i].get());
    }
    for (std::size_t i = 0; i < n; ++i) {
      Slot *slot = mine[i];
      {
        std::lock_guard<std::mutex> lock(slot->mutex);
        slot->tasks.push_back([&, i]() {
          fn(i);
          std::lock_guard<std::mutex> done_lock(done_mutex);
          if (--remaining == 0) done_cv.notify_all();
        });
      }
      slot->cv.notify_one();
    }
    std::unique_lock<std::mutex> lock(done_mutex);
    done_cv.wait(lock, [&] { return remaining == 0; });
  }

 private:
  struct Slot {
    std::mutex mutex;
    std::condition_variable cv;
    std::deque<std::function<void()>> tasks;
  };
  explicit WorkerThreads(int device) : device_(device) {}
  static void threadMain(Slot *slot, int device) {
    qsx_stream_t stream = nullptr;
    // a new thread starts on device 0: move it to the pool's device before anything is created (stream, scratch, staging)
    if (device >= 0 && (qsx_set_current_device(device) != QSX_OK || qsx_stream_create(&stream) != QSX_OK)) {
      stream = nullptr;   // (the default stream then)
    }
    SetCurrentStream(stream);
    for (;;) {
      std::function<void()> task;
      {
        std::unique_lock<std::mutex> lock(slot->mutex);
        slot->cv.wait(lock, [&] { return !slot->tasks.empty(); });
        task = std::move(slot->tasks.front());
        slot->tasks.pop_front();
      }
      task();
    }
  }
  const int device_;
  std::mutex mutex_;
  std::vector<std::unique_ptr<Slot>> slots_;
};
}  // namespace

namespace {
thread_local bool tls_on_worker_thread = false;
}
void ForemanSingleNode::workerMain(std::size_t worker_id) {
  struct OnWorker {
    OnWorker() { tls_on_worker_thread = true; }
    ~OnWorker() { tls_on_worker_thread = false; }
  } on_worker;
  // Worker::run (query_execution/Worker.cpp:54-99): receive a work order, execute(), report completion.
  struct FreshStream {   // QSX_HOST_FRESH_STREAMS (debugging): a stream of this query only on the persistent thread
    qsx_stream_t previous = CurrentStream(), mine = nullptr;
    FreshStream() {
      if (std::getenv("QSX_HOST_FRESH_STREAMS") != nullptr && qsx_device_count() > 0 && qsx_stream_create(&mine) == QSX_OK) SetCurrentStream(mine);
    }
    ~FreshStream() {
      if (mine != nullptr) {
        qsx_stream_synchronize(mine);
        SetCurrentStream(previous);
        qsx_stream_destroy(mine);
      }
    }
  } fresh_stream;
  for (;;) {
    Item item;
    {
      std::unique_lock<std::mutex> lock(mutex_);
      // Work orders that want the device to themselves (WorkOrder::prefersExclusiveDevice) run only next to their own kind:
      // while one is queued nothing else starts, it starts when the others have drained, and the others resume when no
      // such work order is queued or running.  Nothing waits for a work order that is not already on a Worker.
      bool exclusive_queued = false;
      auto runnable = [&](const Item &it) {
        return it.exclusive ? shared_running_ == 0 : (exclusive_running_ == 0 && !exclusive_queued);
      };
      auto first_runnable = [&]() {
        exclusive_queued = false;
        for (const Item &it : ready_) exclusive_queued = exclusive_queued || it.exclusive;
        for (std::size_t i = 0; i < ready_.size(); ++i) {
          if (runnable(ready_[i])) return i;
        }
        return ready_.size();
      };
      std::size_t pick = 0;
      cv_work_.wait(lock, [&] {
        if (shutting_down_) return true;
        pick = first_runnable();
        return pick < ready_.size();
      });
      if (ready_.empty()) break;
      pick = first_runnable();
      if (pick == ready_.size()) break;   // (shutting down with work orders nobody may start: an error elsewhere)
      // The next work order: of the operator with the fewest work orders on Workers right now (first in the queue among
      // equals).  A probe whose build has just finished then starts next to an aggregation that queued seventy work orders
      // before it, instead of behind them: its host-side steps (counts read back, output blocks) overlap the other
      // operator's kernels.  (The reference's PolicyEnforcer picks per query; within one query it is FIFO.)
      for (std::size_t i = pick + 1; i < ready_.size() && executing_[ready_[pick].op] != 0; ++i) {
        if (runnable(ready_[i]) && executing_[ready_[i].op] < executing_[ready_[pick].op]) pick = i;
      }
      item = ready_[pick];
      ready_.erase(ready_.begin() + static_cast<std::ptrdiff_t>(pick));
      ++executing_[item.op];
      ++(item.exclusive ? exclusive_running_ : shared_running_);
    }
    std::unique_ptr<WorkOrder> wo(item.wo);
    const std::uint64_t start = NowMicros();
    std::string error;
    try {
      wo->execute();
    } catch (const std::exception &e) {
      error = e.what();
    }
    wo.reset();
    const std::uint64_t end = NowMicros();
    {
      std::lock_guard<std::mutex> lock(mutex_);
      --outstanding_[item.op];
      --executing_[item.op];
      --(item.exclusive ? exclusive_running_ : shared_running_);
      profile_.push_back(WorkOrderTimeEntry{worker_id, item.op, start, end});
      if (!error.empty() && worker_error_.empty()) worker_error_ = error;
    }
    cv_work_.notify_all();   // (what may start depends on what runs)
    cv_done_.notify_all();
  }
}

void ForemanSingleNode::run() {
  // The process-wide Worker threads run one Foreman's workerMain at a time each: queries admitted concurrently from
  // different threads are served one after the other (the reference's Workers interleave the work orders of admitted
  // queries), and a Foreman started from INSIDE a work order would wait for the very thread it runs on.
  if (tls_on_worker_thread) {
    throw ExecutionError("ForemanSingleNode::run() called from a work order: nested query execution is not supported", QSX_ERR_UNSUPPORTED);
  }
  const std::size_t N = plan_->size();
  // Opt-in: measured on the headline plan (one 100 M-row probe next to 19 aggregation work orders) the probe alone takes
  // 0.95 ms instead of 4.2 ms of wall time next to the aggregation, but the step takes 5.2 ms either way — the device does
  // the same work in both orders and the drain before the probe costs what the undisturbed L2 gains.
  const char *exclusive_env = std::getenv("QSX_HOST_EXCLUSIVE_PROBES");
  const bool exclusive_probes = exclusive_env != nullptr && exclusive_env[0] == '1';
  WorkOrdersContainer container(N);
  std::vector<bool> done_generating(N, false), finished(N, false);
  std::vector<std::size_t> blocks_fed(N, 0);  // per producer: output blocks already fed downstream
  // the process-wide Worker threads serve this query until it shuts them out again (one more thread drives them and waits)
  const int device = WorkerThreads::callersDevice();
  std::thread workers([this, device]() {
    if (std::getenv("QSX_HOST_EPHEMERAL_WORKERS") != nullptr) {   // (debugging: threads and streams of this run() only)
      std::vector<std::thread> own;
      for (std::size_t w = 0; w < num_workers_; ++w) {
        own.emplace_back([this, w, device]() {
          qsx_stream_t stream = nullptr;
          if (device >= 0 && qsx_set_current_device(device) == QSX_OK) (void)qsx_stream_create(&stream);
          SetCurrentStream(stream);
          workerMain(w);
          if (stream != nullptr) qsx_stream_destroy(stream);
        });
      }
      for (auto &t : own) t.join();
      return;
    }
    WorkerThreads::instance(device).run(num_workers_, [this](std::size_t w) { workerMain(w); });
  });

  auto shutdown = [&]() {
    {
      std::lock_guard<std::mutex> lock(mutex_);
      shutting_down_ = true;
    }
    cv_work_.notify_all();
    workers.join();
  };

  try {
    for (;;) {
      std::unique_lock<std::mutex> lock(mutex_);
      if (!worker_error_.empty()) throw std::runtime_error("work order failed: " + worker_error_);
      bool progress = false;
      for (std::size_t op = 0; op < N; ++op) {
        if (finished[op]) continue;
        bool blocked = false, producers_finished = true;
        for (const QueryPlan::Edge &e : plan_->dependencies(op)) {
          if (!finished[e.producer]) {
            producers_finished = false;
            if (e.breaker) blocked = true;
          }
        }
        if (!blocked && !done_generating[op]) {
          // only the Foreman thread ever calls getAllWorkOrders (SURVEY §8b Threading)
          lock.unlock();
          const bool done = plan_->getOperator(op)->getAllWorkOrders(&container, query_context_, storage_manager_, 0, &bus_);
          lock.lock();
          while (WorkOrder *wo = container.getNormalWorkOrder(op)) {
            ready_.push_back(Item{wo, op, exclusive_probes && wo->prefersExclusiveDevice()});
            ++outstanding_[op];
            progress = true;
          }
          if (done) done_generating[op] = true;
        }
        // pipelining: feed newly produced output blocks to streaming consumers (kDataPipelineMessage)
        RelationalOperator *producer = plan_->getOperator(op);
        const QueryContext::insert_destination_id dest_id = producer->getInsertDestinationID();
        if (dest_id != QueryContext::kInvalidInsertDestinationId) {
          const std::vector<InsertDestination::TouchedBlock> touched = query_context_->getInsertDestination(dest_id)->getTouchedBlocksWithPartitions();
          for (; blocks_fed[op] < touched.size(); ++blocks_fed[op]) {
            for (std::size_t consumer = 0; consumer < N; ++consumer) {
              for (const QueryPlan::Edge &e : plan_->dependencies(consumer)) {
                if (e.producer == op && !e.breaker) {
                  // kDataPipelineMessage carries the partition id of the block (InsertDestination.cpp:424-470)
                  plan_->getOperator(consumer)->feedInputBlock(touched[blocks_fed[op]].id, producer->getOutputRelationID(),
                                                               touched[blocks_fed[op]].partition);
                  progress = true;
                }
              }
            }
          }
        }
        if (done_generating[op] && outstanding_[op] == 0 && producers_finished && container.getNumNormalWorkOrders(op) == 0) {
          // re-check that no block appeared between the scan above and now
          if (dest_id == QueryContext::kInvalidInsertDestinationId ||
              blocks_fed[op] == query_context_->getInsertDestination(dest_id)->getTouchedBlocks().size()) {
            finished[op] = true;
            progress = true;
            for (std::size_t consumer = 0; consumer < N; ++consumer) {
              for (const QueryPlan::Edge &e : plan_->dependencies(consumer)) {
                if (e.producer == op && !e.breaker) {
                  plan_->getOperator(consumer)->doneFeedingInputBlocks(producer->getOutputRelationID());
                }
              }
            }
          }
        }
      }
      if (std::all_of(finished.begin(), finished.end(), [](bool f) { return f; })) break;
      if (progress) {
        lock.unlock();
        cv_work_.notify_all();
        continue;
      }
      cv_done_.wait_for(lock, std::chrono::milliseconds(50));
    }
  } catch (...) {
    shutdown();
    throw;
  }
  shutdown();
}

}  // namespace quickstep
