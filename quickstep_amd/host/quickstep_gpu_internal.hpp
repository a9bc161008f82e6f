// quickstep_gpu_internal.hpp — what the host layer's translation units share and the public header does not show:
// the scratch buffer of a work order, the pooled allocations behind it, and the helpers more than one of the files uses.
// (host_runtime.cpp: errors, streams, pools; storage.cpp: catalog, blocks, compression, block images; query_context.cpp:
// predicates, destinations, aggregation state, LIP filters; select_build_operators.cpp, hash_join_operator.cpp,
// aggregation_sort_operators.cpp: the operators and their work orders; foreman.cpp: plan, Foreman, Worker threads.)
#ifndef QUICKSTEP_GPU_INTERNAL_HPP_
#define QUICKSTEP_GPU_INTERNAL_HPP_

#include "quickstep_gpu.hpp"

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>

namespace quickstep {
namespace host_internal {

// Scratch owned by one work order.  A device allocation of a few MB costs ~170 us (hipMalloc + hipFree, whatever the
// size; 0.7 us up to 64 KiB: tools/ubench/alloc_cost.hip) — more than the kernels of a work order over a 4 MB block take —
// and the stream-ordered pool is not an option on this stack (csrc/common.hpp, CallScratch).  So buffers above 64 KiB come
// from a per-thread cache of plain allocations in power-of-two size classes: a worker thread issues all its work on one
// stream, so a buffer handed back while its last kernel is still queued can be handed out again to the same thread —
// the next use is ordered behind it.  (A buffer built by one thread and consumed by another — DISTINCT chunks — is
// published after a stream synchronisation, like a storage block.)  The cache keeps at most kCacheBytes per thread.
void TrimBlockSlabPool();   // (host_runtime.cpp, BlockSlabPool)
void *TakePooled(std::size_t bytes, std::size_t *granted);
void GivePooled(void *p, std::size_t granted);
struct DeviceBuffer {
  // Every scratch buffer is cached per thread, the 8-byte count words of a work order included: a plain allocation is a
  // hipMalloc + hipFree pair, and hipFree waits for everything queued on the device — with eight Workers in flight a step of
  // the operators' bench stalled 6-9 ms on one of them every few steps (366 hipFree calls in 28 steps, 0.65 ms on average,
  // 9 ms at worst: rocprofv3 --hip-trace).  (Until round 5 only buffers above 64 KiB were cached.)
  static constexpr std::size_t kCacheFrom = 1;
  static constexpr std::size_t kCacheBytes = std::size_t(2) << 30;
  // (device, size class): a thread that moves to another device (qsx_set_current_device) must not be handed the other
  // device's memory — Worker threads stay on one device, callers of the layer need not
  static std::pair<int, std::size_t> classKey(std::size_t size_class) {
    int device = 0;
    (void)qsx_current_device(&device);
    return std::make_pair(device, size_class);
  }
  struct Cache {
    std::map<std::pair<int, std::size_t>, std::vector<void *>> free_by_class;
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
  std::pair<int, std::size_t> class_key{0, 0};   // (device the buffer was allocated on, size class): where a release files it
  std::size_t size_class = 0;   // 0: a plain allocation of its own
  std::size_t pooled = 0;       // != 0: from the shared pool (its size class there)
  // (1 MiB, 32 MiB until late in round 5: the per-block bitmaps of a run — two sets of 16 MiB — were cached per thread, and every
  // Worker paid ~7 ms of first allocations the first time a work order of that kind came its way, thirty steps into a run at times)
  static constexpr std::size_t kSharedFrom = std::size_t(1) << 20;
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
      size_class = bytes <= 64 * 1024 ? 256 : 128 * 1024;   // (small ones in classes of their own: 256 B, 512 B, ...)
      while (size_class < bytes) size_class *= 2;
      Cache &c = cache();
      // (the key is taken once, here: a thread that changes its device between constructor and destructor must still file
      // the buffer under the device that owns the memory)
      class_key = classKey(size_class);
      auto it = c.free_by_class.find(class_key);
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
        c.free_by_class[class_key].push_back(ptr);
        c.bytes += size_class;
        return;
      }
    }
    qsx_device_free(ptr);
  }
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
};

std::int64_t ReadCount(const void *dev_count);
std::uint64_t NowMicros();

// NULL handling shared by the aggregation state and the operators (storage.cpp)
std::unique_ptr<DeviceBuffer> NotNullFilter(const StorageBlock &block, const std::vector<attribute_id> &attrs, const std::uint64_t *filter);
void GatherBlockNulls(const StorageBlock &block, attribute_id attr, const void *tids, std::int64_t n, std::uint64_t *dst);
void ProjectNullBitmaps(const StorageBlock &block, const std::vector<attribute_id> &selection, const void *bitmap,
                        std::int64_t num_selected, StorageBlock *out);

// ---- the join key of a block / of a run of blocks as the join table sees it (BuildHash and HashJoin work orders) ----
// The single key column the join table sees for one block: the attribute's stripe, or — for a
// composite key — the LONG fold of the components (qsx_join_key_pack; the QueryContext creates
// the table of a composite key with key type kLong).  `exact` is false when the fold is the
// reference's composite hash and joined pairs still need their components compared.
// A CHAR(n <= 8) key attribute of `block` as a LONG stripe (qsx_join_key_pack_char); wider strings are not join keys here.
inline std::unique_ptr<DeviceBuffer> CharKeyAsLong(const StorageBlock &block, attribute_id a) {
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

// The key stripes of a run of blocks as the join table sees them: the attribute's stripes, or — composite key — one stripe of
// packed keys for the whole run (qsx_join_key_pack_blocks), block b's keys at row first_rows[b] of it.
//
// A single INT / LONG key attribute that a block holds COMPRESSED (CompressedColumnStore: truncated values or dictionary codes,
// each block its own choice) is presented as it lies — ptr[b] = the code stripe, code_width[b] / dictionary[b] its coding — for
// the *_blocks_coded entry points (include/qsx.h qsx_key_coding_t): the kernels widen or look up the codes themselves, the
// attribute is never decoded into a stripe of values (StorageBlock::stripe would on first use).  coding() is nullptr when no
// block of the run is compressed in its key.
struct RunJoinKeys {
  std::vector<const void *> ptr;          // per block
  std::vector<std::int32_t> code_width;   // per block: 0 = values
  std::vector<const void *> dictionary;   // per block: nullptr = values / truncated values
  bool coded = false;
  bool exact = true;
  std::unique_ptr<DeviceBuffer> packed;
  std::vector<std::unique_ptr<DeviceBuffer>> char_keys;
  qsx_key_coding_t coding_{nullptr, nullptr};
  const qsx_key_coding_t *coding() const { return coded ? &coding_ : nullptr; }
  RunJoinKeys(const std::vector<BlockReference> &blocks, const std::vector<attribute_id> &attrs, const std::vector<std::int64_t> &rows) {
    const CatalogRelation &relation = blocks.front()->getRelation();
    if (attrs.size() == 1) {
      const TypeID key_type = relation.getAttributeType(attrs.front()).id;
      for (const BlockReference &b : blocks) {
        const CompressedAttribute *c = (key_type == kInt || key_type == kLong) ? b->compressedAttribute(attrs.front()) : nullptr;
        if (key_type == kChar) {
          char_keys.push_back(CharKeyAsLong(*b, attrs.front()));
          ptr.push_back(char_keys.back()->ptr);
        } else if (c != nullptr && b->numTuples() > 0) {
          ptr.push_back(c->codes);
          coded = true;
        } else {
          ptr.push_back(b->stripe(attrs.front()));
          c = nullptr;
        }
        code_width.push_back(c != nullptr && key_type != kChar ? c->code_width : 0);
        dictionary.push_back(c != nullptr && key_type != kChar && c->kind == CompressedAttribute::kDictionary ? c->dictionary : nullptr);
      }
      coding_.block_code_width = code_width.data();
      coding_.block_dictionaries = dictionary.data();
      return;
    }
    std::vector<const void *> cols;
    std::vector<std::int32_t> types;
    for (attribute_id a : attrs) types.push_back(relation.getAttributeType(a).id == kChar ? static_cast<std::int32_t>(kLong) : relation.getAttributeType(a).id);
    // (a component that a block holds compressed is packed from its code stripe: qsx_join_key_pack_blocks_coded)
    std::vector<std::int32_t> widths;
    std::vector<const void *> dicts;
    std::int64_t total = 0;
    for (std::size_t b = 0; b < blocks.size(); ++b) {
      for (attribute_id a : attrs) {
        const TypeID id = relation.getAttributeType(a).id;
        const CompressedAttribute *c = (id == kInt || id == kLong) && blocks[b]->numTuples() > 0 ? blocks[b]->compressedAttribute(a) : nullptr;
        if (id == kChar) {
          char_keys.push_back(CharKeyAsLong(*blocks[b], a));
          cols.push_back(char_keys.back()->ptr);
        } else if (c != nullptr) {
          cols.push_back(c->codes);
        } else {
          cols.push_back(blocks[b]->stripe(a));
        }
        widths.push_back(c != nullptr ? c->code_width : 0);
        dicts.push_back(c != nullptr && c->kind == CompressedAttribute::kDictionary ? c->dictionary : nullptr);
      }
      total += rows[b];
    }
    packed.reset(new DeviceBuffer(static_cast<std::size_t>(total) * 8 + 8));
    int is_exact = 0;
    CheckStatus(qsx_join_key_pack_blocks_coded(static_cast<int>(attrs.size()), types.data(), static_cast<std::int64_t>(blocks.size()), rows.data(),
                                               cols.data(), widths.data(), dicts.data(), static_cast<std::int64_t *>(packed->ptr), &is_exact,
                                               CurrentStream()),
                "qsx_join_key_pack_blocks");
    exact = is_exact != 0;
    std::int64_t at = 0;
    for (std::size_t b = 0; b < blocks.size(); ++b) {
      ptr.push_back(static_cast<const std::int64_t *>(packed->ptr) + at);
      at += rows[b];
    }
  }
};

}  // namespace host_internal
using namespace host_internal;   // (this header is included by the layer's own sources only)

void CheckRepartition(const char *op, bool has_repartition, const InsertDestination *dest);   // (query_context.cpp)

// A predicate over a run of blocks (query_context.cpp): per-block TupleIdSequences in two device allocations.
struct RunMatches {
  std::unique_ptr<DeviceBuffer> set_a, set_b, counts;
  std::vector<std::uint64_t *> bitmaps;   // block b's bitmap under the whole conjunction
};
bool RunPredicateCovers(const Predicate &predicate, const std::vector<BlockReference> &blocks);
void RunPredicateMatches(const Predicate &predicate, const std::vector<BlockReference> &blocks, const std::vector<std::int64_t> &rows,
                         const std::uint64_t *const *in_filters, RunMatches *out);

}  // namespace quickstep

#endif  // QUICKSTEP_GPU_INTERNAL_HPP_
