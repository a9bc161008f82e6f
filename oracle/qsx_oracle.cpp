// qsx_oracle.cpp — CPU ORACLE (test infrastructure; see qsx_oracle.h header
// comment for what it is pinned against and who may call it).
//
// Every function is a restatement of the reference algorithm it cites
// (paths relative to the Quickstep tree); nothing here is shared with the HIP
// product path.  Build: see oracle/Makefile (g++ -O2 -ffp-contract=off).

#include "qsx_oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <unordered_map>
#include <utility>
#include <set>
#include <vector>

namespace {

constexpr std::size_t kSlotSizeBytes = 0x200000;  // storage/StorageConstants.hpp:50
constexpr std::uint64_t kHashTableLoadFactor = 2;   // slots per entry (storage/StorageConstants.hpp:104)
constexpr std::uint64_t kSizeMax = ~static_cast<std::uint64_t>(0);

// ---------------------------------------------------------------------------
// BitVector<false> / TupleIdSequence bit order (utility/BitVector.hpp:893-935):
// bit i lives in word i>>6 under mask (1<<63) >> (i & 63).
// ---------------------------------------------------------------------------
inline bool bit_get(const std::uint64_t *w, std::int64_t i) {
  return (w[i >> 6] >> (63 - (i & 63))) & 1u;
}
inline void bit_set(std::uint64_t *w, std::int64_t i) {
  w[i >> 6] |= (static_cast<std::uint64_t>(1) << 63) >> (i & 63);
}
inline std::int64_t bitmap_words(std::int64_t n) { return (n + 63) >> 6; }

inline bool row_selected(const std::uint64_t *filter, std::int64_t i) {
  return filter == nullptr || bit_get(filter, i);
}

// ---------------------------------------------------------------------------
// Hashing (types/TypedValue.hpp:95-129 constructors, :575-592, :916-918).
// ---------------------------------------------------------------------------
inline std::uint64_t hash_int(std::int32_t v) { return static_cast<std::uint32_t>(v); }
inline std::uint64_t hash_long(std::int64_t v) { return static_cast<std::uint64_t>(v); }
inline std::uint64_t hash_float(float v) {
  if (v == 0.0f) v = 0.0f;  // -0.0f canonicalised (TypedValue.hpp:113-120)
  std::uint32_t bits;
  std::memcpy(&bits, &v, 4);
  return bits;
}
inline std::uint64_t hash_double(double v) {
  if (v == 0.0) v = 0.0;  // TypedValue.hpp:125-129
  std::uint64_t bits;
  std::memcpy(&bits, &v, 8);
  return bits;
}

inline std::uint64_t hash_typed(int type, const void *p) {
  switch (type) {
    case QSX_INT: { std::int32_t v; std::memcpy(&v, p, 4); return hash_int(v); }
    case QSX_LONG: { std::int64_t v; std::memcpy(&v, p, 8); return hash_long(v); }
    case QSX_FLOAT: { float v; std::memcpy(&v, p, 4); return hash_float(v); }
    case QSX_DOUBLE: { double v; std::memcpy(&v, p, 8); return hash_double(v); }
    // a DATE hashes as the 8 bytes of its TypedValue (getHashScalarLiteral, TypedValue.hpp:575-592, 916-918); the padding
    // bytes of the DateLit are taken as zero
    case QSX_DATE: { std::uint64_t v; std::memcpy(&v, p, 8); return v & 0x0000FFFFFFFFFFFFull; }
    default: std::abort();
  }
}

inline int type_width(int type) {
  switch (type) {
    case QSX_INT: case QSX_FLOAT: return 4;
    case QSX_LONG: case QSX_DOUBLE: case QSX_DATE: return 8;
    default: return 0;
  }
}

// utility/HashPair.hpp:47-58 (64-bit specialisation).
inline std::uint64_t combine_hashes(std::uint64_t first, std::uint64_t second) {
  const std::uint64_t kMul = 0x9ddfea08eb382d69ULL;
  std::uint64_t a = (first ^ second) * kMul;
  a ^= (a >> 47);
  std::uint64_t b = (second ^ a) * kMul;
  b ^= (b >> 47);
  b *= kMul;
  return b;
}

bool is_prime(std::uint64_t n) {
  if (n < 2) return false;
  if (n < 4) return true;
  if ((n & 1) == 0) return false;
  for (std::uint64_t d = 3; d * d <= n; d += 2) {
    if (n % d == 0) return false;
  }
  return true;
}
// utility/PrimeNumber.hpp:38 — least prime >= n.
std::uint64_t next_prime(std::uint64_t n) {
  if (n <= 2) return 2;
  while (!is_prime(n)) ++n;
  return n;
}
// utility/PrimeNumber.hpp:49 — greatest prime <= n, 0 if none.
std::uint64_t prev_prime(std::uint64_t n) {
  if (n < 2) return 0;
  while (!is_prime(n)) --n;
  return n;
}

// catalog/PartitionSchemeHeader.hpp:200-214.
inline std::uint64_t partition_id(std::uint64_t h, std::uint64_t P) {
  if ((P & (P - 1)) == 0) return h & (P - 1);
  return h >= P ? h % P : h;
}

// ---------------------------------------------------------------------------
// Comparison (types/operations/comparisons/LiteralComparators-inl.hpp:317-388).
// ---------------------------------------------------------------------------
template <typename T>
inline bool compare(T a, int op, T b) {
  switch (op) {
    case QSX_EQ: return a == b;
    case QSX_NE: return a != b;
    case QSX_LT: return a < b;
    case QSX_LE: return a <= b;
    case QSX_GT: return a > b;
    case QSX_GE: return a >= b;
    default: return false;
  }
}

// DateLit (types/DatetimeLit.hpp:38-90): {int32 year; uint8 month, day} in 8 bytes, ordered by year, then month, then
// day; the two padding bytes take no part in anything.
struct DateLit {
  std::int32_t year;
  std::uint8_t month, day;
  std::uint8_t padding[2];
  bool operator<(const DateLit &r) const { return year != r.year ? year < r.year : (month != r.month ? month < r.month : day < r.day); }
  bool operator>(const DateLit &r) const { return r < *this; }
  bool operator<=(const DateLit &r) const { return !(*this > r); }
  bool operator>=(const DateLit &r) const { return !(*this < r); }
  bool operator==(const DateLit &r) const { return year == r.year && month == r.month && day == r.day; }
  bool operator!=(const DateLit &r) const { return !(*this == r); }
};
static_assert(sizeof(DateLit) == 8, "DateLit occupies 8 bytes of a column stripe");
inline DateLit date_at(const void *col, std::int64_t i) {
  DateLit d;
  std::memcpy(&d, static_cast<const char *>(col) + i * 8, 8);
  return d;
}
constexpr std::uint64_t kDateValueMask = 0x0000FFFFFFFFFFFFull;

// AsciiStringUncheckedComparator::strcmpHelper (types/operations/comparisons/AsciiStringComparators.hpp:218-251) for a
// CHAR(left_length) value against a literal of right_length bytes, neither necessarily NUL-terminated.
inline int strcmp_helper(const char *left, std::size_t left_length, const char *right, std::size_t right_length) {
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

template <typename T>
void select_cmp_t(const T *col, std::int64_t n, int op, T lit, const std::uint64_t *filter,
                  std::uint64_t *out) {
  std::memset(out, 0, sizeof(std::uint64_t) * bitmap_words(n));
  if (filter != nullptr) {
    // short-circuit path (:344-356): evaluate only the rows of `filter`.
    for (std::int64_t i = 0; i < n; ++i) {
      if (bit_get(filter, i) && compare<T>(col[i], op, lit)) bit_set(out, i);
    }
  } else {
    // :358-371 — per-row result->set(pos, cmp).
    for (std::int64_t i = 0; i < n; ++i) {
      if (compare<T>(col[i], op, lit)) bit_set(out, i);
    }
  }
}

}  // namespace

// ===========================================================================
// join hash table
// ===========================================================================
namespace {

// storage/TupleReference.hpp:36-47
struct TupleRef {
  std::uint64_t block;
  std::int32_t tuple;
};
// storage/SimpleScalarSeparateChainingHashTable.hpp:214-218 (32 bytes)
struct Bucket {
  std::atomic<std::uint64_t> next;
  std::uint64_t hash;
  TupleRef value;
};
static_assert(sizeof(Bucket) == 32, "bucket layout");
constexpr std::size_t kHeaderBytes = 128;  // Header with a cache-line aligned atomic (:207-212)

struct Storage {
  std::uint64_t num_slots = 0;
  std::uint64_t num_buckets = 0;
  std::uint64_t blob_bytes = 0;
  std::unique_ptr<std::atomic<std::uint64_t>[]> slots;
  std::unique_ptr<Bucket[]> buckets;
};

// Sizing rule of the constructor (:283-397) and of resize (:838-895): take a
// prime slot count, round the footprint up to whole 2 MiB storage slots, then
// refit slots/buckets to the blob actually obtained.
void size_storage(std::uint64_t wanted_slots, Storage *st) {
  const std::uint64_t required = kHeaderBytes + wanted_slots * 8 + (wanted_slots * 32) / 2;
  const std::uint64_t blob_slots = (required + kSlotSizeBytes - 1) / kSlotSizeBytes;
  const std::uint64_t available = blob_slots * kSlotSizeBytes - kHeaderBytes;
  std::uint64_t buckets = available / (2 * 8 + 32);
  const std::uint64_t slots = prev_prime(buckets * 2);
  buckets = slots / 2;
  st->num_slots = slots;
  st->num_buckets = buckets;
  st->blob_bytes = blob_slots * kSlotSizeBytes;
  st->slots.reset(new std::atomic<std::uint64_t>[slots]);
  for (std::uint64_t i = 0; i < slots; ++i) st->slots[i].store(0, std::memory_order_relaxed);
  st->buckets.reset(new Bucket[buckets]);
  for (std::uint64_t i = 0; i < buckets; ++i) {
    st->buckets[i].next.store(0, std::memory_order_relaxed);
    st->buckets[i].hash = 0;
    st->buckets[i].value = TupleRef{0, -1};
  }
}

}  // namespace

struct qso_join_table {
  int key_type;
  Storage st;
  std::atomic<std::uint64_t> buckets_allocated{0};
  std::shared_mutex resize_mutex;  // HashTable::resize_shared_mutex_ (storage/HashTable.hpp:1215)

  inline std::uint64_t key_hash(const void *keys, std::int64_t i) const {
    if (key_type == QSX_INT) return hash_int(static_cast<const std::int32_t *>(keys)[i]);
    return hash_long(static_cast<const std::int64_t *>(keys)[i]);
  }

  // resize() (:820-985): grow to ~2x, copy the buckets, rebuild chains by
  // pushing each bucket at the HEAD of its slot's chain (:968-981).
  void resize(std::uint64_t extra_buckets) {
    std::unique_lock<std::shared_mutex> lock(resize_mutex);
    if (buckets_allocated.load() + extra_buckets < st.num_buckets) return;  // isFull (:241-244)
    Storage bigger;
    size_storage(next_prime((st.num_buckets + extra_buckets / 2) * 2 * 2), &bigger);
    const std::uint64_t used = buckets_allocated.load();
    for (std::uint64_t b = 0; b < used; ++b) {
      bigger.buckets[b].hash = st.buckets[b].hash;
      bigger.buckets[b].value = st.buckets[b].value;
    }
    st = std::move(bigger);
    for (std::uint64_t b = 0; b < used; ++b) {
      Bucket &bucket = st.buckets[b];
      const std::uint64_t slot = bucket.hash % st.num_slots;
      const std::uint64_t head = st.slots[slot].load(std::memory_order_relaxed);
      bucket.next.store(head, std::memory_order_relaxed);  // 0 when the slot was empty
      st.slots[slot].store(b + 1, std::memory_order_relaxed);
    }
  }

  // preallocateForBulkInsert (:1003-1024).
  bool preallocate(std::uint64_t total, std::uint64_t *position) {
    std::uint64_t original = buckets_allocated.load(std::memory_order_relaxed);
    std::uint64_t post = original + total;
    while (post <= st.num_buckets &&
           !buckets_allocated.compare_exchange_weak(original, post, std::memory_order_relaxed)) {
      post = original + total;
    }
    if (post > st.num_buckets) return false;
    *position = original;
    return true;
  }

  // putInternal + locateBucketForInsertion with a preallocation state
  // (:1062-1113): walk to the chain tail, CAS 0 -> pending, then publish.
  void put_prealloc(std::uint64_t hash, const TupleRef &value, std::uint64_t *position) {
    std::atomic<std::uint64_t> *pending = &st.slots[hash % st.num_slots];
    for (;;) {
      std::uint64_t existing = 0;
      if (pending->compare_exchange_strong(existing, kSizeMax, std::memory_order_acq_rel)) {
        const std::uint64_t bucket_num = (*position)++;
        Bucket &bucket = st.buckets[bucket_num];
        bucket.hash = hash;
        bucket.value = value;
        pending->store(bucket_num + 1, std::memory_order_release);
        return;
      }
      while (existing == kSizeMax) existing = pending->load(std::memory_order_acquire);
      if (existing == 0) continue;
      pending = &st.buckets[existing - 1].next;
    }
  }

  // putValueAccessor (storage/HashTable.hpp:1358-1461) for one block, with
  // TupleReferenceGenerator values; NULL keys do not exist (non-nullable).
  void put_block(const void *keys, std::int64_t n, std::uint64_t block_id, std::int32_t base_tid,
                 const std::uint64_t *filter) {
    std::uint64_t total = 0;
    if (filter == nullptr) {
      total = static_cast<std::uint64_t>(n);
    } else {
      for (std::int64_t i = 0; i < n; ++i) total += bit_get(filter, i);
    }
    std::uint64_t position = 0;
    for (;;) {
      bool ok;
      {
        std::shared_lock<std::shared_mutex> lock(resize_mutex);
        ok = preallocate(total, &position);
      }
      if (ok) break;
      resize(total);
    }
    std::shared_lock<std::shared_mutex> lock(resize_mutex);
    for (std::int64_t i = 0; i < n; ++i) {
      if (!row_selected(filter, i)) continue;
      put_prealloc(key_hash(keys, i), TupleRef{block_id, static_cast<std::int32_t>(base_tid + i)},
                   &position);
    }
  }
};

namespace {
// EvaluatePredicateForUncompressedSortColumn (storage/ColumnStoreUtil.cpp:40-280): std::lower_bound / std::upper_bound on
// the sorted stripe give [min_match, max_match_bound); != takes the complement; the filter is intersected afterwards
// (BasicColumnStoreTupleStorageSubBlock.cpp:603-606).
template <typename T>
void select_cmp_sorted_t(const T *col, std::int64_t n, int op, T lit, const std::uint64_t *filter, std::uint64_t *out) {
  const std::int64_t lower = std::lower_bound(col, col + n, lit) - col;
  const std::int64_t upper = std::upper_bound(col, col + n, lit) - col;
  std::int64_t min_match = 0, max_match_bound = n;
  switch (op) {
    case QSX_EQ: case QSX_NE: min_match = lower; max_match_bound = upper; break;
    case QSX_LT: max_match_bound = lower; break;
    case QSX_LE: max_match_bound = upper; break;
    case QSX_GT: min_match = upper; break;
    default: min_match = lower; break;
  }
  for (std::int64_t w = 0; w < bitmap_words(n); ++w) out[w] = 0;
  for (std::int64_t i = 0; i < n; ++i) {
    bool match = i >= min_match && i < max_match_bound;
    if (op == QSX_NE) match = !match;
    if (match && row_selected(filter, i)) out[i >> 6] |= (static_cast<std::uint64_t>(1) << 63) >> (i & 63);
  }
}
}  // namespace


extern "C" {

size_t qso_sizeof_agg_config(void) { return sizeof(qsx_agg_config_t); }
uint64_t qso_hash_scalar(int type, const void *value) { return hash_typed(type, value); }
uint64_t qso_combine_hashes(uint64_t a, uint64_t b) { return combine_hashes(a, b); }
uint64_t qso_next_prime(uint64_t n) { return next_prime(n); }
uint64_t qso_prev_prime(uint64_t n) { return prev_prime(n); }
uint64_t qso_partition_id(uint64_t hash, uint64_t num_partitions) {
  return partition_id(hash, num_partitions);
}

void qso_select_cmp(int type, const void *col, int64_t n, int op, const void *literal,
                    const uint64_t *filter, uint64_t *out_bitmap) {
  switch (type) {
    case QSX_INT: {
      std::int32_t lit; std::memcpy(&lit, literal, 4);
      select_cmp_t<std::int32_t>(static_cast<const std::int32_t *>(col), n, op, lit, filter, out_bitmap);
      break;
    }
    case QSX_LONG: {
      std::int64_t lit; std::memcpy(&lit, literal, 8);
      select_cmp_t<std::int64_t>(static_cast<const std::int64_t *>(col), n, op, lit, filter, out_bitmap);
      break;
    }
    case QSX_FLOAT: {
      float lit; std::memcpy(&lit, literal, 4);
      select_cmp_t<float>(static_cast<const float *>(col), n, op, lit, filter, out_bitmap);
      break;
    }
    case QSX_DOUBLE: {
      double lit; std::memcpy(&lit, literal, 8);
      select_cmp_t<double>(static_cast<const double *>(col), n, op, lit, filter, out_bitmap);
      break;
    }
    case QSX_DATE: {
      DateLit lit; std::memcpy(&lit, literal, 8);
      select_cmp_t<DateLit>(static_cast<const DateLit *>(col), n, op, lit, filter, out_bitmap);
      break;
    }
    default: std::abort();
  }
}

void qso_select_cmp_char(const void *col, int width, int64_t n, int op, const void *literal, int literal_length,
                         const uint64_t *filter, uint64_t *out_bitmap) {
  std::memset(out_bitmap, 0, sizeof(uint64_t) * bitmap_words(n));
  for (int64_t i = 0; i < n; ++i) {
    if (filter != nullptr && !bit_get(filter, i)) continue;
    const int res = strcmp_helper(static_cast<const char *>(col) + i * width, static_cast<std::size_t>(width),
                                  static_cast<const char *>(literal), static_cast<std::size_t>(literal_length));
    if (compare<int>(res, op, 0)) bit_set(out_bitmap, i);
  }
}

void qso_select_cmp_sorted(int type, const void *col, int64_t n, int op, const void *literal, const uint64_t *filter,
                           uint64_t *out_bitmap) {
  switch (type) {
    case QSX_INT: { std::int32_t lit; std::memcpy(&lit, literal, 4); select_cmp_sorted_t(static_cast<const std::int32_t *>(col), n, op, lit, filter, out_bitmap); break; }
    case QSX_LONG: { std::int64_t lit; std::memcpy(&lit, literal, 8); select_cmp_sorted_t(static_cast<const std::int64_t *>(col), n, op, lit, filter, out_bitmap); break; }
    case QSX_FLOAT: { float lit; std::memcpy(&lit, literal, 4); select_cmp_sorted_t(static_cast<const float *>(col), n, op, lit, filter, out_bitmap); break; }
    case QSX_DOUBLE: { double lit; std::memcpy(&lit, literal, 8); select_cmp_sorted_t(static_cast<const double *>(col), n, op, lit, filter, out_bitmap); break; }
    case QSX_DATE: { DateLit lit; std::memcpy(&lit, literal, 8); select_cmp_sorted_t(static_cast<const DateLit *>(col), n, op, lit, filter, out_bitmap); break; }
    default: std::abort();
  }
}

int64_t qso_bitmap_count(const uint64_t *bitmap, int64_t n) {
  int64_t c = 0;
  for (int64_t w = 0; w < bitmap_words(n); ++w) c += __builtin_popcountll(bitmap[w]);
  return c;
}

int64_t qso_compact_gather(int width, const void *src, const uint64_t *bitmap, int64_t n, void *dst) {
  // bulkInsertTuplesWithRemappedAttributes: per selected row, memcpy(width)
  // (storage/BasicColumnStoreTupleStorageSubBlock.cpp:339-425).
  const char *s = static_cast<const char *>(src);
  char *d = static_cast<char *>(dst);
  int64_t out = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (bit_get(bitmap, i)) {
      std::memcpy(d + out * width, s + i * width, width);
      ++out;
    }
  }
  return out;
}

int64_t qso_bitmap_to_tids(const uint64_t *bitmap, int64_t n, int32_t base_tid, int32_t *out) {
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (bit_get(bitmap, i)) out[c++] = static_cast<int32_t>(base_tid + i);
  }
  return c;
}

void qso_gather(int width, const void *src, const int32_t *tids, int64_t n, void *dst) {
  // ScalarAttribute::getAllValuesForJoin (expressions/scalar/ScalarAttribute.cpp:185-225).
  const char *s = static_cast<const char *>(src);
  char *d = static_cast<char *>(dst);
  for (int64_t i = 0; i < n; ++i) {
    if (tids[i] < 0) {
      std::memset(d + i * width, 0, width);
    } else {
      std::memcpy(d + i * width, s + static_cast<int64_t>(tids[i]) * width, width);
    }
  }
}

qso_join_table_t *qso_join_table_create(int key_type, int64_t est_entries) {
  if (key_type != QSX_INT && key_type != QSX_LONG) return nullptr;
  qso_join_table_t *t = new qso_join_table_t();
  t->key_type = key_type;
  // constructor (:283-397): num_slots_tmp = next_prime(num_entries * kHashTableLoadFactor)
  size_storage(next_prime(static_cast<uint64_t>(est_entries < 0 ? 0 : est_entries) * 2), &t->st);
  return t;
}

void qso_join_table_destroy(qso_join_table_t *t) { delete t; }

void qso_join_table_info(const qso_join_table_t *t, uint64_t out[4]) {
  out[0] = t->st.num_slots;
  out[1] = t->st.num_buckets;
  out[2] = t->buckets_allocated.load();
  out[3] = t->st.blob_bytes;
}

void qso_join_build(qso_join_table_t *t, const void *keys, int64_t n, uint64_t block_id,
                    int32_t base_tid, const uint64_t *filter) {
  t->put_block(keys, n, block_id, base_tid, filter);
}

int64_t qso_join_probe(const qso_join_table_t *t, const void *keys, int64_t n,
                       int32_t probe_base_tid, const uint64_t *filter, int32_t *out_probe_tid,
                       int32_t *out_build_tid, uint64_t *out_build_block, int64_t capacity) {
  // getAllFromValueAccessorImpl (storage/HashTable.hpp:2145-2181) with
  // getNextEntryForKey (SimpleScalarSeparateChainingHashTable.hpp:751-781).
  int64_t count = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (!row_selected(filter, i)) continue;
    const uint64_t hash = t->key_hash(keys, i);
    uint64_t entry = t->st.slots[hash % t->st.num_slots].load(std::memory_order_relaxed);
    while (entry != 0) {
      const Bucket &bucket = t->st.buckets[entry - 1];
      entry = bucket.next.load(std::memory_order_relaxed);
      if (bucket.hash == hash) {
        if (count < capacity) {
          out_probe_tid[count] = static_cast<int32_t>(probe_base_tid + i);
          out_build_tid[count] = bucket.value.tuple;
          if (out_build_block != nullptr) out_build_block[count] = bucket.value.block;
        }
        ++count;
      }
    }
  }
  return count;
}

void qso_join_probe_exists(const qso_join_table_t *t, const void *keys, int64_t n,
                           const uint64_t *filter, int anti, uint64_t *out_bitmap) {
  // runOverKeysFromValueAccessorIfMatch[Not]Found (storage/HashTable.hpp:1979-2062).
  std::memset(out_bitmap, 0, sizeof(uint64_t) * bitmap_words(n));
  for (int64_t i = 0; i < n; ++i) {
    if (!row_selected(filter, i)) continue;
    const uint64_t hash = t->key_hash(keys, i);
    uint64_t entry = t->st.slots[hash % t->st.num_slots].load(std::memory_order_relaxed);
    bool found = false;
    while (entry != 0 && !found) {
      const Bucket &bucket = t->st.buckets[entry - 1];
      entry = bucket.next.load(std::memory_order_relaxed);
      found = (bucket.hash == hash);
    }
    if (found != (anti != 0)) bit_set(out_bitmap, i);
  }
}

}  // extern "C"

// ---------------------------------------------------------------------------
// composite-key join table: SeparateChainingHashTable restated for fixed-width
// components.  Bucket = {next, hash, value, key components} (storage/
// SeparateChainingHashTable.hpp:166, kValueOffset/key layout via HashTableKeyManager);
// hash = CombineHashes fold over the component hashes (storage/HashTable.hpp:2109-2119);
// put appends at the chain tail (putCompositeKeyInternal -> locateBucketForInsertion,
// .hpp:743-792, 1339-1374); lookup compares hash AND components
// (getNextEntryForCompositeKey, .hpp:1033-1060).  Single-threaded; growth re-chains like
// resize (.hpp:1080-1240) but is not restated step by step (chain order never shows in a result).
// ---------------------------------------------------------------------------
struct qso_cjoin_table {
  int nkeys = 0;
  int key_types[QSX_MAX_KEYS];
  std::uint64_t num_slots = 0;
  std::vector<std::uint64_t> slots;            // 0 = empty, else bucket + 1
  struct CBucket {
    std::uint64_t next;
    std::uint64_t hash;
    TupleRef value;
    std::int64_t key[QSX_MAX_KEYS];
  };
  std::vector<CBucket> buckets;

  std::int64_t component(const void *const *cols, int k, std::int64_t i) const {
    if (key_types[k] == QSX_INT) return static_cast<const std::int32_t *>(cols[k])[i];
    return static_cast<const std::int64_t *>(cols[k])[i];
  }
  std::uint64_t hash_row(const void *const *cols, std::int64_t i) const {
    std::uint64_t h = 0;
    for (int k = 0; k < nkeys; ++k) {
      const std::uint64_t hk = key_types[k] == QSX_INT
                                   ? hash_int(static_cast<const std::int32_t *>(cols[k])[i])
                                   : hash_long(static_cast<const std::int64_t *>(cols[k])[i]);
      h = k == 0 ? hk : combine_hashes(h, hk);
    }
    return h;
  }
  void rechain(std::uint64_t wanted_slots) {
    num_slots = next_prime(wanted_slots);
    slots.assign(num_slots, 0);
    for (std::uint64_t b = 0; b < buckets.size(); ++b) {
      std::uint64_t *pending = &slots[buckets[b].hash % num_slots];
      while (*pending != 0) pending = &buckets[*pending - 1].next;
      buckets[b].next = 0;
      *pending = b + 1;
    }
  }
};

extern "C" {

qso_cjoin_table_t *qso_cjoin_table_create(int nkeys, const int32_t *key_types, int64_t est_entries) {
  if (nkeys < 1 || nkeys > QSX_MAX_KEYS) return nullptr;
  qso_cjoin_table *t = new qso_cjoin_table();
  t->nkeys = nkeys;
  for (int k = 0; k < nkeys; ++k) t->key_types[k] = key_types[k];
  t->rechain(static_cast<std::uint64_t>(est_entries < 1 ? 1 : est_entries) * kHashTableLoadFactor);
  return t;
}
void qso_cjoin_table_destroy(qso_cjoin_table_t *t) { delete t; }

void qso_cjoin_build(qso_cjoin_table_t *t, const void *const *cols, int64_t n, uint64_t block_id, int32_t base_tid,
                     const uint64_t *filter) {
  for (int64_t i = 0; i < n; ++i) {
    if (!row_selected(filter, i)) continue;
    if ((t->buckets.size() + 1) * kHashTableLoadFactor > t->num_slots) t->rechain(t->num_slots * 2);
    qso_cjoin_table::CBucket b;
    b.next = 0;
    b.hash = t->hash_row(cols, i);
    b.value = TupleRef{block_id, static_cast<std::int32_t>(base_tid + i)};
    for (int k = 0; k < QSX_MAX_KEYS; ++k) b.key[k] = k < t->nkeys ? t->component(cols, k, i) : 0;
    t->buckets.push_back(b);
    std::uint64_t *pending = &t->slots[b.hash % t->num_slots];
    while (*pending != 0) pending = &t->buckets[*pending - 1].next;  // chain tail
    *pending = t->buckets.size();
  }
}

int64_t qso_cjoin_probe(const qso_cjoin_table_t *t, const void *const *cols, int64_t n, int32_t probe_base_tid,
                        const uint64_t *filter, int32_t *out_probe_tid, int32_t *out_build_tid, int64_t capacity) {
  int64_t count = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (!row_selected(filter, i)) continue;
    const uint64_t hash = t->hash_row(cols, i);
    uint64_t entry = t->slots[hash % t->num_slots];
    while (entry != 0) {
      const qso_cjoin_table::CBucket &bucket = t->buckets[entry - 1];
      entry = bucket.next;
      if (bucket.hash != hash) continue;
      bool same = true;  // compositeKeyCollisionCheck
      for (int k = 0; k < t->nkeys && same; ++k) same = bucket.key[k] == t->component(cols, k, i);
      if (!same) continue;
      if (count < capacity) {
        out_probe_tid[count] = static_cast<int32_t>(probe_base_tid + i);
        out_build_tid[count] = bucket.value.tuple;
      }
      ++count;
    }
  }
  return count;
}

// CombineHashes fold of one row's components (what a hashed composite key of the device path must equal).
uint64_t qso_cjoin_hash_row(const qso_cjoin_table_t *t, const void *const *cols, int64_t i) { return t->hash_row(cols, i); }

// LiteralUncheckedComparator::compareColumnVectors (LiteralComparators-inl.hpp:52-125).
void qso_select_cmp_columns(int type, const void *lhs, const void *rhs, int64_t n, int op, const uint64_t *filter,
                            uint64_t *out_bitmap) {
  std::memset(out_bitmap, 0, sizeof(uint64_t) * bitmap_words(n));
  for (int64_t i = 0; i < n; ++i) {
    if (filter != nullptr && !bit_get(filter, i)) continue;
    bool r;
    switch (type) {
      case QSX_INT: r = compare<std::int32_t>(static_cast<const std::int32_t *>(lhs)[i], op, static_cast<const std::int32_t *>(rhs)[i]); break;
      case QSX_LONG: r = compare<std::int64_t>(static_cast<const std::int64_t *>(lhs)[i], op, static_cast<const std::int64_t *>(rhs)[i]); break;
      case QSX_FLOAT: r = compare<float>(static_cast<const float *>(lhs)[i], op, static_cast<const float *>(rhs)[i]); break;
      case QSX_DATE: r = compare<DateLit>(date_at(lhs, i), op, date_at(rhs, i)); break;
      default: r = compare<double>(static_cast<const double *>(lhs)[i], op, static_cast<const double *>(rhs)[i]); break;
    }
    if (r) bit_set(out_bitmap, i);
  }
}

void qso_tids_to_bitmap(const int32_t *tids, int64_t n, int32_t base_tid, int64_t num_bits, uint64_t *out_bitmap) {
  std::memset(out_bitmap, 0, sizeof(uint64_t) * bitmap_words(num_bits));
  for (int64_t i = 0; i < n; ++i) {
    const int64_t bit = static_cast<int64_t>(tids[i]) - base_tid;
    if (bit >= 0 && bit < num_bits) bit_set(out_bitmap, bit);
  }
}

}  // extern "C"

// ===========================================================================
// aggregation
// ===========================================================================
namespace {

struct StateLayout {
  // One "state column" per int64/double accumulator.  COUNT -> 1 int64;
  // SUM -> 1 (int64 for INT/LONG arguments, double otherwise:
  // expressions/aggregation/AggregationHandleSum.cpp:45-80); AVG -> sum + count
  // (AggregationHandleAvg.cpp:45-93).
  // MIN / MAX -> current extremum + "has a value" (the reference keeps a nullable TypedValue and
  // compares with the type's less / greater comparator: AggregationHandleMin.hpp:190-215,
  // AggregationHandleMax.hpp:190-215).
  int num_states = 0;
  bool any_nullable = false;          // some column is nullable: SUM keeps its "saw a value" count
  bool state_is_int[2 * QSX_MAX_AGGS];
  int state_op[2 * QSX_MAX_AGGS];     // 0 add, 1 min, 2 max; a min/max word is followed by its has-value word
  int agg_first_state[QSX_MAX_AGGS];
  bool agg_arg_is_int[QSX_MAX_AGGS];
};

bool operand_is_int_column(const qsx_agg_config_t &c, const qsx_operand_t &o) {
  return o.kind == QSX_OPD_COLUMN &&
         (c.column_type[o.index] == QSX_INT || c.column_type[o.index] == QSX_LONG);
}

StateLayout make_layout(const qsx_agg_config_t &c) {
  StateLayout L;
  for (int col = 0; col < c.num_columns; ++col) L.any_nullable = L.any_nullable || c.column_nullable[col] != 0;
  for (int s = 0; s < 2 * QSX_MAX_AGGS; ++s) L.state_op[s] = 0;
  for (int a = 0; a < c.num_aggs; ++a) {
    L.agg_first_state[a] = L.num_states;
    const bool is_int = c.aggs[a].fn != QSX_AGG_COUNT_STAR && operand_is_int_column(c, c.aggs[a].arg);
    L.agg_arg_is_int[a] = is_int;
    switch (c.aggs[a].fn) {
      case QSX_AGG_COUNT_STAR:
        L.state_is_int[L.num_states++] = true;
        break;
      case QSX_AGG_COUNT:
        L.state_is_int[L.num_states++] = true;
        break;
      case QSX_AGG_SUM:
        L.state_is_int[L.num_states++] = is_int;
        L.state_is_int[L.num_states++] = true;   // non-NULL arguments seen (AggregationStateSum::null_, AggregationHandleSum.hpp:58-62)
        break;
      case QSX_AGG_AVG:
        L.state_is_int[L.num_states++] = is_int;
        L.state_is_int[L.num_states++] = true;
        break;
      case QSX_AGG_MIN:
      case QSX_AGG_MAX:
        L.state_op[L.num_states] = c.aggs[a].fn == QSX_AGG_MIN ? 1 : 2;
        L.state_is_int[L.num_states++] = is_int;
        L.state_is_int[L.num_states++] = true;   // has-value flag (adds like a count)
        break;
      default: std::abort();
    }
  }
  return L;
}

union StateWord {
  std::int64_t i;
  double d;
};

// Per-row evaluation context: reads typed columns, evaluates the fused
// expression program (ScalarBinaryExpression::getAllValues semantics: every
// node is an IEEE double, expressions/scalar/ScalarBinaryExpression.cpp:100-195
// over ArithmeticBinaryOperators.hpp:203-340) and the state's predicate.
// Group-by key of more than 8 bytes: 32 bytes, components at running offsets, rest zero.
struct WideKey {
  std::uint64_t w[4];
  bool operator==(const WideKey &o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; }
};
struct WideKeyHash {
  std::size_t operator()(const WideKey &k) const {
    std::uint64_t h = k.w[0];
    for (int i = 1; i < 4; ++i) h = combine_hashes(h, k.w[i]);
    return static_cast<std::size_t>(h);
  }
};

struct RowReader {
  const qsx_agg_config_t &c;
  const void *const *cols;
  const std::uint64_t *const *nulls;   // per column: null bitmap of the block (MSB-first) or nullptr
  double temps[QSX_MAX_TEMPS];
  bool temp_null[QSX_MAX_TEMPS];

  RowReader(const qsx_agg_config_t &cfg, const void *const *columns, const std::uint64_t *const *null_bitmaps = nullptr)
      : c(cfg), cols(columns), nulls(null_bitmaps) {
    for (bool &t : temp_null) t = false;
  }

  // getUntypedValue<true>() == nullptr of the reference's nullable accessors
  inline bool col_is_null(int col, std::int64_t i) const {
    return nulls != nullptr && nulls[col] != nullptr && ((nulls[col][i >> 6] >> (63 - (i & 63))) & 1u) != 0;
  }
  // NULL propagates through arithmetic (ArithmeticBinaryOperators.hpp: the nullable applyToColumnVectors variants
  // produce NULL when either operand is NULL)
  inline bool operand_is_null(const qsx_operand_t &o, std::int64_t i) const {
    return o.kind == QSX_OPD_COLUMN ? col_is_null(o.index, i) : (o.kind == QSX_OPD_TEMP ? temp_null[o.index] : false);
  }
  // a tuple with a NULL group-by key is not aggregated (PackedPayloadHashTable.hpp:861-867); a comparison with NULL
  // is not true, so neither is one with a NULL predicate operand
  inline bool row_has_null_key_or_predicate_operand(std::int64_t i) const {
    if (nulls == nullptr) return false;
    for (int k = 0; k < c.num_keys; ++k) if (col_is_null(c.key_column[k], i)) return true;
    for (int t = 0; t < c.num_pred_terms; ++t) if (col_is_null(c.pred[t].column, i)) return true;
    return false;
  }

  inline double col_as_double(int col, std::int64_t i) const {
    switch (c.column_type[col]) {
      case QSX_INT: return static_cast<const std::int32_t *>(cols[col])[i];
      case QSX_LONG: return static_cast<double>(static_cast<const std::int64_t *>(cols[col])[i]);
      case QSX_FLOAT: return static_cast<const float *>(cols[col])[i];
      case QSX_DOUBLE: return static_cast<const double *>(cols[col])[i];
      default: std::abort();
    }
  }
  inline std::int64_t col_as_int(int col, std::int64_t i) const {
    if (c.column_type[col] == QSX_INT) return static_cast<const std::int32_t *>(cols[col])[i];
    return static_cast<const std::int64_t *>(cols[col])[i];
  }
  inline double operand(const qsx_operand_t &o, std::int64_t i) const {
    switch (o.kind) {
      case QSX_OPD_COLUMN: return col_as_double(o.index, i);
      case QSX_OPD_CONST: return c.consts[o.index];
      default: return temps[o.index];
    }
  }
  inline void eval(std::int64_t i) {
    for (int k = 0; k < c.num_instrs; ++k) {
      const qsx_expr_instr_t &in = c.instrs[k];
      const double a = operand(in.a, i), b = operand(in.b, i);
      if (nulls != nullptr) temp_null[in.dst] = operand_is_null(in.a, i) || operand_is_null(in.b, i);
      double r;
      switch (in.op) {
        case QSX_EX_ADD: r = a + b; break;
        case QSX_EX_SUB: r = a - b; break;
        case QSX_EX_MUL: r = a * b; break;
        default: r = a / b; break;
      }
      temps[in.dst] = r;
    }
  }
  inline bool predicate(std::int64_t i) const {
    for (int t = 0; t < c.num_pred_terms; ++t) {
      const qsx_pred_term_t &p = c.pred[t];
      bool ok;
      switch (c.column_type[p.column]) {
        case QSX_INT: ok = compare<std::int32_t>(static_cast<const std::int32_t *>(cols[p.column])[i], p.op, p.literal.i32); break;
        case QSX_LONG: ok = compare<std::int64_t>(static_cast<const std::int64_t *>(cols[p.column])[i], p.op, p.literal.i64); break;
        case QSX_FLOAT: ok = compare<float>(static_cast<const float *>(cols[p.column])[i], p.op, p.literal.f32); break;
        case QSX_DOUBLE: ok = compare<double>(static_cast<const double *>(cols[p.column])[i], p.op, p.literal.f64); break;
        case QSX_DATE: { DateLit lit; std::memcpy(&lit, &p.literal.i64, 8); ok = compare<DateLit>(date_at(cols[p.column], i), p.op, lit); break; }
        default: std::abort();
      }
      if (!ok) return false;
    }
    return true;
  }
  // Compact key code (storage/ThreadPrivateCompactKeyHashTable.cpp:216-232,
  // .hpp:125-142): key bytes memcpy'd at running offsets into a zeroed uint64.
  inline std::uint64_t key_code(std::int64_t i) const {
    std::uint64_t code = 0;
    int offset = 0;
    for (int k = 0; k < c.num_keys; ++k) {
      const int col = c.key_column[k];
      const int w = c.column_width[col];
      std::memcpy(reinterpret_cast<char *>(&code) + offset,
                  static_cast<const char *>(cols[col]) + i * w, c.column_type[col] == QSX_DATE ? 6 : w);   // not the padding of a DateLit
      offset += w;
    }
    return code;
  }
  // A key wider than 8 bytes (PackedPayloadHashTable keeps the key components in the bucket,
  // storage/PackedPayloadHashTable.hpp:499-521, HashTableKeyManager): the same memcpy at running offsets into 32 zeroed bytes.
  inline WideKey wide_key(std::int64_t i) const {
    WideKey key{};
    int offset = 0;
    for (int k = 0; k < c.num_keys; ++k) {
      const int col = c.key_column[k];
      const int w = c.column_width[col];
      std::memcpy(reinterpret_cast<char *>(key.w) + offset, static_cast<const char *>(cols[col]) + i * w, c.column_type[col] == QSX_DATE ? 6 : w);
      offset += w;
    }
    return key;
  }
  // HashCompositeKey (utility/CompositeHash.hpp:39-48).
  inline std::uint64_t composite_hash(std::int64_t i) const {
    std::uint64_t h = 0;
    for (int k = 0; k < c.num_keys; ++k) {
      const int col = c.key_column[k];
      const int w = c.column_width[col];
      const std::uint64_t hk =
          hash_typed(c.column_type[col], static_cast<const char *>(cols[col]) + i * w);
      h = (k == 0) ? hk : combine_hashes(h, hk);
    }
    return h;
  }
};

inline void accumulate(const qsx_agg_config_t &c, const StateLayout &L, const RowReader &rr,
                       std::int64_t i, StateWord *st /* num_states words of one group */) {
  for (int a = 0; a < c.num_aggs; ++a) {
    StateWord *s = st + L.agg_first_state[a];
    // every handle but COUNT(*) skips a NULL argument (iterateUnaryInl, AggregationHandleSum.hpp:105-120;
    // AggregationHandleCount.hpp:98-118 for COUNT(x))
    if (rr.nulls != nullptr && c.aggs[a].fn != QSX_AGG_COUNT_STAR && rr.operand_is_null(c.aggs[a].arg, i)) continue;
    switch (c.aggs[a].fn) {
      case QSX_AGG_COUNT_STAR:
      case QSX_AGG_COUNT:
        s[0].i += 1;
        break;
      case QSX_AGG_SUM:
      case QSX_AGG_AVG:
        if (L.agg_arg_is_int[a]) {
          s[0].i += rr.col_as_int(c.aggs[a].arg.index, i);
        } else {
          s[0].d += rr.operand(c.aggs[a].arg, i);
        }
        // (SUM's "saw a value" word is only read for nullable arguments; AVG's second word is its count)
        if (L.any_nullable || c.aggs[a].fn == QSX_AGG_AVG) s[1].i += 1;
        break;
      case QSX_AGG_MIN:
      case QSX_AGG_MAX: {
        // iterateUnaryInl: take the value when the state is NULL or the value compares less / greater
        const bool want_less = c.aggs[a].fn == QSX_AGG_MIN;
        if (L.agg_arg_is_int[a]) {
          const std::int64_t v = rr.col_as_int(c.aggs[a].arg.index, i);
          if (s[1].i == 0 || (want_less ? v < s[0].i : v > s[0].i)) s[0].i = v;
        } else {
          const double v = rr.operand(c.aggs[a].arg, i);
          if (s[1].i == 0 || (want_less ? v < s[0].d : v > s[0].d)) s[0].d = v;
        }
        s[1].i += 1;
        break;
      }
    }
  }
}

// dst (op)= src for one state word; min/max words come with their has-value word at s + 1.
inline void merge_word(const StateLayout &L, int s, StateWord *dst, const StateWord *src) {
  if (L.state_op[s] == 0) {
    if (L.state_is_int[s]) dst[s].i += src[s].i; else dst[s].d += src[s].d;
    return;
  }
  if (src[s + 1].i == 0) return;   // mergeStates: a NULL source state changes nothing
  const bool want_less = L.state_op[s] == 1;
  if (L.state_is_int[s]) {
    if (dst[s + 1].i == 0 || (want_less ? src[s].i < dst[s].i : src[s].i > dst[s].i)) dst[s].i = src[s].i;
  } else {
    if (dst[s + 1].i == 0 || (want_less ? src[s].d < dst[s].d : src[s].d > dst[s].d)) dst[s].d = src[s].d;
  }
}
inline void merge_words(const StateLayout &L, StateWord *dst, const StateWord *src) {
  // extrema first: they look at dst's has-value word before the add below changes it
  for (int s = 0; s < L.num_states; ++s) if (L.state_op[s] != 0) merge_word(L, s, dst, src);
  for (int s = 0; s < L.num_states; ++s) if (L.state_op[s] == 0) merge_word(L, s, dst, src);
}

}  // namespace

struct qso_agg_state {
  qsx_agg_config_t c;
  StateLayout L;
  int key_bytes = 0;

  // SINGLE_STATE (storage/AggregationOperationState.cpp:476-519)
  std::vector<StateWord> single;
  std::int64_t single_rows = 0;

  // COMPACT_KEY (storage/ThreadPrivateCompactKeyHashTable.{hpp,cpp}) and
  // GENERIC (PackedPayloadHashTable; insertion-ordered buckets + index).
  std::unordered_map<std::uint64_t, std::uint32_t> index;
  std::vector<std::uint64_t> keys;        // bucket -> key code
  std::vector<std::uint64_t> key_hashes;  // GENERIC: composite hash per bucket
  // keys wider than 8 bytes: the same bucket numbering through these two instead of index / keys
  std::unordered_map<WideKey, std::uint32_t, WideKeyHash> wide_index;
  std::vector<WideKey> wide_keys;
  std::vector<StateWord> states;          // bucket-major: [bucket][state]

  // COLLISION_FREE (storage/CollisionFreeVectorTable.hpp)
  std::vector<std::uint64_t> existence;   // LSB-first bit array
  std::vector<StateWord> dense;           // state-major: [state][key]

  explicit qso_agg_state(const qsx_agg_config_t &cfg) : c(cfg), L(make_layout(cfg)) {
    for (int k = 0; k < c.num_keys; ++k) key_bytes += c.column_width[c.key_column[k]];
    if (c.strategy == QSX_AGG_SINGLE_STATE) {
      single.assign(L.num_states, StateWord{0});
    } else if (c.strategy == QSX_AGG_COLLISION_FREE) {
      existence.assign((c.num_entries + 63) / 64, 0);
      dense.assign(static_cast<std::size_t>(L.num_states) * c.num_entries, StateWord{0});
    }
  }

  std::uint32_t bucket_for(std::uint64_t code, std::uint64_t hash) {
    auto it = index.find(code);
    if (it != index.end()) return it->second;
    const std::uint32_t b = static_cast<std::uint32_t>(keys.size());
    index.emplace(code, b);
    keys.push_back(code);
    key_hashes.push_back(hash);
    states.resize(states.size() + L.num_states, StateWord{0});
    return b;
  }

  std::uint32_t bucket_for_wide(const WideKey &key, std::uint64_t hash) {
    auto it = wide_index.find(key);
    if (it != wide_index.end()) return it->second;
    const std::uint32_t b = static_cast<std::uint32_t>(wide_keys.size());
    wide_index.emplace(key, b);
    wide_keys.push_back(key);
    key_hashes.push_back(hash);
    states.resize(states.size() + L.num_states, StateWord{0});
    return b;
  }
  std::size_t num_buckets() const { return key_bytes > 8 ? wide_keys.size() : keys.size(); }

  void update(const void *const *cols, std::int64_t n, const std::uint64_t *filter, const std::uint64_t *const *nulls = nullptr) {
    RowReader rr(c, cols, nulls);
    if (c.strategy == QSX_AGG_SINGLE_STATE) {
      // aggregateBlockSingleState: accumulate a block-local state, then merge
      // into the global one (mergeStates under a mutex, :476-519).
      std::vector<StateWord> local(L.num_states, StateWord{0});
      std::int64_t rows = 0;
      for (std::int64_t i = 0; i < n; ++i) {
        if (!row_selected(filter, i) || rr.row_has_null_key_or_predicate_operand(i) || !rr.predicate(i)) continue;
        rr.eval(i);
        accumulate(c, L, rr, i, local.data());
        ++rows;
      }
      merge_words(L, single.data(), local.data());
      single_rows += rows;
      return;
    }
    for (std::int64_t i = 0; i < n; ++i) {
      if (!row_selected(filter, i) || rr.row_has_null_key_or_predicate_operand(i) || !rr.predicate(i)) continue;
      rr.eval(i);
      if (c.strategy == QSX_AGG_COLLISION_FREE) {
        // upsertValueAccessor* (CollisionFreeVectorTable.hpp:530-645): loc = key.
        const std::int64_t loc = rr.col_as_int(c.key_column[0], i);
        existence[loc >> 6] |= (static_cast<std::uint64_t>(1) << (loc & 63));
        StateWord group[2 * QSX_MAX_AGGS];
        for (int s = 0; s < L.num_states; ++s) group[s] = dense[static_cast<std::size_t>(s) * c.num_entries + loc];
        accumulate(c, L, rr, i, group);
        for (int s = 0; s < L.num_states; ++s) dense[static_cast<std::size_t>(s) * c.num_entries + loc] = group[s];
      } else {
        const std::uint64_t h = (c.strategy == QSX_AGG_GENERIC) ? rr.composite_hash(i) : 0;
        const std::uint32_t b = key_bytes > 8 ? bucket_for_wide(rr.wide_key(i), h) : bucket_for(rr.key_code(i), h);
        accumulate(c, L, rr, i, &states[static_cast<std::size_t>(b) * L.num_states]);
      }
    }
  }

  void merge_from(const qso_agg_state &src) {
    switch (c.strategy) {
      case QSX_AGG_SINGLE_STATE:
        merge_words(L, single.data(), src.single.data());
        single_rows += src.single_rows;
        break;
      case QSX_AGG_COLLISION_FREE:
        for (std::size_t w = 0; w < existence.size(); ++w) existence[w] |= src.existence[w];
        for (std::int64_t k = 0; k < c.num_entries; ++k) {
          StateWord d[2 * QSX_MAX_AGGS], r[2 * QSX_MAX_AGGS];
          for (int s = 0; s < L.num_states; ++s) {
            d[s] = dense[static_cast<std::size_t>(s) * c.num_entries + k];
            r[s] = src.dense[static_cast<std::size_t>(s) * c.num_entries + k];
          }
          merge_words(L, d, r);
          for (int s = 0; s < L.num_states; ++s) dense[static_cast<std::size_t>(s) * c.num_entries + k] = d[s];
        }
        break;
      default:
        // mergeFrom (ThreadPrivateCompactKeyHashTable.cpp:306-363): map source
        // buckets to destination buckets (new keys appended), then add columns.
        for (std::size_t b = 0; b < src.num_buckets(); ++b) {
          const std::uint32_t d = key_bytes > 8 ? bucket_for_wide(src.wide_keys[b], src.key_hashes[b])
                                                : bucket_for(src.keys[b], src.key_hashes[b]);
          merge_words(L, &states[static_cast<std::size_t>(d) * L.num_states],
                      &src.states[b * L.num_states]);
        }
        break;
    }
  }
};

namespace {

// Write one finalized group row.  SUM(int) -> int64, SUM(fp) -> double,
// COUNT -> int64, AVG -> sum / (double)count (AggregationHandleAvg.cpp:144-155).
void emit_values(const qsx_agg_config_t &c, const StateLayout &L, const StateWord *st, bool empty_group,
                 std::int64_t row, void *const *out_val_cols, std::uint8_t *const *out_null_cols) {
  for (int a = 0; a < c.num_aggs; ++a) {
    const StateWord *s = st + L.agg_first_state[a];
    bool is_null = false;
    switch (c.aggs[a].fn) {
      case QSX_AGG_COUNT_STAR:
      case QSX_AGG_COUNT:
        static_cast<std::int64_t *>(out_val_cols[a])[row] = s[0].i;
        break;
      case QSX_AGG_SUM: {
        // SUM over zero rows is NULL (AggregationHandleSum.cpp:100-120); so is the SUM of a group whose nullable
        // argument was NULL in every row.  (A non-nullable argument: the compact-key and collision-free tables keep
        // no null flag — a key that only has its existence bit finalizes as 0, CollisionFreeVectorTable.hpp:700-727.)
        bool temp_nullable[QSX_MAX_TEMPS] = {};
        auto nullable = [&](const qsx_operand_t &o) {
          return o.kind == QSX_OPD_COLUMN ? c.column_nullable[o.index] != 0 : (o.kind == QSX_OPD_TEMP && temp_nullable[o.index]);
        };
        for (int k = 0; k < c.num_instrs; ++k) temp_nullable[c.instrs[k].dst] = nullable(c.instrs[k].a) || nullable(c.instrs[k].b);
        is_null = empty_group || (nullable(c.aggs[a].arg) && s[1].i == 0);
        if (L.agg_arg_is_int[a]) static_cast<std::int64_t *>(out_val_cols[a])[row] = is_null ? 0 : s[0].i;
        else static_cast<double *>(out_val_cols[a])[row] = is_null ? 0.0 : s[0].d;
        break;
      }
      case QSX_AGG_AVG: {
        const double sum = L.agg_arg_is_int[a] ? static_cast<double>(s[0].i) : s[0].d;
        is_null = (s[1].i == 0);
        static_cast<double *>(out_val_cols[a])[row] = is_null ? 0.0 : sum / static_cast<double>(s[1].i);
        break;
      }
      case QSX_AGG_MIN:
      case QSX_AGG_MAX: {
        // finalize: the extremum, typed like the argument; NULL when no value was seen
        // (AggregationHandleMin.cpp:100-120)
        is_null = (s[1].i == 0);
        const int vt = c.aggs[a].arg.kind == QSX_OPD_COLUMN ? c.column_type[c.aggs[a].arg.index] : QSX_DOUBLE;
        switch (vt) {
          case QSX_INT: static_cast<std::int32_t *>(out_val_cols[a])[row] = is_null ? 0 : static_cast<std::int32_t>(s[0].i); break;
          case QSX_LONG: static_cast<std::int64_t *>(out_val_cols[a])[row] = is_null ? 0 : s[0].i; break;
          case QSX_FLOAT: static_cast<float *>(out_val_cols[a])[row] = is_null ? 0.0f : static_cast<float>(s[0].d); break;
          default: static_cast<double *>(out_val_cols[a])[row] = is_null ? 0.0 : s[0].d; break;
        }
        break;
      }
    }
    if (out_null_cols != nullptr && out_null_cols[a] != nullptr) out_null_cols[a][row] = is_null ? 1 : 0;
  }
}

void emit_keys_from_bytes(const qsx_agg_config_t &c, const char *key_bytes, std::int64_t row, void *const *out_key_cols) {
  int offset = 0;
  for (int k = 0; k < c.num_keys; ++k) {
    const int w = c.column_width[c.key_column[k]];
    std::memcpy(static_cast<char *>(out_key_cols[k]) + row * w, key_bytes + offset, w);
    offset += w;
  }
}
void emit_keys_from_code(const qsx_agg_config_t &c, std::uint64_t code, std::int64_t row,
                         void *const *out_key_cols) {
  // finalize (ThreadPrivateCompactKeyHashTable.cpp:365-421): memcpy width_i bytes from code+offset_i.
  emit_keys_from_bytes(c, reinterpret_cast<const char *>(&code), row, out_key_cols);
}

}  // namespace

extern "C" {

// ScalarBinaryExpression::getAllValues (expressions/scalar/ScalarBinaryExpression.cpp:100-195): every node evaluated into
// its own vector in IEEE double, left to right as written; here row by row with the same per-node rounding.
void qso_eval_expression(int num_columns, const void *const *cols, const int32_t *types, int num_instrs,
                         const qsx_expr_instr_t *instrs, const double *consts, qsx_operand_t result, int64_t n, double *out) {
  (void)num_columns;
  for (int64_t i = 0; i < n; ++i) {
    double temps[QSX_MAX_TEMPS] = {};
    auto operand = [&](const qsx_operand_t &o) -> double {
      switch (o.kind) {
        case QSX_OPD_COLUMN:
          switch (types[o.index]) {
            case QSX_INT: return static_cast<double>(static_cast<const std::int32_t *>(cols[o.index])[i]);
            case QSX_LONG: return static_cast<double>(static_cast<const std::int64_t *>(cols[o.index])[i]);
            case QSX_FLOAT: return static_cast<double>(static_cast<const float *>(cols[o.index])[i]);
            default: return static_cast<const double *>(cols[o.index])[i];
          }
        case QSX_OPD_CONST: return consts[o.index];
        default: return temps[o.index];
      }
    };
    for (int k = 0; k < num_instrs; ++k) {
      const double a = operand(instrs[k].a), b = operand(instrs[k].b);
      double r;
      switch (instrs[k].op) {
        case QSX_EX_ADD: r = a + b; break;
        case QSX_EX_SUB: r = a - b; break;
        case QSX_EX_MUL: r = a * b; break;
        default: r = a / b; break;
      }
      temps[instrs[k].dst] = r;
    }
    out[i] = operand(result);
  }
}

qso_agg_state_t *qso_agg_state_create(const qsx_agg_config_t *config) {
  return new qso_agg_state(*config);
}
void qso_agg_state_destroy(qso_agg_state_t *s) { delete s; }

void qso_agg_update(qso_agg_state_t *s, const void *const *cols, int64_t n, const uint64_t *filter) {
  s->update(cols, n, filter);
}
void qso_agg_update_nullable(qso_agg_state_t *s, const void *const *cols, const uint64_t *const *null_bitmaps, int64_t n,
                             const uint64_t *filter) {
  s->update(cols, n, filter, null_bitmaps);
}
void qso_agg_mark_existence(qso_agg_state_t *s, int key_type, const void *keys, int64_t n, const uint64_t *filter) {
  // ExecuteBuild (BuildAggregationExistenceMapOperator.cpp:50-67): setBit(value) for every tuple of the accessor
  for (int64_t i = 0; i < n; ++i) {
    if (!row_selected(filter, i)) continue;
    const int64_t loc = key_type == QSX_INT ? static_cast<const int32_t *>(keys)[i] : static_cast<const int64_t *>(keys)[i];
    if (loc < 0 || loc >= static_cast<int64_t>(s->c.num_entries)) continue;
    s->existence[loc >> 6] |= (static_cast<uint64_t>(1) << (loc & 63));
  }
}
void qso_agg_update_coded(qso_agg_state_t *s, const void *const *cols, const void *const *dictionaries, int64_t n,
                          const uint64_t *filter) {
  // CompressedColumnStoreValueAccessor::getUntypedValue (storage/CompressedColumnStoreValueAccessor.hpp:90-150): a
  // dictionary-coded attribute returns dictionary[code], a truncated one the code widened to the attribute's type.
  const qsx_agg_config_t &c = s->c;
  std::vector<std::vector<unsigned char>> decoded(c.num_columns);
  std::vector<const void *> plain(c.num_columns);
  for (int col = 0; col < c.num_columns; ++col) {
    const int cw = c.column_code_width[col];
    plain[col] = cols[col];
    if (cw == 0) continue;
    const int w = c.column_width[col];
    decoded[col].resize(static_cast<size_t>(n) * w);
    for (int64_t i = 0; i < n; ++i) {
      uint32_t code = 0;
      std::memcpy(&code, static_cast<const char *>(cols[col]) + i * cw, cw);
      unsigned char *out = decoded[col].data() + i * w;
      if (dictionaries != nullptr && dictionaries[col] != nullptr) {
        std::memcpy(out, static_cast<const char *>(dictionaries[col]) + static_cast<size_t>(code) * w, w);
      } else {
        switch (c.column_type[col]) {
          case QSX_INT: { const int32_t v = static_cast<int32_t>(code); std::memcpy(out, &v, 4); break; }
          case QSX_LONG: { const int64_t v = static_cast<int64_t>(code); std::memcpy(out, &v, 8); break; }
          case QSX_FLOAT: { const float v = static_cast<float>(code); std::memcpy(out, &v, 4); break; }
          default: { const double v = static_cast<double>(code); std::memcpy(out, &v, 8); break; }
        }
      }
    }
    plain[col] = decoded[col].data();
  }
  s->update(plain.data(), n, filter);
}
void qso_agg_merge(qso_agg_state_t *dst, const qso_agg_state_t *src) { dst->merge_from(*src); }

int64_t qso_agg_num_groups(const qso_agg_state_t *s) {
  switch (s->c.strategy) {
    case QSX_AGG_SINGLE_STATE: return 1;
    case QSX_AGG_COLLISION_FREE: {
      int64_t c = 0;
      for (uint64_t w : s->existence) c += __builtin_popcountll(w);
      return c;
    }
    default: return static_cast<int64_t>(s->num_buckets());
  }
}

int64_t qso_agg_finalize(const qso_agg_state_t *s, int partition, int num_partitions,
                         void *const *out_key_cols, void *const *out_val_cols,
                         uint8_t *const *out_null_cols, int64_t capacity) {
  const qsx_agg_config_t &c = s->c;
  const StateLayout &L = s->L;
  int64_t row = 0;
  switch (c.strategy) {
    case QSX_AGG_SINGLE_STATE:
      // finalizeSingleState (:652-670): always exactly one row (partition 0).
      if (partition != 0 || capacity < 1) return 0;
      emit_values(c, L, s->single.data(), s->single_rows == 0, 0, out_val_cols, out_null_cols);
      return 1;
    case QSX_AGG_COLLISION_FREE: {
      // finalizeKey/finalizeState (CollisionFreeVectorTable.hpp:647-727) on the
      // key range of this partition (:192-208): ascending key order.
      const int64_t len = (c.num_entries + num_partitions - 1) / num_partitions;
      const int64_t begin = static_cast<int64_t>(partition) * len;
      const int64_t end = std::min<int64_t>(begin + len, c.num_entries);
      const int kw = c.column_width[c.key_column[0]];
      for (int64_t loc = begin; loc < end; ++loc) {
        if (!((s->existence[loc >> 6] >> (loc & 63)) & 1u)) continue;
        if (row >= capacity) return row;
        if (kw == 4) static_cast<int32_t *>(out_key_cols[0])[row] = static_cast<int32_t>(loc);
        else static_cast<int64_t *>(out_key_cols[0])[row] = loc;
        StateWord group[2 * QSX_MAX_AGGS];
        for (int st = 0; st < L.num_states; ++st) group[st] = s->dense[static_cast<size_t>(st) * c.num_entries + loc];
        emit_values(c, L, group, false, row, out_val_cols, out_null_cols);
        ++row;
      }
      return row;
    }
    default:
      for (size_t b = 0; b < s->num_buckets(); ++b) {
        if (c.strategy == QSX_AGG_GENERIC) {
          // partitioned mode routes a row to HashCompositeKey % P
          // (storage/AggregationOperationState.cpp:576-583).
          if (static_cast<int>(s->key_hashes[b] % static_cast<uint64_t>(num_partitions)) != partition) continue;
        } else if (partition != 0) {
          continue;  // compact-key tables finalize in one piece (:925-948)
        }
        if (row >= capacity) return row;
        if (s->key_bytes > 8) emit_keys_from_bytes(c, reinterpret_cast<const char *>(s->wide_keys[b].w), row, out_key_cols);
        else emit_keys_from_code(c, s->keys[b], row, out_key_cols);
        emit_values(c, L, &s->states[b * L.num_states], false, row, out_val_cols, out_null_cols);
        ++row;
      }
      return row;
  }
}

}  // extern "C"

// ===========================================================================
// LIP filters
// ===========================================================================
struct qso_lip_filter {
  int kind;
  std::int64_t cardinality;
  std::int64_t min_value;
  bool is_anti;
  std::vector<std::uint64_t> bits;
};

namespace {
inline std::int64_t key_as_i64(int key_type, const void *keys, std::int64_t i) {
  if (key_type == QSX_INT) return static_cast<const std::int32_t *>(keys)[i];
  return static_cast<const std::int64_t *>(keys)[i];
}
// SingleIdentityHashFilter (utility/lip_filter/SingleIdentityHashFilter.hpp:156-169):
// hash = value % filter_cardinality_ with the value converted to size_t first.
inline std::uint64_t identity_bit(std::int64_t v, std::int64_t card) {
  return static_cast<std::uint64_t>(v) % static_cast<std::uint64_t>(card);
}
}  // namespace

extern "C" {

qso_lip_filter_t *qso_lip_filter_create(int kind, int64_t cardinality, int64_t min_value, int is_anti) {
  qso_lip_filter_t *f = new qso_lip_filter_t();
  f->kind = kind;
  f->cardinality = cardinality;
  f->min_value = min_value;
  f->is_anti = is_anti != 0;
  f->bits.assign((cardinality + 63) / 64, 0);
  return f;
}
void qso_lip_filter_destroy(qso_lip_filter_t *f) { delete f; }

void qso_lip_build(qso_lip_filter_t *f, int key_type, const void *keys, int64_t n, const uint64_t *filter) {
  for (int64_t i = 0; i < n; ++i) {
    if (!row_selected(filter, i)) continue;
    const int64_t v = key_as_i64(key_type, keys, i);
    uint64_t bit;
    if (f->kind == QSX_LIP_BITVECTOR_EXACT) {
      bit = static_cast<uint64_t>(v - f->min_value);  // BitVectorExactFilter.hpp:150-156
    } else {
      bit = identity_bit(v, f->cardinality);
    }
    f->bits[bit >> 6] |= static_cast<uint64_t>(1) << (bit & 63);
  }
}

void qso_lip_probe(const qso_lip_filter_t *f, int key_type, const void *keys, int64_t n,
                   const uint64_t *in_bitmap, uint64_t *out_bitmap) {
  std::memset(out_bitmap, 0, sizeof(uint64_t) * bitmap_words(n));
  for (int64_t i = 0; i < n; ++i) {
    if (!row_selected(in_bitmap, i)) continue;
    const int64_t v = key_as_i64(key_type, keys, i);
    bool hit;
    if (f->kind == QSX_LIP_BITVECTOR_EXACT) {
      // contains (BitVectorExactFilter.hpp:158-172)
      if (v < f->min_value || v > f->min_value + f->cardinality - 1) {
        hit = f->is_anti;
      } else {
        const uint64_t bit = static_cast<uint64_t>(v - f->min_value);
        const bool set = (f->bits[bit >> 6] >> (bit & 63)) & 1u;
        hit = f->is_anti ? !set : set;
      }
    } else {
      const uint64_t bit = identity_bit(v, f->cardinality);
      hit = (f->bits[bit >> 6] >> (bit & 63)) & 1u;
    }
    if (hit) bit_set(out_bitmap, i);
  }
}

// ---------------------------------------------------------------------------
// partition scatter (PartitionAwareInsertDestination routing, storage/
// InsertDestination.hpp:490-660, with HashPartitionSchemeHeader ids).
// ---------------------------------------------------------------------------
void qso_partition_offsets(int key_type, const void *keys, int64_t n, int num_partitions, int64_t *offsets) {
  std::vector<int64_t> counts(num_partitions, 0);
  const int w = type_width(key_type);
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t h = hash_typed(key_type, static_cast<const char *>(keys) + i * w);
    ++counts[partition_id(h, num_partitions)];
  }
  offsets[0] = 0;
  for (int p = 0; p < num_partitions; ++p) offsets[p + 1] = offsets[p] + counts[p];
}

void qso_partition_scatter_col(int key_type, const void *keys, int64_t n, int num_partitions,
                               int width, const void *col, void *out_col) {
  std::vector<int64_t> offsets(num_partitions + 1);
  qso_partition_offsets(key_type, keys, n, num_partitions, offsets.data());
  const int w = type_width(key_type);
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t h = hash_typed(key_type, static_cast<const char *>(keys) + i * w);
    const int64_t pos = offsets[partition_id(h, num_partitions)]++;
    std::memcpy(static_cast<char *>(out_col) + pos * width, static_cast<const char *>(col) + i * width, width);
  }
}

}  // extern "C"

// ===========================================================================
// CPU baseline drivers
// ===========================================================================
namespace {

using Clock = std::chrono::steady_clock;
inline double seconds_since(Clock::time_point t0) {
  return std::chrono::duration<double>(Clock::now() - t0).count();
}

// Stand-in for Foreman/Worker (query_execution/ForemanSingleNode.cpp:102-178,
// Worker.cpp:54-139): work orders = block indices, workers pull from a shared
// atomic queue and run execute() until it is drained.
template <typename Fn>
void run_work_orders(std::int64_t num_blocks, int num_threads, Fn execute) {
  std::atomic<std::int64_t> next{0};
  std::vector<std::thread> workers;
  for (int t = 0; t < num_threads; ++t) {
    workers.emplace_back([&, t]() {
      for (;;) {
        const std::int64_t b = next.fetch_add(1, std::memory_order_relaxed);
        if (b >= num_blocks) break;
        execute(b, t);
      }
    });
  }
  for (auto &w : workers) w.join();
}

}  // namespace

extern "C" {

void qso_bench_join(int key_type, const void *build_keys, int64_t n_build, const void *probe_keys,
                    int64_t n_probe, int64_t block_rows, int num_threads,
                    qso_join_bench_result_t *out) {
  const int kw = type_width(key_type);
  qso_join_table_t *table = qso_join_table_create(key_type, n_build);
  const int64_t build_blocks = (n_build + block_rows - 1) / block_rows;
  const int64_t probe_blocks = (n_probe + block_rows - 1) / block_rows;

  Clock::time_point t0 = Clock::now();
  run_work_orders(build_blocks, num_threads, [&](int64_t b, int) {
    // BuildHashWorkOrder::execute (relational_operators/BuildHashOperator.cpp:162-207)
    const int64_t begin = b * block_rows, rows = std::min(block_rows, n_build - begin);
    table->put_block(static_cast<const char *>(build_keys) + begin * kw, rows,
                     static_cast<uint64_t>(b), static_cast<int32_t>(begin), nullptr);
  });
  out->build_seconds = seconds_since(t0);

  std::vector<int64_t> matches(num_threads, 0);
  std::vector<uint64_t> checksums(num_threads, 0);
  t0 = Clock::now();
  run_work_orders(probe_blocks, num_threads, [&](int64_t b, int t) {
    // HashInnerJoinWorkOrder::execute (relational_operators/HashJoinOperator.cpp:450-541):
    // collect (probe_tid, build_tid) pairs per build block in an unordered_map
    // of vectors (:76-102), then visit each build block's pair list.
    const int64_t begin = b * block_rows, rows = std::min(block_rows, n_probe - begin);
    const char *keys = static_cast<const char *>(probe_keys) + begin * kw;
    std::unordered_map<uint64_t, std::vector<std::pair<int32_t, int32_t>>> joined;
    for (int64_t i = 0; i < rows; ++i) {
      const uint64_t hash = table->key_hash(keys, i);
      uint64_t entry = table->st.slots[hash % table->st.num_slots].load(std::memory_order_relaxed);
      while (entry != 0) {
        const Bucket &bucket = table->st.buckets[entry - 1];
        entry = bucket.next.load(std::memory_order_relaxed);
        if (bucket.hash == hash) {
          joined[bucket.value.block].emplace_back(static_cast<int32_t>(begin + i), bucket.value.tuple);
        }
      }
    }
    for (const auto &kv : joined) {
      for (const auto &pr : kv.second) {
        checksums[t] += static_cast<uint64_t>(static_cast<uint32_t>(pr.first)) * 1000003ULL +
                        static_cast<uint64_t>(static_cast<uint32_t>(pr.second));
      }
      matches[t] += static_cast<int64_t>(kv.second.size());
    }
  });
  out->probe_seconds = seconds_since(t0);
  out->matches = 0;
  out->checksum = 0;
  for (int t = 0; t < num_threads; ++t) {
    out->matches += matches[t];
    out->checksum += checksums[t];
  }
  qso_join_table_destroy(table);
}

double qso_bench_agg(const qsx_agg_config_t *config, const void *const *cols, int64_t n,
                     int64_t block_rows, int num_threads, qso_agg_state_t **out_state) {
  const int64_t blocks = (n + block_rows - 1) / block_rows;
  // One private state per worker (HashTablePool hands a returned table to the
  // next work order; with T workers at most T tables exist), merged at
  // finalize (storage/AggregationOperationState.cpp:925-948).
  std::vector<qso_agg_state_t *> priv(num_threads, nullptr);
  for (int t = 0; t < num_threads; ++t) priv[t] = qso_agg_state_create(config);
  Clock::time_point t0 = Clock::now();
  run_work_orders(blocks, num_threads, [&](int64_t b, int t) {
    const int64_t begin = b * block_rows, rows = std::min(block_rows, n - begin);
    const void *block_cols[QSX_MAX_COLUMNS];
    for (int c = 0; c < config->num_columns; ++c) {
      block_cols[c] = static_cast<const char *>(cols[c]) + begin * config->column_width[c];
    }
    priv[t]->update(block_cols, rows, nullptr);
  });
  for (int t = 1; t < num_threads; ++t) priv[0]->merge_from(*priv[t]);
  const double elapsed = seconds_since(t0);
  for (int t = 1; t < num_threads; ++t) qso_agg_state_destroy(priv[t]);
  if (out_state != nullptr) *out_state = priv[0]; else qso_agg_state_destroy(priv[0]);
  return elapsed;
}

double qso_bench_agg_coded(const qsx_agg_config_t *config, const void *const *cols, const void *const *dictionaries,
                           int64_t n, int64_t block_rows, int num_threads, qso_agg_state_t **out_state) {
  // AggregationWorkOrder over CompressedColumnStore blocks: the accessor hands out dictionary[code] / the widened code
  // per value (storage/CompressedColumnStoreValueAccessor.hpp:90-150); the rest is qso_bench_agg (per-worker tables
  // from the pool, merged at finalize).
  const int64_t blocks = (n + block_rows - 1) / block_rows;
  std::vector<qso_agg_state_t *> priv(num_threads, nullptr);
  for (int t = 0; t < num_threads; ++t) priv[t] = qso_agg_state_create(config);
  Clock::time_point t0 = Clock::now();
  run_work_orders(blocks, num_threads, [&](int64_t b, int t) {
    const int64_t begin = b * block_rows, rows = std::min(block_rows, n - begin);
    const void *block_cols[QSX_MAX_COLUMNS];
    for (int c = 0; c < config->num_columns; ++c) {
      const int w = config->column_code_width[c] != 0 ? config->column_code_width[c] : config->column_width[c];
      block_cols[c] = static_cast<const char *>(cols[c]) + begin * w;
    }
    qso_agg_update_coded(priv[t], block_cols, dictionaries, rows, nullptr);
  });
  for (int t = 1; t < num_threads; ++t) priv[0]->merge_from(*priv[t]);
  const double elapsed = seconds_since(t0);
  for (int t = 1; t < num_threads; ++t) qso_agg_state_destroy(priv[t]);
  if (out_state != nullptr) *out_state = priv[0]; else qso_agg_state_destroy(priv[0]);
  return elapsed;
}

}  // extern "C"

namespace {

// One relation of (INT key, 8-byte payload) kept as blocks: what a PartitionAwareInsertDestination leaves behind for one
// partition (storage/InsertDestination.hpp:490-660) — a list of blocks, each filled up to `block_rows` tuples.
struct KeyPayloadBlock {
  std::vector<std::int32_t> key;
  std::vector<std::int64_t> payload;
};
struct PartitionBlocks {
  std::mutex mutex;                       // available_block_refs_ / done_block_ids_ are guarded per partition (:640-660)
  std::vector<KeyPayloadBlock> blocks;    // the last one is the partially filled block
};

// Select with has_repartition (relational_operators/SelectOperator.cpp:161-195 -> PartitionAwareInsertDestination::
// bulkInsertTuples, storage/InsertDestination.cpp:560-640): per input block the partition id of every tuple
// (HashPartitionSchemeHeader::getPartitionId, catalog/PartitionSchemeHeader.hpp:200-214), then one bulk insert per
// partition into that partition's current block(s).
void repartition_relation(const std::int32_t *key, const std::int64_t *payload, std::int64_t n, int num_partitions,
                          std::int64_t block_rows, int num_threads, std::vector<PartitionBlocks> *parts) {
  const std::int64_t blocks = (n + block_rows - 1) / block_rows;
  run_work_orders(blocks, num_threads, [&](std::int64_t b, int) {
    const std::int64_t begin = b * block_rows, rows = std::min(block_rows, n - begin);
    std::vector<std::vector<std::int32_t>> membership(num_partitions);     // tuple ids per partition (:575-590)
    for (std::int64_t i = 0; i < rows; ++i) {
      const std::uint64_t h = hash_int(key[begin + i]);
      membership[partition_id(h, num_partitions)].push_back(static_cast<std::int32_t>(i));
    }
    for (int p = 0; p < num_partitions; ++p) {
      const std::vector<std::int32_t> &mine = membership[p];
      std::size_t done = 0;
      while (done < mine.size()) {
        // getBlockForInsertionInPartition: take the partition's partially filled block out of the pool under its mutex,
        // fill it outside the lock, return it (:600-640)
        KeyPayloadBlock blk;
        {
          std::lock_guard<std::mutex> lock((*parts)[p].mutex);
          std::vector<KeyPayloadBlock> &list = (*parts)[p].blocks;
          if (!list.empty() && static_cast<std::int64_t>(list.back().key.size()) < block_rows) {
            blk = std::move(list.back());
            list.pop_back();
          }
        }
        const std::size_t room = static_cast<std::size_t>(block_rows) - blk.key.size();
        const std::size_t take = std::min(room, mine.size() - done);
        for (std::size_t j = 0; j < take; ++j) {
          blk.key.push_back(key[begin + mine[done + j]]);
          blk.payload.push_back(payload[begin + mine[done + j]]);
        }
        done += take;
        std::lock_guard<std::mutex> lock((*parts)[p].mutex);
        std::vector<KeyPayloadBlock> &list = (*parts)[p].blocks;
        // keep the invariant "only the last block may be partially filled"
        if (!list.empty() && static_cast<std::int64_t>(list.back().key.size()) < block_rows &&
            static_cast<std::int64_t>(blk.key.size()) == block_rows) {
          list.insert(list.end() - 1, std::move(blk));
        } else {
          list.push_back(std::move(blk));
        }
      }
    }
  });
}

inline void atomic_or_bit(std::atomic<std::uint64_t> *words, std::uint64_t bit) {
  // BarrieredReadWriteConcurrentBitVector::setBit (utility/BarrieredReadWriteConcurrentBitVector.hpp): fetch_or, relaxed
  words[bit >> 6].fetch_or(static_cast<std::uint64_t>(1) << (bit & 63), std::memory_order_relaxed);
}
inline bool test_bit(const std::atomic<std::uint64_t> *words, std::uint64_t bit) {
  return (words[bit >> 6].load(std::memory_order_relaxed) >> (bit & 63)) & 1u;
}

}  // namespace

extern "C" {

void qso_bench_partitioned_join(const int32_t *o_key, const int64_t *o_payload, int64_t n_o, const int32_t *l_key,
                                const int64_t *l_payload, int64_t n_l, int num_partitions, int64_t block_rows,
                                int num_threads, qso_partitioned_join_result_t *out) {
  // BASELINE config 4 the way one reference process runs it (all partitions in one address space): both relations pass
  // a repartitioning Select, then per partition BuildHashWorkOrders (BuildHashOperator.cpp:82-91, 162-207) and
  // HashInnerJoinWorkOrders (HashJoinOperator.cpp:220-231, 450-541) over that partition's blocks and table.
  std::vector<PartitionBlocks> o_parts(num_partitions), l_parts(num_partitions);
  Clock::time_point t0 = Clock::now();
  repartition_relation(o_key, o_payload, n_o, num_partitions, block_rows, num_threads, &o_parts);
  repartition_relation(l_key, l_payload, n_l, num_partitions, block_rows, num_threads, &l_parts);
  out->repartition_seconds = seconds_since(t0);

  // one JoinHashTable per partition, sized by the optimizer's estimate of the partition (ExecutionGenerator.cpp:903-904)
  std::vector<qso_join_table_t *> tables(num_partitions);
  struct Unit { int part; std::int64_t block; };
  std::vector<Unit> build_units, probe_units;
  for (int p = 0; p < num_partitions; ++p) {
    tables[p] = qso_join_table_create(QSX_INT, std::max<std::int64_t>(1, n_o / num_partitions));
    for (std::size_t b = 0; b < o_parts[p].blocks.size(); ++b) build_units.push_back({p, static_cast<std::int64_t>(b)});
    for (std::size_t b = 0; b < l_parts[p].blocks.size(); ++b) probe_units.push_back({p, static_cast<std::int64_t>(b)});
  }
  t0 = Clock::now();
  run_work_orders(static_cast<std::int64_t>(build_units.size()), num_threads, [&](std::int64_t u, int) {
    const Unit &unit = build_units[u];
    const KeyPayloadBlock &blk = o_parts[unit.part].blocks[unit.block];
    tables[unit.part]->put_block(blk.key.data(), static_cast<std::int64_t>(blk.key.size()),
                                 static_cast<std::uint64_t>(unit.block), 0, nullptr);
  });
  out->build_seconds = seconds_since(t0);

  std::vector<std::int64_t> rows_out(num_threads, 0);
  std::vector<std::uint64_t> checks(num_threads, 0);
  std::vector<std::int64_t> violations(num_threads, 0);
  t0 = Clock::now();
  run_work_orders(static_cast<std::int64_t>(probe_units.size()), num_threads, [&](std::int64_t u, int t) {
    const Unit &unit = probe_units[u];
    const KeyPayloadBlock &probe = l_parts[unit.part].blocks[unit.block];
    const qso_join_table_t *table = tables[unit.part];
    // collect (probe_tid, build_tid) per build block (HashJoinOperator.cpp:76-102) ...
    std::unordered_map<std::uint64_t, std::vector<std::pair<std::int32_t, std::int32_t>>> joined;
    const std::int64_t rows = static_cast<std::int64_t>(probe.key.size());
    for (std::int64_t i = 0; i < rows; ++i) {
      const std::uint64_t hash = hash_int(probe.key[i]);
      std::uint64_t entry = table->st.slots[hash % table->st.num_slots].load(std::memory_order_relaxed);
      while (entry != 0) {
        const Bucket &bucket = table->st.buckets[entry - 1];
        entry = bucket.next.load(std::memory_order_relaxed);
        if (bucket.hash == hash) joined[bucket.value.block].emplace_back(static_cast<std::int32_t>(i), bucket.value.tuple);
      }
    }
    // ... then one output ColumnVector per attribute and build block, bulk-inserted (:494-540): (key, o_payload, l_payload)
    for (const auto &kv : joined) {
      const KeyPayloadBlock &build = o_parts[unit.part].blocks[kv.first];
      const std::size_t m = kv.second.size();
      std::vector<std::int32_t> out_key(m);
      std::vector<std::int64_t> out_o(m), out_l(m);
      for (std::size_t j = 0; j < m; ++j) out_key[j] = probe.key[kv.second[j].first];
      for (std::size_t j = 0; j < m; ++j) out_o[j] = build.payload[kv.second[j].second];
      for (std::size_t j = 0; j < m; ++j) out_l[j] = probe.payload[kv.second[j].first];
      for (std::size_t j = 0; j < m; ++j) {
        checks[t] += static_cast<std::uint64_t>(out_o[j]) * 1000003ULL + static_cast<std::uint64_t>(out_l[j]) +
                     static_cast<std::uint32_t>(out_key[j]);
        violations[t] += build.key[kv.second[j].second] != out_key[j];
      }
      rows_out[t] += static_cast<std::int64_t>(m);
    }
  });
  out->probe_seconds = seconds_since(t0);
  out->output_rows = 0;
  out->checksum = 0;
  out->violations = 0;
  for (int t = 0; t < num_threads; ++t) {
    out->output_rows += rows_out[t];
    out->checksum += checks[t];
    out->violations += violations[t];
  }
  for (int p = 0; p < num_partitions; ++p) qso_join_table_destroy(tables[p]);
}

void qso_bench_q3(const qso_q3_inputs_t *in, int64_t block_rows, int num_threads, qso_q3_result_t *out) {
  // BASELINE config 5 as one reference process runs TPC-H Q3 (benchmarks/tpch/queries/03.sql) with LIP filters attached
  // (query_optimizer/rules/AttachLIPFilters.cpp:131-164: exact bit vectors on custkey and orderkey) and the
  // CollisionFreeVector aggregation on l_orderkey (StarSchemaSimpleCostModel.cpp:614-776).  Every operator is
  // block-at-a-time work orders over num_threads workers; an operator starts when its producer has finished.
  const std::int64_t n_c = in->n_customer, n_o = in->n_orders, n_l = in->n_lineitem;
  const std::int64_t c_words = (in->customers_total + 1 + 63) / 64, o_words = (in->orders_total + 1 + 63) / 64;
  std::unique_ptr<std::atomic<std::uint64_t>[]> lip_c(new std::atomic<std::uint64_t>[c_words]);
  std::unique_ptr<std::atomic<std::uint64_t>[]> lip_o(new std::atomic<std::uint64_t>[o_words]);
  std::unique_ptr<std::atomic<std::uint64_t>[]> exist(new std::atomic<std::uint64_t>[o_words]);
  std::unique_ptr<std::atomic<double>[]> revenue(new std::atomic<double>[in->orders_total + 1]);
  Clock::time_point t_all = Clock::now();
  // InitializeAggregationWorkOrders zero the state in partitions (CollisionFreeVectorTable.hpp:136-143): part of the query
  run_work_orders(num_threads, num_threads, [&](std::int64_t p, int) {
    const std::int64_t per = (in->orders_total + 1 + num_threads - 1) / num_threads;
    const std::int64_t a = p * per, b = std::min<std::int64_t>(in->orders_total + 1, a + per);
    for (std::int64_t i = a; i < b; ++i) revenue[i].store(0.0, std::memory_order_relaxed);
    const std::int64_t wper = (o_words + num_threads - 1) / num_threads;
    for (std::int64_t i = p * wper; i < std::min(o_words, (p + 1) * wper); ++i) {
      lip_o[i].store(0, std::memory_order_relaxed);
      exist[i].store(0, std::memory_order_relaxed);
    }
    const std::int64_t cper = (c_words + num_threads - 1) / num_threads;
    for (std::int64_t i = p * cper; i < std::min(c_words, (p + 1) * cper); ++i) lip_c[i].store(0, std::memory_order_relaxed);
  });

  // customer: BuildHashWorkOrder under the predicate c_mktsegment = 'BUILDING', with its LIPFilterBuilder
  // (BuildHashOperator.cpp:162-207)
  qso_join_table_t *t_c = qso_join_table_create(QSX_INT, std::max<std::int64_t>(1, n_c / 5));
  Clock::time_point t0 = Clock::now();
  run_work_orders((n_c + block_rows - 1) / block_rows, num_threads, [&](std::int64_t b, int) {
    const std::int64_t begin = b * block_rows, rows = std::min(block_rows, n_c - begin);
    std::vector<std::uint64_t> sel(bitmap_words(rows));
    qso_select_cmp(QSX_INT, in->c_mktsegment + begin, rows, QSX_EQ, &in->segment, nullptr, sel.data());
    for (std::int64_t i = 0; i < rows; ++i) {
      if (bit_get(sel.data(), i)) atomic_or_bit(lip_c.get(), static_cast<std::uint64_t>(in->c_custkey[begin + i]));
    }
    t_c->put_block(in->c_custkey + begin, rows, static_cast<std::uint64_t>(b), 0, sel.data());
  });
  out->customer_seconds = seconds_since(t0);

  // orders: HashInnerJoinWorkOrder probing the customer table under o_orderdate < DATE and the LIP filter on o_custkey;
  // output relation (o_orderkey) into temporary blocks (one per work order here)
  struct OrdersOut { std::vector<std::int32_t> orderkey; };
  const std::int64_t o_blocks = (n_o + block_rows - 1) / block_rows;
  std::vector<OrdersOut> o_out(o_blocks);
  t0 = Clock::now();
  run_work_orders(o_blocks, num_threads, [&](std::int64_t b, int) {
    const std::int64_t begin = b * block_rows, rows = std::min(block_rows, n_o - begin);
    std::vector<std::uint64_t> sel(bitmap_words(rows));
    qso_select_cmp(QSX_INT, in->o_orderdate + begin, rows, QSX_LT, &in->date_cut, nullptr, sel.data());
    // LIPFilterAdaptiveProber::filterValueAccessor (utility/lip_filter/LIPFilterAdaptiveProber.hpp:113-228)
    std::vector<std::int32_t> live;
    for (std::int64_t i = 0; i < rows; ++i) {
      if (!bit_get(sel.data(), i)) continue;
      const std::int32_t ck = in->o_custkey[begin + i];
      if (ck < 0 || ck > in->customers_total || !test_bit(lip_c.get(), static_cast<std::uint64_t>(ck))) continue;
      live.push_back(static_cast<std::int32_t>(i));
    }
    std::unordered_map<std::uint64_t, std::vector<std::pair<std::int32_t, std::int32_t>>> joined;
    for (std::int32_t i : live) {
      const std::uint64_t hash = hash_int(in->o_custkey[begin + i]);
      std::uint64_t entry = t_c->st.slots[hash % t_c->st.num_slots].load(std::memory_order_relaxed);
      while (entry != 0) {
        const Bucket &bucket = t_c->st.buckets[entry - 1];
        entry = bucket.next.load(std::memory_order_relaxed);
        if (bucket.hash == hash) joined[bucket.value.block].emplace_back(i, bucket.value.tuple);
      }
    }
    for (const auto &kv : joined) {
      for (const auto &pr : kv.second) o_out[b].orderkey.push_back(in->o_orderkey[begin + pr.first]);
    }
  });
  // BuildHashWorkOrder over the join's output blocks, LIP filter on orderkey built alongside
  std::int64_t qualifying = 0;
  for (const OrdersOut &o : o_out) qualifying += static_cast<std::int64_t>(o.orderkey.size());
  qso_join_table_t *t_o = qso_join_table_create(QSX_INT, std::max<std::int64_t>(1, qualifying));
  run_work_orders(o_blocks, num_threads, [&](std::int64_t b, int) {
    const std::vector<std::int32_t> &keys = o_out[b].orderkey;
    for (std::int32_t k : keys) atomic_or_bit(lip_o.get(), static_cast<std::uint64_t>(k));
    t_o->put_block(keys.data(), static_cast<std::int64_t>(keys.size()), static_cast<std::uint64_t>(b), 0, nullptr);
  });
  out->orders_seconds = seconds_since(t0);

  // lineitem: HashInnerJoinWorkOrder under l_shipdate > DATE and the LIP filter on l_orderkey, output (l_orderkey,
  // l_extendedprice, l_discount) into a temporary block; AggregationWorkOrder on that block: revenue[l_orderkey] +=
  // l_extendedprice * (1 - l_discount) (CollisionFreeVectorTable.hpp:530-645: fetch_or of the existence bit, a
  // compare-exchange loop on the atomic<double>, :640-643)
  std::vector<std::int64_t> pairs(num_threads, 0);
  t0 = Clock::now();
  run_work_orders((n_l + block_rows - 1) / block_rows, num_threads, [&](std::int64_t b, int t) {
    const std::int64_t begin = b * block_rows, rows = std::min(block_rows, n_l - begin);
    std::vector<std::uint64_t> sel(bitmap_words(rows));
    qso_select_cmp(QSX_INT, in->l_shipdate + begin, rows, QSX_GT, &in->date_cut, nullptr, sel.data());
    std::vector<std::int32_t> live;
    for (std::int64_t i = 0; i < rows; ++i) {
      if (!bit_get(sel.data(), i)) continue;
      const std::int32_t ok = in->l_orderkey[begin + i];
      if (ok < 0 || ok > in->orders_total || !test_bit(lip_o.get(), static_cast<std::uint64_t>(ok))) continue;
      live.push_back(static_cast<std::int32_t>(i));
    }
    std::unordered_map<std::uint64_t, std::vector<std::pair<std::int32_t, std::int32_t>>> joined;
    for (std::int32_t i : live) {
      const std::uint64_t hash = hash_int(in->l_orderkey[begin + i]);
      std::uint64_t entry = t_o->st.slots[hash % t_o->st.num_slots].load(std::memory_order_relaxed);
      while (entry != 0) {
        const Bucket &bucket = t_o->st.buckets[entry - 1];
        entry = bucket.next.load(std::memory_order_relaxed);
        if (bucket.hash == hash) joined[bucket.value.block].emplace_back(i, bucket.value.tuple);
      }
    }
    std::vector<std::int32_t> j_key;
    std::vector<double> j_price, j_disc;
    for (const auto &kv : joined) {
      for (const auto &pr : kv.second) j_key.push_back(in->l_orderkey[begin + pr.first]);
      for (const auto &pr : kv.second) j_price.push_back(in->l_extendedprice[begin + pr.first]);
      for (const auto &pr : kv.second) j_disc.push_back(in->l_discount[begin + pr.first]);
    }
    pairs[t] += static_cast<std::int64_t>(j_key.size());
    // ScalarBinaryExpression::getAllValues: one temporary vector per node (expressions/scalar/ScalarBinaryExpression.cpp:100-195)
    std::vector<double> one_minus(j_key.size()), value(j_key.size());
    for (std::size_t j = 0; j < j_key.size(); ++j) one_minus[j] = 1.0 - j_disc[j];
    for (std::size_t j = 0; j < j_key.size(); ++j) value[j] = j_price[j] * one_minus[j];
    for (std::size_t j = 0; j < j_key.size(); ++j) {
      const std::uint64_t loc = static_cast<std::uint64_t>(j_key[j]);
      atomic_or_bit(exist.get(), loc);
      double seen = revenue[loc].load(std::memory_order_relaxed);
      while (!revenue[loc].compare_exchange_weak(seen, seen + value[j], std::memory_order_relaxed)) {}
    }
  });
  out->lineitem_seconds = seconds_since(t0);

  // FinalizeAggregationWorkOrders over key ranges (CollisionFreeVectorTable.hpp:647-727), then ORDER BY revenue DESC LIMIT 10
  t0 = Clock::now();
  std::vector<std::vector<std::pair<double, std::int32_t>>> tops(num_threads);
  std::vector<std::int64_t> groups(num_threads, 0);
  const std::int64_t fin_parts = std::max<std::int64_t>(1, std::min<std::int64_t>((in->orders_total + 1) / 4096, 2 * num_threads));
  run_work_orders(fin_parts, num_threads, [&](std::int64_t p, int t) {
    const std::int64_t len = (in->orders_total + 1 + fin_parts - 1) / fin_parts;
    const std::int64_t a = p * len, b = std::min<std::int64_t>(in->orders_total + 1, a + len);
    std::vector<std::pair<double, std::int32_t>> &top = tops[t];
    for (std::int64_t loc = a; loc < b; ++loc) {
      if (!test_bit(exist.get(), static_cast<std::uint64_t>(loc))) continue;
      ++groups[t];
      top.emplace_back(revenue[loc].load(std::memory_order_relaxed), static_cast<std::int32_t>(loc));
      if (top.size() >= 4096) {
        std::partial_sort(top.begin(), top.begin() + 10, top.end(), std::greater<std::pair<double, std::int32_t>>());
        top.resize(10);
      }
    }
  });
  std::vector<std::pair<double, std::int32_t>> all;
  out->groups = 0;
  for (int t = 0; t < num_threads; ++t) {
    all.insert(all.end(), tops[t].begin(), tops[t].end());
    out->groups += groups[t];
  }
  const std::size_t k = std::min<std::size_t>(10, all.size());
  std::partial_sort(all.begin(), all.begin() + k, all.end(), std::greater<std::pair<double, std::int32_t>>());
  for (std::size_t i = 0; i < 10; ++i) {
    out->top_revenue[i] = i < k ? all[i].first : 0.0;
    out->top_orderkey[i] = i < k ? all[i].second : -1;
  }
  out->finalize_seconds = seconds_since(t0);
  out->total_seconds = seconds_since(t_all);
  out->pairs = 0;
  for (int t = 0; t < num_threads; ++t) out->pairs += pairs[t];
  out->qualifying_orders = qualifying;
  qso_join_table_destroy(t_c);
  qso_join_table_destroy(t_o);
}

double qso_bench_select(int type, const void *col, int64_t n, int op, const void *literal,
                        int64_t block_rows, int num_threads, void *out_col, int64_t *out_rows) {
  // SelectWorkOrder::execute (relational_operators/SelectOperator.cpp:161-195):
  // predicate bitmap per block, then selectSimple into the destination.  The
  // destination here is one contiguous column; each block's rows land at an
  // offset reserved atomically (InsertDestination hands out blocks under a mutex).
  const int w = type_width(type);
  const int64_t blocks = (n + block_rows - 1) / block_rows;
  std::atomic<int64_t> out_pos{0};
  Clock::time_point t0 = Clock::now();
  run_work_orders(blocks, num_threads, [&](int64_t b, int) {
    const int64_t begin = b * block_rows, rows = std::min(block_rows, n - begin);
    const char *c = static_cast<const char *>(col) + begin * w;
    std::vector<uint64_t> bitmap(bitmap_words(rows));
    qso_select_cmp(type, c, rows, op, literal, nullptr, bitmap.data());
    const int64_t cnt = qso_bitmap_count(bitmap.data(), rows);
    const int64_t at = out_pos.fetch_add(cnt);
    qso_compact_gather(w, c, bitmap.data(), rows, static_cast<char *>(out_col) + at * w);
  });
  const double elapsed = seconds_since(t0);
  *out_rows = out_pos.load();
  return elapsed;
}

}  // extern "C"

// ===========================================================================
// compressed attributes
// ===========================================================================
namespace {

template <typename T>
void compress_column_t(int type, const T *values, std::int64_t n, qso_compressed_info_t *info, void *out_codes,
                       void *out_dictionary) {
  const std::size_t width = sizeof(T);
  // CompressionDictionaryBuilder: distinct values, code length in bits grows when the count passes a
  // power of two (compression/CompressionDictionaryBuilder.cpp:128-134), padded to 1/2/4 bytes (.hpp:87-95),
  // dictionary = 2 uint32 header words + the values (.hpp:103-110); values sorted at build time.
  std::vector<T> dict(values, values + n);
  std::sort(dict.begin(), dict.end());
  dict.erase(std::unique(dict.begin(), dict.end()), dict.end());
  unsigned code_bits = 0;
  for (std::size_t num_values = 1; num_values <= dict.size(); ++num_values) {
    if (code_bits == 0 || num_values == (1ull << code_bits) + 1) ++code_bits;
  }
  const std::size_t dict_code_bytes = code_bits < 9 ? 1 : (code_bits < 17 ? 2 : 4);
  const std::size_t dictionary_bytes = 2 * sizeof(std::uint32_t) + dict.size() * width + static_cast<std::size_t>(n) * dict_code_bytes;
  // computeTruncatedByteLengthForAttribute (storage/CompressedBlockBuilder.cpp:508-566): INT/LONG only, no
  // negative value, by the leading zeros of the maximum; LONG maximum == UINT32_MAX is not truncated.
  std::size_t truncated_width = width;
  if ((type == QSX_INT || type == QSX_LONG) && n > 0) {
    bool negative = false;
    std::int64_t mx = 0;
    for (std::int64_t i = 0; i < n; ++i) {
      const std::int64_t v = static_cast<std::int64_t>(values[i]);
      negative = negative || v < 0;
      mx = v > mx ? v : mx;
    }
    if (!negative && !(type == QSX_LONG && mx == 0xFFFFFFFFll)) {
      unsigned needed_bits = 0;
      while (needed_bits < 64 && (static_cast<std::uint64_t>(mx) >> needed_bits) != 0) ++needed_bits;
      if (needed_bits < 9) truncated_width = 1;
      else if (needed_bits < 17) truncated_width = 2;
      else if (needed_bits < 33) truncated_width = 4;
    }
  }
  const std::size_t truncated_bytes = static_cast<std::size_t>(n) * truncated_width;
  // buildCompressionInfo (:590-650): the smaller representation wins, ties go to the dictionary
  if (truncated_bytes < dictionary_bytes) {
    info->kind = truncated_width < width ? 1 : 0;
    info->code_width = static_cast<int>(truncated_width);
    info->num_codes = 0;
    for (std::int64_t i = 0; i < n; ++i) {
      const std::uint64_t v = static_cast<std::uint64_t>(static_cast<std::int64_t>(values[i]));
      switch (truncated_width) {
        case 1: static_cast<std::uint8_t *>(out_codes)[i] = static_cast<std::uint8_t>(v); break;
        case 2: static_cast<std::uint16_t *>(out_codes)[i] = static_cast<std::uint16_t>(v); break;
        default:
          if (truncated_width == 4 && info->kind == 1) static_cast<std::uint32_t *>(out_codes)[i] = static_cast<std::uint32_t>(v);
          else std::memcpy(static_cast<char *>(out_codes) + i * width, &values[i], width);   // uncompressed: as it is
          break;
      }
    }
    return;
  }
  info->kind = 2;
  info->code_width = static_cast<int>(dict_code_bytes);
  info->num_codes = static_cast<std::uint32_t>(dict.size());
  std::memcpy(out_dictionary, dict.data(), dict.size() * width);
  for (std::int64_t i = 0; i < n; ++i) {
    const std::uint32_t code = static_cast<std::uint32_t>(std::lower_bound(dict.begin(), dict.end(), values[i]) - dict.begin());
    switch (dict_code_bytes) {
      case 1: static_cast<std::uint8_t *>(out_codes)[i] = static_cast<std::uint8_t>(code); break;
      case 2: static_cast<std::uint16_t *>(out_codes)[i] = static_cast<std::uint16_t>(code); break;
      default: static_cast<std::uint32_t *>(out_codes)[i] = code; break;
    }
  }
}

constexpr std::uint32_t kU32Max = 0xFFFFFFFFu;

// TransformPredicateOnCompressedAttribute for a literal of the attribute's own type, no NULLs
// (storage/CompressedStoreUtil.cpp:51-140, 425-616).
template <typename T>
void transform_predicate_t(const qso_compressed_info_t &info, const T *dict, int op, T lit, qso_code_predicate_t *out) {
  out->result = 1;  // NONE
  out->comp = QSX_CODE_EQ;
  out->first = out->second = 0;
  auto basic = [&](int comp, std::uint32_t code) { out->result = 2; out->comp = comp; out->first = code; };
  std::pair<std::uint32_t, std::uint32_t> range(0, 0);
  if (info.kind == 2) {
    const T *end = dict + info.num_codes;
    const std::uint32_t lower = static_cast<std::uint32_t>(std::lower_bound(dict, end, lit) - dict);
    const std::uint32_t upper = static_cast<std::uint32_t>(std::upper_bound(dict, end, lit) - dict);
    const bool present = lower != upper;
    if (op == QSX_EQ) {        // TransformEqualPredicate... (:425-470)
      if (present) basic(QSX_CODE_EQ, lower);
      return;
    }
    if (op == QSX_NE) {        // TransformNotEqualPredicate... (:472-535), dictionary without a null code
      if (!present) { out->result = 0; return; }
      basic(QSX_CODE_NE, lower);
      return;
    }
    // getLimitCodesForComparisonTyped (compression/CompressionDictionary.cpp:276-305)
    switch (op) {
      case QSX_LT: range = {0, lower}; break;
      case QSX_LE: range = {0, upper}; break;
      case QSX_GT: range = {upper, info.num_codes}; break;
      default: range = {lower, info.num_codes}; break;
    }
    if (range.first >= range.second) return;                       // NONE
    if (range.second == info.num_codes) range.second = kU32Max;    // skips one comparison (:553-556)
  } else {
    // truncated attribute: TruncationHelper (:144-236) + always-true / always-false tests (:266-420)
    const std::int64_t max_truncated = info.code_width == 4 ? 0xFFFFFFFFll : (1ll << (8 * info.code_width)) - 1;
    const double as_double = static_cast<double>(lit);
    const bool long_exact = std::is_integral<T>::value || as_double == static_cast<double>(static_cast<std::int64_t>(as_double));
    const std::int64_t as_long = static_cast<std::int64_t>(lit);
    const bool in_range = as_long >= 0 && as_long <= max_truncated;
    if (op == QSX_EQ) {
      if (long_exact && in_range) basic(QSX_CODE_EQ, static_cast<std::uint32_t>(as_long));
      return;
    }
    if (op == QSX_NE) {
      if (!long_exact || !in_range) { out->result = 0; return; }
      basic(QSX_CODE_NE, static_cast<std::uint32_t>(as_long));
      return;
    }
    std::int64_t eff = as_long;   // GetEffectiveLiteralForComparison (:208-230): round towards the matching side
    if (!long_exact) eff = (op == QSX_LT || op == QSX_GE) ? static_cast<std::int64_t>(std::ceil(as_double)) : static_cast<std::int64_t>(std::floor(as_double));
    bool always_true = false, always_false = false;
    switch (op) {
      case QSX_LT: always_true = eff > max_truncated; always_false = eff <= 0; break;
      case QSX_LE: always_true = eff >= max_truncated; always_false = eff < 0; break;
      case QSX_GT: always_true = eff < 0; always_false = eff >= max_truncated; break;
      default: always_true = eff <= 0; always_false = eff > max_truncated; break;
    }
    if (always_true) { out->result = 0; return; }
    if (always_false) return;
    switch (op) {
      case QSX_LT: range = {0, static_cast<std::uint32_t>(eff)}; break;
      case QSX_LE: range = {0, static_cast<std::uint32_t>(eff + 1)}; break;
      case QSX_GT: range = {static_cast<std::uint32_t>(eff + 1), kU32Max}; break;
      default: range = {static_cast<std::uint32_t>(eff), kU32Max}; break;
    }
  }
  // :590-612
  if (range.first == 0) {
    if (range.second == kU32Max) out->result = 0;
    else basic(QSX_CODE_LT, range.second);
  } else if (range.second == kU32Max) {
    basic(QSX_CODE_GE, range.first);
  } else {
    out->result = 3;
    out->comp = QSX_CODE_RANGE;
    out->first = range.first;
    out->second = range.second;
  }
}

template <typename C>
void select_codes_t(const C *codes, std::int64_t n, int op, std::uint32_t first, std::uint32_t second,
                    const std::uint64_t *filter, std::uint64_t *out) {
  // getEqualCodes / getNotEqualCodes / getCodesSatisfyingComparison / getCodesInRange
  // (storage/CompressedColumnStoreTupleStorageSubBlock.cpp:420-760); the short-circuit variants only
  // visit the filter's tuples, the others intersect afterwards: same result.
  std::memset(out, 0, sizeof(std::uint64_t) * bitmap_words(n));
  for (std::int64_t i = 0; i < n; ++i) {
    if (filter != nullptr && !bit_get(filter, i)) continue;
    const std::uint32_t c = codes[i];
    bool m;
    switch (op) {
      case QSX_CODE_EQ: m = c == first; break;
      case QSX_CODE_NE: m = c != first; break;
      case QSX_CODE_LT: m = c < first; break;
      case QSX_CODE_GE: m = c >= first; break;
      default: m = c >= first && c < second; break;
    }
    if (m) bit_set(out, i);
  }
}

}  // namespace

extern "C" {

void qso_compress_column(int type, const void *values, int64_t n, qso_compressed_info_t *info, void *out_codes,
                         void *out_dictionary) {
  switch (type) {
    case QSX_INT: compress_column_t(type, static_cast<const std::int32_t *>(values), n, info, out_codes, out_dictionary); break;
    case QSX_LONG: compress_column_t(type, static_cast<const std::int64_t *>(values), n, info, out_codes, out_dictionary); break;
    case QSX_FLOAT: compress_column_t(type, static_cast<const float *>(values), n, info, out_codes, out_dictionary); break;
    default: compress_column_t(type, static_cast<const double *>(values), n, info, out_codes, out_dictionary); break;
  }
}

void qso_transform_predicate(const qso_compressed_info_t *info, int type, const void *dictionary, int op, const void *literal,
                             qso_code_predicate_t *out) {
  switch (type) {
    case QSX_INT: { std::int32_t l; std::memcpy(&l, literal, 4); transform_predicate_t(*info, static_cast<const std::int32_t *>(dictionary), op, l, out); break; }
    case QSX_LONG: { std::int64_t l; std::memcpy(&l, literal, 8); transform_predicate_t(*info, static_cast<const std::int64_t *>(dictionary), op, l, out); break; }
    case QSX_FLOAT: { float l; std::memcpy(&l, literal, 4); transform_predicate_t(*info, static_cast<const float *>(dictionary), op, l, out); break; }
    default: { double l; std::memcpy(&l, literal, 8); transform_predicate_t(*info, static_cast<const double *>(dictionary), op, l, out); break; }
  }
}

void qso_select_codes(int code_width, const void *codes, int64_t n, int op, uint32_t first, uint32_t second, const uint64_t *filter,
                      uint64_t *out_bitmap) {
  switch (code_width) {
    case 1: select_codes_t(static_cast<const std::uint8_t *>(codes), n, op, first, second, filter, out_bitmap); break;
    case 2: select_codes_t(static_cast<const std::uint16_t *>(codes), n, op, first, second, filter, out_bitmap); break;
    default: select_codes_t(static_cast<const std::uint32_t *>(codes), n, op, first, second, filter, out_bitmap); break;
  }
}

void qso_decode_codes(int code_width, const void *codes, int64_t n, const void *dictionary, int value_width, void *out) {
  for (int64_t i = 0; i < n; ++i) {
    std::uint32_t c;
    switch (code_width) {
      case 1: c = static_cast<const std::uint8_t *>(codes)[i]; break;
      case 2: c = static_cast<const std::uint16_t *>(codes)[i]; break;
      default: c = static_cast<const std::uint32_t *>(codes)[i]; break;
    }
    if (dictionary != nullptr) {
      std::memcpy(static_cast<char *>(out) + i * value_width, static_cast<const char *>(dictionary) + static_cast<size_t>(c) * value_width, value_width);
    } else if (value_width == 4) {
      static_cast<std::uint32_t *>(out)[i] = c;
    } else {
      static_cast<std::uint64_t *>(out)[i] = c;
    }
  }
}

}  // extern "C"

// ===========================================================================
// ORDER BY
// ===========================================================================
extern "C" {

// Stable sort of the row numbers by the comparator chain of a SortConfiguration (utility/SortConfiguration.hpp:51-130;
// StorageBlock::sort, storage/StorageBlock.cpp:561-640): key 0 is the most significant, descending[k] flips key k.
void qso_sort_permutation(int nkeys, const void *const *key_cols, const int32_t *key_types, const int32_t *descending, int64_t n,
                          int32_t *out_tids) {
  std::vector<int32_t> order(static_cast<size_t>(n));
  for (int64_t i = 0; i < n; ++i) order[static_cast<size_t>(i)] = static_cast<int32_t>(i);
  auto less = [&](int32_t a, int32_t b) {
    for (int k = 0; k < nkeys; ++k) {
      int cmp = 0;
      switch (key_types[k]) {
        case QSX_INT: { const auto *c = static_cast<const std::int32_t *>(key_cols[k]); cmp = c[a] < c[b] ? -1 : (c[a] > c[b] ? 1 : 0); break; }
        case QSX_LONG: { const auto *c = static_cast<const std::int64_t *>(key_cols[k]); cmp = c[a] < c[b] ? -1 : (c[a] > c[b] ? 1 : 0); break; }
        case QSX_FLOAT: { const auto *c = static_cast<const float *>(key_cols[k]); cmp = c[a] < c[b] ? -1 : (c[a] > c[b] ? 1 : 0); break; }
        case QSX_CHAR: { const auto *c = static_cast<const std::uint8_t *>(key_cols[k]); cmp = c[a] < c[b] ? -1 : (c[a] > c[b] ? 1 : 0); break; }
        case QSX_DATE: { const DateLit x = date_at(key_cols[k], a), y = date_at(key_cols[k], b); cmp = x < y ? -1 : (x > y ? 1 : 0); break; }
        default: { const auto *c = static_cast<const double *>(key_cols[k]); cmp = c[a] < c[b] ? -1 : (c[a] > c[b] ? 1 : 0); break; }
      }
      if (descending != nullptr && descending[k]) cmp = -cmp;
      if (cmp != 0) return cmp < 0;
    }
    return false;
  };
  std::stable_sort(order.begin(), order.end(), less);
  std::memcpy(out_tids, order.data(), sizeof(int32_t) * static_cast<size_t>(n));
}

// The distinctify hash table of a DISTINCT aggregate (AggregationOperationState.cpp:172-207; insert
// AggregationConcreteHandle::insertValueAccessorIntoDistinctifyHashTable, AggregationConcreteHandle.hpp:120-140) keeps
// one entry per distinct (group-by..., argument) tuple.  Restated as a set of tuples: the result is the row number of
// the first occurrence of every distinct tuple, listed in tuple order (the table's own iteration order is unspecified).
int64_t qso_distinct_rows(int ncols, const void *const *cols, const int32_t *types, int64_t n, const uint64_t *filter,
                          int32_t *out_tids) {
  auto cmp_rows = [&](int32_t a, int32_t b) {
    for (int k = 0; k < ncols; ++k) {
      int cmp = 0;
      switch (types[k]) {
        case QSX_INT: { const auto *c = static_cast<const std::int32_t *>(cols[k]); cmp = c[a] < c[b] ? -1 : (c[a] > c[b] ? 1 : 0); break; }
        case QSX_LONG: { const auto *c = static_cast<const std::int64_t *>(cols[k]); cmp = c[a] < c[b] ? -1 : (c[a] > c[b] ? 1 : 0); break; }
        case QSX_FLOAT: { const auto *c = static_cast<const float *>(cols[k]); cmp = c[a] < c[b] ? -1 : (c[a] > c[b] ? 1 : 0); break; }
        case QSX_CHAR: { const auto *c = static_cast<const std::uint8_t *>(cols[k]); cmp = c[a] < c[b] ? -1 : (c[a] > c[b] ? 1 : 0); break; }
        case QSX_DATE: { const DateLit x = date_at(cols[k], a), y = date_at(cols[k], b); cmp = x < y ? -1 : (x > y ? 1 : 0); break; }
        default: { const auto *c = static_cast<const double *>(cols[k]); cmp = c[a] < c[b] ? -1 : (c[a] > c[b] ? 1 : 0); break; }
      }
      if (cmp != 0) return cmp;
    }
    return 0;
  };
  auto less = [&](int32_t a, int32_t b) { return cmp_rows(a, b) < 0; };
  std::set<int32_t, decltype(less)> seen(less);       // keyed by tuple value; holds the first row seen with it
  for (int64_t i = 0; i < n; ++i) {
    if (!row_selected(filter, i)) continue;
    seen.insert(static_cast<int32_t>(i));             // insert() keeps the existing (earlier) row on a duplicate
  }
  int64_t out = 0;
  for (int32_t row : seen) out_tids[out++] = row;
  return out;
}

}  // extern "C"
