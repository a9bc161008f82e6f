// join.hip — K3 (hash-table build) and K4 (probe + match compaction).
//
// Reference loops replaced (paths in the Quickstep tree):
//   K3  storage/HashTable.hpp:1358-1461 (putValueAccessor) ->
//       storage/SimpleScalarSeparateChainingHashTable.hpp:1062-1113
//   K4  storage/HashTable.hpp:2145-2181 (getAllFromValueAccessorImpl) +
//       relational_operators/HashJoinOperator.cpp:76-130 (pair collectors),
//       storage/HashTable.hpp:1979-2062 (existence probes for semi/anti)
//
// Device table (NOT the reference's 32-byte separate-chaining buckets): open
// addressing, power-of-two capacity, load factor <= 1/2, 16-byte probe unit so
// that one global_load_dwordx4 inspects a whole unit:
//   INT keys : unit = two 8-byte entries {key:32 | tid:32}, empty = all ones
//   LONG keys: unit = one 16-byte entry  {key:64, tid:32, pad:32}, empty tid = -1
// Duplicate keys simply occupy further slots of the probe sequence (the
// reference keeps duplicates too, allow_duplicate_keys = true), so the build
// never compares keys: it only claims an empty slot with one CAS.  A probe
// walks units from hash(key) until it meets an empty slot.
//
// Match compaction: a 256-thread workgroup stages the (probe_tid, build_tid)
// pairs of a 4096-row tile in LDS (wave ballot + mbcnt prefix, one LDS atomic
// per wave step), then reserves its output range with ONE global atomic and
// streams the pairs out coalesced.

#include "common.hpp"
#include "join_dense.hpp"
#include "join_lds.hpp"
#include "scan.hpp"

#include <atomic>
#include <cstdlib>
#include <vector>
#include <mutex>
#include <shared_mutex>
#include <type_traits>

namespace qsx {

constexpr int kJBlock = 256;
constexpr uint64_t kEmpty64 = ~0ull;
constexpr uint32_t kEmptyTid = 0xFFFFFFFFu;

struct alignas(16) LongEntry {
  int64_t key;
  uint32_t tid;
  uint32_t pad;
};

// BUCKETS of 16 slots, for both key widths.
//   slots[16 b .. 16 b + 15]: INT keys {tid:32 | key:32} words (one bucket = one 128-byte line, all ones = empty); LONG keys
//     LongEntry {key, tid, pad} (one bucket = two lines, tid all ones = empty);
//   fp[16 b .. 16 b + 15] = one aligned 16-byte word of 1-byte fingerprints (0 = empty slot).
// A key's home bucket comes from the high bits of its hash by multiply-shift (any bucket count: the table is sized for
// load 0.8, not rounded to a power of two), its fingerprint from other bits of the hash.  Slots of a bucket fill in order
// (slot number = what a fetch-add on the bucket's fill counter returned: fill[buckets], behind the fingerprint plane)
// and never empty again, a key that finds its bucket full goes on to the next one — so a probe reads ONE fingerprint word
// per bucket of its sequence (the plane is a tenth / a seventeenth of the table and stays in every XCD's L2), compares
// sixteen bytes in registers, touches the table only where a fingerprint matches (a miss almost never does: 16 / 255 per
// bucket), and stops at the first bucket that still has an empty byte.  1 M INT keys: 10 MiB of slots + 1.25 MiB of
// fingerprints instead of 16 MiB of half-empty units no L2 holds; a probe that misses costs an L2 hit.
constexpr int kBucketSlots = 16;
struct TableView {
  void *slots;    // uint64_t[16 * buckets] (INT) or LongEntry[16 * buckets] (LONG)
  unsigned char *fp;   // fingerprint plane, 16 * buckets bytes
  uint64_t buckets;
  unsigned int *dup_flag;  // set by seal_scan_kernel (the first probe after the builds) when two entries of an INT table share a key
  unsigned int *fill;      // [buckets] slots handed out per bucket (may count past 16: claims that found the bucket full)
  __device__ __host__ uint64_t num_slots() const { return buckets * kBucketSlots; }
};

__device__ __forceinline__ uint64_t home_bucket(int32_t key, const TableView &t) {
  const uint32_t h = static_cast<uint32_t>(key) * 0x9E3779B9u;
  return (static_cast<uint64_t>(h) * t.buckets) >> 32;
}
__device__ __forceinline__ uint32_t fingerprint(int32_t key) {
  const uint32_t f = (static_cast<uint32_t>(key) * 0x85EBCA6Bu) >> 24;
  return f != 0u ? f : 1u;
}
__device__ __forceinline__ uint64_t home_bucket(int64_t key, const TableView &t) {
  const uint64_t h = mix64(static_cast<uint64_t>(key)) * 0x9E3779B97F4A7C15ull;
  return ((h >> 32) * t.buckets) >> 32;
}
__device__ __forceinline__ uint32_t fingerprint(int64_t key) {
  const uint32_t f = static_cast<uint32_t>(mix64(static_cast<uint64_t>(key)) * 0x9E3779B97F4A7C15ull) & 0xFFu;   // (low bits: the bucket took the high ones)
  return f != 0u ? f : 1u;
}
// 0x80 in every byte of x that is zero (exact: nothing carries from one byte into the next)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); }
// the four 0x80 flags of a word as bits 0..3
__device__ __forceinline__ uint32_t flags_to_nibble(uint32_t m) { return (((m >> 7) * 0x00204081u) >> 21) & 0xFu; }
// A bucket's fingerprint word against fingerprint f: bits 0..15 = slots holding f, bits 16..31 = empty slots.
__device__ __forceinline__ uint32_t bucket_masks(const uint4 &w, uint32_t f) {
  const uint32_t ffff = f * 0x01010101u;
  const uint32_t same = flags_to_nibble(zero_bytes(w.x ^ ffff)) | flags_to_nibble(zero_bytes(w.y ^ ffff)) << 4 |
                        flags_to_nibble(zero_bytes(w.z ^ ffff)) << 8 | flags_to_nibble(zero_bytes(w.w ^ ffff)) << 12;
  const uint32_t empty = flags_to_nibble(zero_bytes(w.x)) | flags_to_nibble(zero_bytes(w.y)) << 4 | flags_to_nibble(zero_bytes(w.z)) << 8 |
                         flags_to_nibble(zero_bytes(w.w)) << 12;
  return same | empty << 16;
}

// ---- the COMPACT plane of a sealed bucketed table over INT keys (round 5) ------------------------------------------------------
// What a probe that hits pays for is the slot line: 8-byte slots make 1 M keys a 10 MiB table no L2 holds.  The slot is 8
// bytes because it stores the whole key — but the key's hash h = key * odd is a bijection of 32 bits, the bucket index is
// hi32(h * buckets), and what is left of h inside a bucket, lo32(h * buckets), steps by `buckets` from one h to the next: with
// more than 65 536 buckets id = lo32 * 65024 >> 32 (16 bits, < 65024) is different for every key of a bucket.  The first
// probe after the builds (seal) therefore writes, slot for slot, a second plane: fingerprint byte = (id >> 8) + 1 (1..254;
// 0 = empty, 255 = a key whose home is an earlier bucket), 4-byte slot = {id & 255 : 8, tuple id : 24} — bucket, fingerprint
// and slot together ARE the key.  5 MiB + 1.25 MiB for 1 M keys.  A probe reads its home bucket there; a key that is not in
// its home bucket although the bucket is full was displaced by the build: the probe walks on in the 8-byte table from the
// next bucket, as before.  Only for tables without duplicate keys, tuple ids below 2^24, at least kCompactMinBuckets buckets.
constexpr uint64_t kCompactMinBuckets = 66053;   // buckets * 65024 >= 2^32: ids of one bucket's keys differ
struct CompactView {
  const uint32_t *slots;        // [16 * buckets]
  const unsigned char *fp;      // [16 * buckets]
};
struct CompactId {
  uint32_t bucket, fp, low;
};
__device__ __forceinline__ CompactId compact_id(int32_t key, uint64_t buckets) {
  const uint32_t h = static_cast<uint32_t>(key) * 0x9E3779B9u;   // (home_bucket's hash)
  const uint64_t p = static_cast<uint64_t>(h) * buckets;
  const uint32_t id = static_cast<uint32_t>((static_cast<uint64_t>(static_cast<uint32_t>(p)) * 65024ull) >> 32);
  return CompactId{static_cast<uint32_t>(p >> 32), (id >> 8) + 1u, id & 0xFFu};
}

__device__ __forceinline__ bool row_in_filter(const uint64_t *filter, int64_t row) {
  return filter == nullptr || ((filter[row >> 6] >> (63 - (row & 63))) & 1u);
}

// An insert claims its slot with ONE returning fetch-add on the bucket's fill counter and writes the slot with a plain store —
// nothing is read, and the fingerprint plane is not touched: the first probe after the builds writes it, slot for slot, from
// the keys (seal_scan_kernel: coalesced, 1.25 MB for a million keys, instead of a million scattered byte stores that each
// pull a line into L2).  (Until round 6 an insert read the bucket's fingerprint word at agent scope and claimed the first slot
// that looked empty by compare-and-swap: on this part an agent-scope load is served by the memory side, not by the XCD's L2,
// at 3-4 G loads/s — build_kernel took 238 us per 1 M keys, a sixth of the atomic units' rate.)  What the compare-and-swap
// also did — the later of two inserts of one INT key saw the earlier — is done by the same scan.  Slots and fingerprints are
// only read by kernels launched after the builds (BuildHash -> HashJoin is a pipeline breaker,
// ExecutionGenerator.cpp:1110-1124).
__device__ __forceinline__ void store_entry(const TableView &t, uint64_t slot, int32_t key, uint32_t tid) {
  static_cast<uint64_t *>(t.slots)[slot] = (static_cast<uint64_t>(tid) << 32) | static_cast<uint32_t>(key);
}
__device__ __forceinline__ void store_entry(const TableView &t, uint64_t slot, int64_t key, uint32_t tid) {
  LongEntry e;
  e.key = key;
  e.tid = tid;
  e.pad = 0;
  static_cast<LongEntry *>(t.slots)[slot] = e;
}
// The claim behind fetch-add result `c` on bucket b (c < 16: the slot is this entry's; else the bucket was full: walk on).
template <typename KeyT>
__device__ __forceinline__ void finish_insert(const TableView &t, KeyT key, uint32_t tid, uint64_t b, unsigned int c) {
  while (c >= static_cast<unsigned int>(kBucketSlots)) {
    b = b + 1 == t.buckets ? 0 : b + 1;
    c = atomicAdd(&t.fill[b], 1u);
  }
  store_entry(t, b * kBucketSlots + c, key, tid);
}
template <typename KeyT>
__device__ __forceinline__ void insert_entry(const TableView &t, KeyT key, uint32_t tid) {
  const uint64_t b = home_bucket(key, t);
  finish_insert(t, key, tid, b, atomicAdd(&t.fill[b], 1u));
}

// Launched by the first probe after the builds (plain cached loads: the builds' kernels have finished), one thread per slot:
//  * the slot's fingerprint byte, from its key (0 for an empty slot) — the whole plane is written here;
//  * INT tables: the build side's "some key occurs twice" flag (control word 2).  A thread looks at the entries in FRONT of
//    its own on its key's walk — its home bucket up to its own bucket, there only the slots before its own; every pair of
//    equal keys is seen by the later of the two.  Probes of a table whose flag stays clear stop at their first match.
template <typename KeyT>
__global__ __launch_bounds__(kJBlock) void seal_scan_kernel(TableView t) {
  const uint64_t s = static_cast<uint64_t>(blockIdx.x) * kJBlock + threadIdx.x;
  if (s >= t.num_slots()) return;
  if constexpr (sizeof(KeyT) == 8) {
    const LongEntry e = static_cast<const LongEntry *>(t.slots)[s];
    t.fp[s] = e.tid == kEmptyTid ? static_cast<unsigned char>(0) : static_cast<unsigned char>(fingerprint(e.key));
  } else {
    const uint64_t *slots = static_cast<const uint64_t *>(t.slots);
    const uint64_t e = slots[s];
    if (e == kEmpty64) {
      t.fp[s] = 0;
      return;
    }
    const uint32_t key = static_cast<uint32_t>(e);
    t.fp[s] = static_cast<unsigned char>(fingerprint(static_cast<int32_t>(key)));
    const uint64_t own = s / kBucketSlots;
    uint64_t b = home_bucket(static_cast<int32_t>(key), t);
    for (;;) {   // (the buckets in front of `own` on the walk are full; a bucket's sixteen slots are one line)
      const int end = b == own ? static_cast<int>(s % kBucketSlots) : kBucketSlots;
      for (int j = 0; j < end; ++j) {
        if (static_cast<uint32_t>(slots[b * kBucketSlots + j]) == key) {
          *t.dup_flag = 1u;
          return;
        }
      }
      if (b == own) return;
      b = b + 1 == t.buckets ? 0 : b + 1;
    }
  }
}

// ---------------------------------------------------------------------------
// K3 build
// ---------------------------------------------------------------------------
// The control words of a table come in kControlReplicas copies, each on a cache line of its own: [0] entries, [2] duplicate-key
// flag (hashed) / overflow entries (dense), [3] error flag (dense) — both only in copy 0 —, [4] [5] bounds of the inserted keys:
// max of ~(key + 2^63) and max of (key + 2^63) as unsigned words, all zero = none (one memset clears a table's words).  A
// workgroup adds to copy blockIdx & 31; readers (the host) sum / max over the copies.  One copy made every workgroup's three
// atomics same-address atomics, which complete one at a time device-wide: 977 workgroups cost 44 of a 1 M-key build's 116 us,
// 3 907 workgroups 0.29 ms (tools/hashed_build_exp.py).
constexpr int kControlReplicas = 32;
constexpr int kControlStride = 16;   // words between two copies (128 bytes)
constexpr int kControlWords = 6;
static_assert(kControlReplicas == 32 && kControlStride == 16, "join_dense.hpp's dense_build_kernel picks its copy with the same numbers");
__device__ __forceinline__ unsigned long long *control_replica(unsigned long long *control, unsigned int who) {
  return control + static_cast<size_t>(who & (kControlReplicas - 1)) * kControlStride;
}
struct KeyBounds {
  unsigned long long lo_inv = 0ull, hi = 0ull;
  __device__ __forceinline__ void add(int64_t key) {
    const unsigned long long u = static_cast<unsigned long long>(key) ^ 0x8000000000000000ull;
    lo_inv = ~u > lo_inv ? ~u : lo_inv;
    hi = u > hi ? u : hi;
  }
  // The workgroup's bounds and its count of inserted rows go to its copy of the control words once per workgroup.
  __device__ __forceinline__ void publish(unsigned long long *control, unsigned long long inserted) {
    __shared__ unsigned long long wg[3];   // count, ~min, max
    if (threadIdx.x == 0) wg[0] = wg[1] = wg[2] = 0ull;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long a = __shfl_xor(lo_inv, o, kWave), b = __shfl_xor(hi, o, kWave);
      lo_inv = a > lo_inv ? a : lo_inv;
      hi = b > hi ? b : hi;
    }
    inserted = wave_reduce_add(inserted);
    if (lane_id() == 0 && inserted != 0) {
      atomicAdd(&wg[0], inserted);
      atomicMax(&wg[1], lo_inv);
      atomicMax(&wg[2], hi);
    }
    __syncthreads();
    if (threadIdx.x == 0 && wg[0] != 0) {
      unsigned long long *mine = control_replica(control, blockIdx.x);
      atomicAdd(mine, wg[0]);
      atomicMax(mine + 4, wg[1]);
      atomicMax(mine + 5, wg[2]);
    }
  }
};

template <typename KeyT>
__global__ __launch_bounds__(kJBlock) void build_kernel(TableView t, const KeyT *__restrict__ keys,
                                                       int64_t n, int32_t base_tid,
                                                       const uint64_t *__restrict__ filter,
                                                       unsigned long long *__restrict__ entries) {
  constexpr int R = 4;   // a lane's rows of one round: their fetch-adds are all issued before the first result is used
  unsigned long long inserted = 0;
  KeyBounds bounds;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kJBlock;
  for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * kJBlock + threadIdx.x; i0 < n; i0 += stride * R) {
    KeyT key[R];
    uint64_t b[R];
    unsigned int c[R];
    bool live[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t i = i0 + r * stride;
      live[r] = i < n && row_in_filter(filter, i);
      key[r] = live[r] ? keys[i] : KeyT(0);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      b[r] = home_bucket(key[r], t);
      c[r] = live[r] ? atomicAdd(&t.fill[b[r]], 1u) : 0u;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!live[r]) continue;
      finish_insert(t, key[r], static_cast<uint32_t>(base_tid + i0 + r * stride), b[r], c[r]);
      bounds.add(key[r]);
      ++inserted;
    }
  }
  bounds.publish(entries, inserted);
}
// The build side as a run of blocks (qsx_join_build_blocks): a wave takes groups of kBuildTile rows of ONE block.
template <typename KeyT>
__global__ __launch_bounds__(kJBlock) void build_runs_kernel(TableView t, const long long *__restrict__ runs,
                                                            unsigned long long *__restrict__ entries) {
  unsigned long long inserted = 0;
  KeyBounds bounds;
  const int lane = lane_id();
  const int num_groups = static_cast<int>(runs[2]);
  for (int group = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * (kJBlock / kWave) + (threadIdx.x >> 6)));
       group < num_groups; group += static_cast<int>(gridDim.x) * (kJBlock / kWave)) {
    const RunTile at = run_locate(runs, group);
    const KeyT *keys = run_in<KeyT>(runs, at.block);
    const uint64_t *filter = run_filter(runs, at.block);
    const int64_t n = run_rows(runs, at.block);
    const uint32_t base_tid = static_cast<uint32_t>(run_base(runs, at.block));
    // (a compressed key stripe, block_runs.hpp: read as it lies)
    const bool coded = run_has_coding(runs);
    const ProbeTileSource<KeyT> src{keys, n, 0, 0, filter, nullptr, at.block, coded ? run_code_width(runs, at.block) : 0,
                                    coded ? run_dictionary(runs, at.block) : nullptr};
#pragma unroll 2
    for (int r = 0; r < kBuildR; ++r) {
      const int64_t i = static_cast<int64_t>(at.tile_in_block) * kBuildTile + r * kWave + lane;
      if (i >= n || !row_in_filter(filter, i)) continue;
      const KeyT key = src.code_width != 0 ? coded_key(src, i) : keys[i];
      insert_entry(t, key, base_tid + static_cast<uint32_t>(i));
      bounds.add(key);
      ++inserted;
    }
  }
  bounds.publish(entries, inserted);
}

// Re-insert every entry of an old table into a bigger one (resize).
__global__ __launch_bounds__(kJBlock) void rehash_kernel(int is_long, TableView src, TableView dst) {
  const int64_t cap = static_cast<int64_t>(src.num_slots());
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kJBlock + threadIdx.x; i < cap;
       i += static_cast<int64_t>(gridDim.x) * kJBlock) {
    if (is_long) {
      const LongEntry e = static_cast<const LongEntry *>(src.slots)[i];
      if (e.tid != kEmptyTid) insert_entry(dst, e.key, e.tid);
    } else {
      const uint64_t e = static_cast<const uint64_t *>(src.slots)[i];
      if (e != kEmpty64) {
        insert_entry(dst, static_cast<int32_t>(static_cast<uint32_t>(e)), static_cast<uint32_t>(e >> 32));
      }
    }
  }
}

// The compact plane from the sealed table, slot for slot (no atomics: slot s of the plane describes slot s of the table).
__global__ __launch_bounds__(kJBlock) void compact_build_kernel(TableView t, uint32_t *__restrict__ cslots, unsigned char *__restrict__ cfp) {
  const uint64_t s = static_cast<uint64_t>(blockIdx.x) * kJBlock + threadIdx.x;
  if (s >= t.num_slots()) return;
  unsigned char f = 0;
  uint32_t packed = 0xFFFFFFFFu;
  if (t.fp[s] != 0) {
    const uint64_t e = static_cast<const uint64_t *>(t.slots)[s];
    const CompactId id = compact_id(static_cast<int32_t>(static_cast<uint32_t>(e)), t.buckets);
    if (id.bucket == static_cast<uint32_t>(s / kBucketSlots)) {
      f = static_cast<unsigned char>(id.fp);
      packed = (id.low << 24) | (static_cast<uint32_t>(e >> 32) & 0xFFFFFFu);
    } else {
      f = 0xFFu;     // a key displaced from an earlier bucket: occupies the slot, matches nothing
    }
  }
  cfp[s] = f;
  cslots[s] = packed;
}

// head[] -> the 3-byte copy the probes read (sealed_pack).  Entries beyond 23 bits cannot occur: the host checks the row
// and overflow counts first.
__global__ __launch_bounds__(kJBlock) void dense_pack_kernel(const uint32_t *__restrict__ head, uint64_t range, unsigned char *__restrict__ head3) {
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kJBlock + threadIdx.x; i < range; i += static_cast<uint64_t>(gridDim.x) * kJBlock) {
    const uint32_t w = head[i];
    const uint32_t packed = (w & 0x7FFFFFu) | ((w & kChainBit) >> 8);
    head3[i * 3] = static_cast<unsigned char>(packed);
    head3[i * 3 + 1] = static_cast<unsigned char>(packed >> 8);
    head3[i * 3 + 2] = static_cast<unsigned char>(packed >> 16);
  }
}

// Every entry of a hashed table into a directly addressed one (seal_table): what dense_build_kernel does per build row.
__global__ __launch_bounds__(kJBlock) void dense_build_from_slots_kernel(int is_long, TableView src, DenseTableView d) {
  const int64_t cap = static_cast<int64_t>(src.num_slots());
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kJBlock + threadIdx.x; i < cap;
       i += static_cast<int64_t>(gridDim.x) * kJBlock) {
    int64_t key;
    uint32_t tid;
    if (is_long) {
      const LongEntry e = static_cast<const LongEntry *>(src.slots)[i];
      if (e.tid == kEmptyTid) continue;
      key = e.key;
      tid = e.tid;
    } else {
      const uint64_t e = static_cast<const uint64_t *>(src.slots)[i];
      if (e == kEmpty64) continue;
      key = static_cast<int32_t>(static_cast<uint32_t>(e));
      tid = static_cast<uint32_t>(e >> 32);
    }
    const uint64_t idx = dense_index(d, key);
    if (idx == ~0ull) {
      atomicExch(d.error, 1);
      continue;
    }
    const uint32_t old = atomicCAS(&d.head[idx], 0u, tid + 1u);
    if (old != 0u) {
      const unsigned int e = atomicAdd(d.ov_count, 1u);
      if (e >= d.ov_capacity) {
        atomicExch(d.error, 2);
        continue;
      }
      d.ov[e].x = tid;
      d.ov[e].y = atomicExch(&d.head[idx], kChainBit | e);
    }
  }
}

// ---------------------------------------------------------------------------
// K4 probe.  MODE 0: emit pairs, 1: count only, 2: existence bitmap.
// ---------------------------------------------------------------------------
constexpr int kRowsPerThread = 16;
constexpr int kProbeTile = kJBlock * kRowsPerThread;  // 4096 rows per workgroup tile
constexpr int kStage = kProbeTile;                    // staged pairs per tile (FK joins never exceed it)

struct PairSink {
  int32_t *stage_probe;  // LDS
  int32_t *stage_build;  // LDS
  int *stage_fill;       // LDS
  int32_t *out_probe;    // global
  int32_t *out_build;    // global
  unsigned long long capacity;
  unsigned long long *out_count;
};

// Append one match per matching lane of the wave: ballot, one LDS atomic for
// the whole wave, mbcnt rank.  Pairs that do not fit the LDS stage (tiles with
// many duplicate matches) take the slow path: a wave-aggregated reservation
// straight on the global counter.
__device__ __forceinline__ void emit_match(const PairSink &sink, bool match, int32_t probe_tid,
                                           int32_t build_tid) {
  const uint64_t m = __ballot(match);
  if (m == 0) return;  // wave-uniform
  const int leader = __ffsll(static_cast<long long>(m)) - 1;
  int base = 0;
  if (lane_id() == leader) base = atomicAdd(sink.stage_fill, __popcll(m));
  base = __shfl(base, leader, kWave);
  const int pos = base + rank_below(m);
  const bool over = match && pos >= kStage;
  if (match && !over) {
    sink.stage_probe[pos] = probe_tid;
    sink.stage_build[pos] = build_tid;
  }
  const uint64_t mo = __ballot(over);
  if (mo != 0) {
    const int leader2 = __ffsll(static_cast<long long>(mo)) - 1;
    unsigned long long gbase = 0;
    if (lane_id() == leader2) gbase = atomicAdd(sink.out_count, static_cast<unsigned long long>(__popcll(mo)));
    gbase = __shfl(gbase, leader2, kWave);
    if (over) {
      const unsigned long long o = gbase + rank_below(mo);
      if (o < sink.capacity) {
        sink.out_probe[o] = probe_tid;
        sink.out_build[o] = build_tid;
      }
    }
  }
}

// K4 over the bucketed table (TableView).  MODE 0: emit pairs, 1: count only, 2: existence bitmap.  kRuns: the probe side is a
// run of blocks, as in dense_probe_kernel (join_dense.hpp).
// One slot of either key width as (key, tuple id).
template <typename KeyT>
struct SlotOf;
template <>
struct SlotOf<int32_t> {
  using Raw = uint64_t;
  __device__ static __forceinline__ bool holds(const Raw &e, int32_t key) { return static_cast<uint32_t>(e) == static_cast<uint32_t>(key); }
  __device__ static __forceinline__ int32_t tid(const Raw &e) { return static_cast<int32_t>(e >> 32); }
};
template <>
struct SlotOf<int64_t> {
  using Raw = LongEntry;
  __device__ static __forceinline__ bool holds(const Raw &e, int64_t key) { return e.key == key; }
  __device__ static __forceinline__ int32_t tid(const Raw &e) { return static_cast<int32_t>(e.tid); }
};
//   phase 1  the fingerprint word of every row's home bucket: 16 independent 16-byte reads of the L2-resident plane;
//   phase 2  the slot under the first matching fingerprint: 16 independent 8-byte reads of the table — rows without a
//            matching fingerprint (nearly every probe that misses) read nothing;
//   phase 3  key compare + emit; whatever is left — another slot under the same fingerprint (a false positive, or duplicate
//            build keys), a home bucket without an empty byte (go on in the next bucket) — is walked by a wave-uniform loop.
// kCompact (INT keys): the home bucket is looked up in the table's compact plane (CompactView, above); whatever is not settled
// there — a key displaced by the build — is walked in the 8-byte table from the next bucket on.
template <typename KeyT, int MODE, bool kRuns = false, bool kCompact = false>
__global__ __launch_bounds__(kJBlock) void probe_fp_kernel(
    TableView t, const KeyT *__restrict__ keys, int64_t n, int32_t probe_base_tid,
    const uint64_t *__restrict__ filter, int32_t *__restrict__ out_probe,
    int32_t *__restrict__ out_build, int64_t capacity, unsigned long long *__restrict__ out_count,
    uint64_t *__restrict__ out_bitmap, int anti, const long long *__restrict__ runs = nullptr, CompactView cv = CompactView{}) {
  static_assert(!kCompact || sizeof(KeyT) == 4, "the compact plane is for INT keys");
  using Key = KeyT;
  using Slot = SlotOf<KeyT>;
  using Raw = typename Slot::Raw;
  using Source = ProbeTileSource<Key>;
  __shared__ int32_t s_probe[MODE == 0 ? kStage : 1];
  __shared__ int32_t s_build[MODE == 0 ? kStage : 1];
  __shared__ int s_fill;
  __shared__ unsigned long long s_base;
  const Raw *__restrict__ slots = static_cast<const Raw *>(t.slots);
  const uint4 *__restrict__ fp_words = reinterpret_cast<const uint4 *>(t.fp);
  // INT keys: no build key occurs twice (the build looked at every occupant that could be this key) -> a probe ends at its
  // first match.  LONG keys: the build never compares keys, probes always walk on.
  const bool unique = sizeof(Key) == 4 && *t.dup_flag == 0u;
  PairSink sink{s_probe, s_build, &s_fill, out_probe, out_build, static_cast<unsigned long long>(capacity), out_count};
  const int64_t num_tiles = kRuns ? runs[2] : (n + kProbeTile - 1) / kProbeTile;
  auto source_of = [&](int64_t tile) {
    return probe_tile_source<Key, kProbeTile, kRuns>(runs, tile, keys, n, probe_base_tid, filter, out_bitmap);
  };
  unsigned long long local_count = 0;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  Key key[kRowsPerThread], next_key[kRowsPerThread];
  uint64_t filter_words = ~0ull, next_filter_words = ~0ull;
  auto request = [&](const Source &src, Key (&k)[kRowsPerThread], uint64_t &words) {
    if (kRuns && src.code_width != 0) {   // a compressed key stripe (block_runs.hpp): read as it lies
      coded_keys(src, k, [&](int r) {
        const int64_t row = src.base + r * kJBlock + threadIdx.x;
        return row < src.n ? row : src.n - 1;
      });
    } else {
#pragma unroll
      for (int r = 0; r < kRowsPerThread; ++r) {
        const int64_t row = src.base + r * kJBlock + threadIdx.x;
        k[r] = __builtin_nontemporal_load(&src.keys[row < src.n ? row : src.n - 1]);   // clamped, not guarded; streamed once
      }
    }
    words = ~0ull;
    if (src.filter != nullptr && lane < kRowsPerThread) {
      const int64_t w = (src.base >> 6) + lane * (kJBlock / kWave) + wave;
      if (w < ((src.n + 63) >> 6)) words = src.filter[w];
    }
  };
  Source cur = Source(), next = Source();
  if (static_cast<int64_t>(blockIdx.x) < num_tiles) {
    cur = source_of(blockIdx.x);
    request(cur, key, filter_words);
  }
  for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    if (MODE == 0) {
      if (threadIdx.x == 0) s_fill = 0;
      __syncthreads();
    }
    if (tile + gridDim.x < num_tiles) {
      next = source_of(tile + gridDim.x);
      request(next, next_key, next_filter_words);
    }
    const int64_t tile_base = cur.base;
    const int64_t n_rows = cur.n;
    const int32_t base_tid = cur.base_tid;
    uint64_t *const tile_bitmap = cur.out_bitmap;
    cur = next;
    bool live[kRowsPerThread];
    uint64_t exists_word = 0;
#pragma unroll
    for (int r = 0; r < kRowsPerThread; ++r) {
      const int64_t row = tile_base + r * kJBlock + threadIdx.x;
      const uint64_t filter_word = __shfl(filter_words, r, kWave);   // before the branch: every lane takes part
      live[r] = row < n_rows && msb_bit(filter_word, lane);
    }
    // phase 1: fingerprint words (unconditional reads: dead rows read bucket 0)
    uint32_t bucket[kRowsPerThread], masks[kRowsPerThread];
    {
      uint4 w[kRowsPerThread];
      const uint4 *__restrict__ plane = kCompact ? reinterpret_cast<const uint4 *>(cv.fp) : fp_words;
#pragma unroll
      for (int r = 0; r < kRowsPerThread; ++r) {
        if constexpr (kCompact) bucket[r] = compact_id(key[r], t.buckets).bucket; else bucket[r] = static_cast<uint32_t>(home_bucket(key[r], t));
        w[r] = plane[live[r] ? bucket[r] : 0u];
      }
#pragma unroll
      for (int r = 0; r < kRowsPerThread; ++r) {
        uint32_t f;
        if constexpr (kCompact) f = compact_id(key[r], t.buckets).fp; else f = fingerprint(key[r]);
        masks[r] = live[r] ? bucket_masks(w[r], f) : 0x10000u;   // dead: nothing, "empty seen"
      }
    }
    // phase 2: the slot under the first matching fingerprint
    using First = typename std::conditional<kCompact, uint32_t, Raw>::type;
    First first[kRowsPerThread];
#pragma unroll
    for (int r = 0; r < kRowsPerThread; ++r) {
      const uint32_t same = masks[r] & 0xFFFFu;
      const uint64_t at = same != 0u ? static_cast<uint64_t>(bucket[r]) * kBucketSlots + (__ffs(same) - 1) : 0ull;
      if constexpr (kCompact) first[r] = cv.slots[at]; else first[r] = slots[at];
    }
    // phase 3: the first candidate of every row
    uint32_t walking = 0, found = 0;                    // bit r: row r of this thread
    uint32_t in_plane = kCompact ? 0xFFFFu : 0u;        // bit r: still among the home bucket's entries of the compact plane
    auto tid_of_row = [&](int r) { return static_cast<int32_t>(base_tid + tile_base + r * kJBlock + threadIdx.x); };
#pragma unroll
    for (int r = 0; r < kRowsPerThread; ++r) {
      uint32_t same = masks[r] & 0xFFFFu;
      const uint32_t empty = masks[r] >> 16;
      bool hit;
      int32_t hit_tid;
      if constexpr (kCompact) {
        hit = same != 0u && (first[r] >> 24) == compact_id(key[r], t.buckets).low;
        hit_tid = static_cast<int32_t>(first[r] & 0xFFFFFFu);
      } else {
        hit = same != 0u && Slot::holds(first[r], key[r]);
        hit_tid = Slot::tid(first[r]);
      }
      if (MODE == 0) emit_match(sink, hit, tid_of_row(r), hit_tid);
      if (MODE == 1) local_count += hit ? 1u : 0u;
      found |= hit ? (1u << r) : 0u;
      same &= same - 1u;
      masks[r] = same | (empty << 16);
      // done: one match of a duplicate-free table (or any match of an existence probe), or nothing else under this
      // fingerprint and the bucket still has room (its sequence ends here)
      const bool more = live[r] && !(hit && (unique || MODE == 2)) && !(same == 0u && empty != 0u);
      walking |= more ? (1u << r) : 0u;
    }
    // Whatever is left — another slot under the same fingerprint (a false positive, or duplicate build keys), a bucket without
    // an empty byte (go on in the next bucket; a compact plane: a key the build displaced) — in ROUNDS over all rows of the
    // thread: the fingerprint words of every row that moves on are read together, then the next candidate slots of every row.
    // (Row by row — a wave-uniform loop per row — nearly every wave walked at least once per row, for one lane in twenty, and
    // waited out two dependent reads each time: 16 x 2 round trips per tile instead of ~2 x 2.)
    while (__any(walking != 0u)) {
      uint32_t advance = 0;    // rows whose bucket is used up (and was full)
#pragma unroll
      for (int r = 0; r < kRowsPerThread; ++r) advance |= (((walking >> r) & 1u) != 0u && (masks[r] & 0xFFFFu) == 0u) ? (1u << r) : 0u;
#pragma unroll
      for (int half = 0; half < kRowsPerThread; half += 8) {
        uint4 w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = half + j;
          const bool adv = ((advance >> r) & 1u) != 0u;
          if (adv) bucket[r] = bucket[r] + 1u == static_cast<uint32_t>(t.buckets) ? 0u : bucket[r] + 1u;
          w[j] = fp_words[adv ? bucket[r] : 0u];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = half + j;
          if (((advance >> r) & 1u) != 0u) masks[r] = bucket_masks(w[j], fingerprint(key[r]));
        }
      }
      in_plane &= ~advance;    // (the next bucket is the 8-byte table's)
#pragma unroll
      for (int half = 0; half < kRowsPerThread; half += 8) {
        Raw e[8];
        uint32_t e4[kCompact ? 8 : 1];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = half + j;
          const uint32_t same = masks[r] & 0xFFFFu;
          const bool have = ((walking >> r) & 1u) != 0u && same != 0u;
          const bool plane_row = kCompact && ((in_plane >> r) & 1u) != 0u;
          const uint64_t at = have ? static_cast<uint64_t>(bucket[r]) * kBucketSlots + (__ffs(same) - 1) : 0ull;
          e[j] = slots[plane_row ? 0ull : at];
          if constexpr (kCompact) e4[j] = cv.slots[plane_row ? at : 0ull];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = half + j;
          uint32_t same = masks[r] & 0xFFFFu;
          const uint32_t empty = masks[r] >> 16;
          const bool is_walking = ((walking >> r) & 1u) != 0u;
          const bool have = is_walking && same != 0u;
          bool hit;
          int32_t hit_tid;
          if (kCompact && ((in_plane >> r) & 1u) != 0u) {
            hit = have && (e4[kCompact ? j : 0] >> 24) == compact_id(key[r], t.buckets).low;
            hit_tid = static_cast<int32_t>(e4[kCompact ? j : 0] & 0xFFFFFFu);
          } else {
            hit = have && Slot::holds(e[j], key[r]);
            hit_tid = Slot::tid(e[j]);
          }
          if (have) same &= same - 1u;
          masks[r] = same | (empty << 16);
          found |= hit ? (1u << r) : 0u;
          if (MODE == 0) emit_match(sink, hit, tid_of_row(r), hit_tid);
          if (MODE == 1) local_count += hit ? 1u : 0u;
          const bool done = (hit && (unique || MODE == 2)) || (same == 0u && empty != 0u);
          if (is_walking && done) walking &= ~(1u << r);
        }
      }
    }
    if (MODE == 2) {
#pragma unroll
      for (int r = 0; r < kRowsPerThread; ++r) {
        const bool bit = live[r] && ((((found >> r) & 1u) != 0u) != (anti != 0));
        const uint64_t word = msb_first(__ballot(bit));
        if (lane == r) exists_word = word;
        if (lane == 0) local_count += __popcll(word);
      }
    }
    if (MODE == 2) {   // lane r holds the word of step r: one store instruction per tile and wave
      const int64_t w = (tile_base >> 6) + lane * (kJBlock / kWave) + wave;
      if (lane < kRowsPerThread && w < ((n_rows + 63) >> 6)) tile_bitmap[w] = exists_word;
    }
#pragma unroll
    for (int r = 0; r < kRowsPerThread; ++r) key[r] = next_key[r];
    filter_words = next_filter_words;
    if (MODE == 0) {
      __syncthreads();
      const int produced = s_fill;
      const int staged = produced < kStage ? produced : kStage;
      if (threadIdx.x == 0) s_base = atomicAdd(out_count, static_cast<unsigned long long>(staged));
      __syncthreads();
      const unsigned long long base = s_base;
      for (int i = threadIdx.x; i < staged; i += kJBlock) {
        const unsigned long long o = base + i;
        if (o < sink.capacity) {
          __builtin_nontemporal_store(s_probe[i], &out_probe[o]);
          __builtin_nontemporal_store(s_build[i], &out_build[o]);
        }
      }
      __syncthreads();
    }
  }
  if (MODE != 0) {
    local_count = wave_reduce_add(local_count);
    if (lane_id() == 0 && local_count != 0 && out_count != nullptr) atomicAdd(out_count, local_count);
  }
}

}  // namespace qsx
#include "join_lds_bucket.hpp"
namespace qsx {

// ---------------------------------------------------------------------------
// composite keys -> one LONG key (qsx_join_key_pack)
// ---------------------------------------------------------------------------
struct KeyPackArgs {
  const void *cols[QSX_MAX_KEYS];
  int is_long[QSX_MAX_KEYS];
  int shift[QSX_MAX_KEYS];  // exact packing: bit offset of the component
  int ncols;
  int exact;
};

// utility/HashPair.hpp:47-58 (64-bit CombineHashes)
__device__ __forceinline__ uint64_t combine_hashes(uint64_t first, uint64_t second) {
  const uint64_t kMul = 0x9ddfea08eb382d69ull;
  uint64_t a = (first ^ second) * kMul;
  a ^= (a >> 47);
  uint64_t b = (second ^ a) * kMul;
  b ^= (b >> 47);
  b *= kMul;
  return b;
}

__global__ __launch_bounds__(kJBlock) void key_pack_kernel(KeyPackArgs a, int64_t n, int64_t *__restrict__ out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kJBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kJBlock) {
    uint64_t acc = 0;
    for (int k = 0; k < a.ncols; ++k) {
      // identity hash = the zero-extended bit pattern (types/TypedValue.hpp:575-592)
      const uint64_t v = a.is_long[k] ? static_cast<uint64_t>(static_cast<const int64_t *>(a.cols[k])[i])
                                      : static_cast<uint64_t>(static_cast<const uint32_t *>(a.cols[k])[i]);
      if (a.exact) acc |= v << a.shift[k];
      else acc = k == 0 ? v : combine_hashes(acc, v);
    }
    out[i] = static_cast<int64_t>(acc);
  }
}

// A CHAR(n <= 8) join key as a LONG: the bytes up to the first NUL (AsciiStringComparators.hpp:218-251: a value ends
// there), zero behind it — equal strings give equal keys whatever lies behind the terminator in the stripe.
__global__ __launch_bounds__(kJBlock) void key_pack_char_kernel(const unsigned char *__restrict__ col, int width, int64_t n,
                                                               int64_t *__restrict__ out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kJBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kJBlock) {
    uint64_t key = 0;
    bool ended = false;
    for (int b = 0; b < width; ++b) {
      const unsigned char c = col[i * width + b];
      ended = ended || c == 0;
      key |= static_cast<uint64_t>(ended ? 0 : c) << (8 * b);
    }
    out[i] = static_cast<int64_t>(key);
  }
}

// The composite keys of a run of blocks -> ONE stripe of packed keys, block after block (qsx_join_key_pack_blocks): a wave
// takes tiles of 512 rows of one block; the blocks' key columns follow the run table at word cols_offset
// ([block * ncols + component]), the table's `base` is the block's first row in the output.
// coded != 0: behind the column pointers lie, per (block, component), a code width (0 = values) and a dictionary address — the
// component of a block that holds it compressed is read as it lies (block_runs.hpp "Coded key stripes").
__global__ __launch_bounds__(kJBlock) void key_pack_runs_kernel(KeyPackArgs a, const long long *__restrict__ runs, long long cols_offset,
                                                               int64_t *__restrict__ out, int coded) {
  constexpr int kTileRows = 512;
  const int lane = lane_id();
  const int num_tiles = static_cast<int>(runs[2]);
  for (int tile = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * (kJBlock / kWave) + (threadIdx.x >> 6))); tile < num_tiles;
       tile += static_cast<int>(gridDim.x) * (kJBlock / kWave)) {
    const RunTile at = run_locate(runs, tile);
    const long long n = run_rows(runs, at.block);
    const long long out_first = run_base(runs, at.block);
    const long long *cols = runs + cols_offset + static_cast<long long>(at.block) * a.ncols;
    const long long total_entries = runs[0] * a.ncols;
    const long long *widths = cols + total_entries, *dicts = widths + total_entries;   // (only read when coded)
    for (int r = lane; r < kTileRows; r += kWave) {
      const long long i = static_cast<long long>(at.tile_in_block) * kTileRows + r;
      if (i >= n) break;
      uint64_t acc = 0;
      for (int k = 0; k < a.ncols; ++k) {
        const void *col = as_global(reinterpret_cast<const void *>(cols[k]));
        const int code_width = coded != 0 ? static_cast<int>(widths[k]) : 0;
        uint64_t v;
        if (code_width == 0) {
          v = a.is_long[k] ? static_cast<uint64_t>(static_cast<const int64_t *>(col)[i]) : static_cast<uint64_t>(static_cast<const uint32_t *>(col)[i]);
        } else {
          const uint32_t code = code_width == 1 ? static_cast<const uint8_t *>(col)[i]
                                                : (code_width == 2 ? static_cast<const uint16_t *>(col)[i] : static_cast<const uint32_t *>(col)[i]);
          const void *dict = as_global(reinterpret_cast<const void *>(dicts[k]));
          if (dict == nullptr) v = code;   // a truncated value (never negative): as wide as it gets
          else v = a.is_long[k] ? static_cast<uint64_t>(static_cast<const int64_t *>(dict)[code]) : static_cast<uint64_t>(static_cast<const uint32_t *>(dict)[code]);
        }
        if (a.exact) acc |= v << a.shift[k];
        else acc = k == 0 ? v : combine_hashes(acc, v);
      }
      out[out_first + i] = static_cast<int64_t>(acc);
    }
  }
}

}  // namespace qsx

using namespace qsx;

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct qsx_join_table {
  int key_type = QSX_INT;
  uint64_t capacity = 0;  // slots: 16 x buckets (load <= 0.8), the fingerprint plane behind them
  void *slots = nullptr;
  unsigned long long *entries_dev = nullptr;
  int64_t reserved = 0;   // host-side upper bound of entries (rows handed to build so far)
  std::atomic<int64_t> max_tid{-1};   // host-side upper bound of the tuple ids stored so far (base_tid + rows of every build)
  // Builds/probes take it shared, growth takes it exclusive — the role of
  // HashTable::resize_shared_mutex_ (storage/HashTable.hpp:1215).
  std::shared_mutex mutex;


  // Hashed flavour: a directly addressed shadow of the entries, made by the first probe after the builds when the keys
  // turned out to span a small range (seal_table).  0 = not looked at since the last build / clear, 1 = stays hashed,
  // 2 = `shadow` answers the probes.
  // 3 = stays hashed AND `compact` (the 4-byte plane of the slots, CompactView) is valid.
  std::mutex seal_mutex;
  std::atomic<int> seal_state{0};
  // What the first probe waits for instead of the whole device: one event per stream that cleared or built this (hashed)
  // table, re-recorded behind every such call (mark_stream); the probing stream is made to wait for them, reads the control
  // words into control_host behind them and only that stream is synchronised.  seal_event / seal_stream: recorded behind the
  // kernels that make the shadow or the compact plane — probes of OTHER streams wait for it (like pack_event).
  std::atomic<bool> scanned{false};   // seal_scan_kernel has been issued since the last build (seal_event orders probes behind it)
  std::mutex marks_mutex;
  std::vector<std::pair<hipStream_t, hipEvent_t>> marks;
  unsigned long long *control_host = nullptr;   // pinned, kControlBytes
  hipEvent_t seal_event = nullptr;
  hipStream_t seal_stream = nullptr;
  void *compact = nullptr;          // [capacity] 4-byte slots, then [capacity] fingerprint bytes
  uint64_t compact_capacity = 0;
  CompactView compact_view() const {
    return CompactView{static_cast<const uint32_t *>(compact), static_cast<const unsigned char *>(compact) + compact_capacity * 4};
  }
  // Covering array of the projection last asked for (join_dense.hpp ProjectionView::cover; directly addressed tables):
  // cover_sig = what it was built for (column widths, build stripes, segment starts); cover_state 0 = none, 1 = this
  // projection cannot have one (duplicate keys, entries wider than 16 bytes, an ambiguous entry), 2 = valid.  A build or
  // clear drops it.  Probes of other streams wait for cover_event.
  std::mutex cover_mutex;
  std::vector<long long> cover_sig;
  void *cover = nullptr;
  size_t cover_bytes = 0;
  int cover_entry_bytes = 0;
  int cover_state = 0;
  hipEvent_t cover_event = nullptr;
  hipStream_t cover_stream = nullptr;
  qsx_join_table *shadow = nullptr;

  // Directly addressed flavour (join_dense.hpp): head[] + overflow chain entries; `slots` unused.
  bool dense = false;
  int64_t min_key = 0;
  int stride_shift = 0;
  uint64_t range = 0;
  uint32_t *head = nullptr;
  unsigned char *head3 = nullptr;   // the probes' 3-byte copy of head[] (sealed_pack); valid while seal_state == 2
  hipEvent_t pack_event = nullptr;  // recorded behind the pack kernel on pack_stream
  hipStream_t pack_stream = nullptr;
  uint2 *ov = nullptr;
  unsigned int ov_capacity = 0;
  DenseTableView dense_view() const {
    DenseTableView v;
    v.head = head;
    v.head3 = seal_state.load(std::memory_order_acquire) == 2 ? head3 : nullptr;
    v.ov = ov;
    v.min_key = min_key;
    v.stride_shift = stride_shift;
    v.range = range;
    v.ov_count = reinterpret_cast<unsigned int *>(entries_dev + 2);
    v.ov_capacity = ov_capacity;
    v.error = reinterpret_cast<int *>(entries_dev + 3);
    return v;
  }

  size_t entry_bytes() const { return key_type == QSX_INT ? 8 : 16; }
  // bytes of the table: slots, then one fingerprint byte per slot
  size_t table_bytes() const { return capacity * entry_bytes() + capacity; }
  // ... and behind them the buckets' fill counters (only the builds touch them)
  static size_t alloc_bytes(size_t entry, uint64_t cap) { return cap * entry + cap + cap / kBucketSlots * sizeof(unsigned int); }
  // rows the table takes at its load limit
  uint64_t room() const { return capacity / 5 * 4; }
  TableView view() const {
    TableView v{};
    v.slots = slots;
    v.dup_flag = reinterpret_cast<unsigned int *>(entries_dev + 2);
    v.fp = static_cast<unsigned char *>(slots) + capacity * entry_bytes();
    v.buckets = capacity / kBucketSlots;
    v.fill = reinterpret_cast<unsigned int *>(v.fp + capacity);
    return v;
  }
};
static void drop_cover(qsx_join_table *t);

// control words behind entries_dev: kControlReplicas copies (see KeyBounds).
constexpr size_t kControlBytes = static_cast<size_t>(kControlReplicas) * kControlStride * sizeof(unsigned long long);
static int reset_control_words(unsigned long long *control, hipStream_t stream) {
  QSX_HIP_TRY(hipMemsetAsync(control, 0, kControlBytes, stream));
  return QSX_OK;
}
// The copies folded into one set of words: v[0] entries, v[2], v[3] as they are, v[4] = min key + 2^63, v[5] = max key + 2^63
// (unsigned images); returns false when no key was ever inserted.
static bool fold_control_words(const unsigned long long *copies, unsigned long long *v) {
  unsigned long long entries = 0, lo_inv = 0, hi = 0;
  for (int r = 0; r < kControlReplicas; ++r) {
    const unsigned long long *c = copies + static_cast<size_t>(r) * kControlStride;
    entries += c[0];
    lo_inv = c[4] > lo_inv ? c[4] : lo_inv;
    hi = c[5] > hi ? c[5] : hi;
  }
  v[0] = entries;
  v[1] = 0;
  v[2] = copies[2];
  v[3] = copies[3];
  v[4] = ~lo_inv;
  v[5] = hi;
  return !(lo_inv == 0 && hi == 0);
}
// (host waits for `stream`, or for the null stream's ordering when the copy is synchronous)
static int copy_control_words(const unsigned long long *control_dev, hipStream_t stream, unsigned long long *v, bool *any_key = nullptr) {
  unsigned long long copies[kControlReplicas * kControlStride];
  QSX_HIP_TRY(hipMemcpyAsync(copies, control_dev, kControlBytes, hipMemcpyDeviceToHost, stream));
  QSX_HIP_TRY(hipStreamSynchronize(stream));
  const bool any = fold_control_words(copies, v);
  if (any_key != nullptr) *any_key = any;
  return QSX_OK;
}

// Behind a clear or build of a hashed table on `stream` (see qsx_join_table::marks).  A failure to record only costs the
// first probe its shortcut: it then waits for the device (sealed_shadow).
static void mark_stream(qsx_join_table *t, hipStream_t stream) {
  if (t->dense) return;
  std::lock_guard<std::mutex> lock(t->marks_mutex);
  for (auto &m : t->marks) {
    if (m.first == stream) {
      if (m.second != nullptr && hipEventRecord(m.second, stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipEventDestroy(m.second);
        m.second = nullptr;
      }
      return;
    }
  }
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess || hipEventRecord(e, stream) != hipSuccess) {
    (void)hipGetLastError();
    if (e != nullptr) (void)hipEventDestroy(e);
    e = nullptr;
  }
  t->marks.emplace_back(stream, e);
}

static uint64_t capacity_for(int key_type, int64_t entries) {
  (void)key_type;
  const uint64_t rows = static_cast<uint64_t>(entries < 512 ? 512 : entries);
  // buckets of 16 slots at load 0.8 (12.8 rows per bucket), any bucket count — the reference sizes for load 0.5
  // (kHashTableLoadFactor = 2 slots per entry, storage/StorageConstants.hpp:104); a fingerprint probe does not need the room
  return (rows * 5 + 63) / 64 * kBucketSlots + kBucketSlots;
}

static int fill_empty(const qsx_join_table *t, void *slots, uint64_t capacity, hipStream_t stream) {
  QSX_HIP_TRY(hipMemsetAsync(slots, 0xFF, capacity * t->entry_bytes(), stream));
  // (the fingerprint plane is written whole by seal_scan_kernel in front of the first probe: only the fill counters behind it)
  QSX_HIP_TRY(hipMemsetAsync(static_cast<char *>(slots) + capacity * t->entry_bytes() + capacity, 0, capacity / kBucketSlots * sizeof(unsigned int), stream));
  return QSX_OK;
}

static int allocate_slots(qsx_join_table *t, uint64_t capacity, void **out) {
  QSX_HIP_TRY(device_malloc(out, qsx_join_table::alloc_bytes(t->entry_bytes(), capacity)));
  int rc = fill_empty(t, *out, capacity, nullptr);
  if (rc == QSX_OK) {
    const hipError_t waited = hipStreamSynchronize(nullptr);
    if (waited != hipSuccess) {
      set_last_error("hipStreamSynchronize(nullptr)", waited);
      rc = QSX_ERR_HIP;
    }
  }
  if (rc != QSX_OK) {        // the callers drop *out on failure: give the allocation back here
    (void)device_free(*out);
    *out = nullptr;
  }
  return rc;
}

// Make room for `additional` more rows.  Counterpart of HashTable::resize
// (storage/HashTable.hpp:1437-1440, SimpleScalarSeparateChainingHashTable.hpp:820-985).
// Dense flavour: every row could be a duplicate, so the overflow list has room for every row handed in.
static int ensure_room_dense(qsx_join_table *t, int64_t additional) {
  std::unique_lock<std::shared_mutex> lock(t->mutex);
  if (static_cast<uint64_t>(t->reserved + additional) <= t->ov_capacity) {
    t->reserved += additional;
    return QSX_OK;
  }
  QSX_HIP_TRY(hipDeviceSynchronize());  // drain in-flight builds on every stream
  uint64_t want = static_cast<uint64_t>(t->reserved + additional) * 2;
  if (want > 0x7FFFFFFFull) want = 0x7FFFFFFFull;
  if (want < static_cast<uint64_t>(t->reserved + additional)) return QSX_ERR_CAPACITY;
  uint2 *bigger = nullptr;
  QSX_HIP_TRY(device_malloc(reinterpret_cast<void **>(&bigger), want * sizeof(uint2)));
  if (t->ov_capacity != 0) {
    QSX_HIP_TRY(hipMemcpy(bigger, t->ov, static_cast<size_t>(t->ov_capacity) * sizeof(uint2), hipMemcpyDeviceToDevice));
  }
  (void)device_free(t->ov);
  t->ov = bigger;
  t->ov_capacity = static_cast<unsigned int>(want);
  t->reserved += additional;
  return QSX_OK;
}

static int ensure_room(qsx_join_table *t, int64_t additional) {
  if (t->dense) return ensure_room_dense(t, additional);
  std::unique_lock<std::shared_mutex> lock(t->mutex);
  if (static_cast<uint64_t>(t->reserved + additional) <= t->room()) {
    t->reserved += additional;
    return QSX_OK;
  }
  QSX_HIP_TRY(hipDeviceSynchronize());  // drain in-flight builds on every stream
  unsigned long long words[kControlWords];
  const int rc_words = copy_control_words(t->entries_dev, nullptr, words);
  if (rc_words != QSX_OK) return rc_words;
  const unsigned long long actual = words[0];
  // rows filtered out by a bitmap never became entries: tighten the bound
  t->reserved = static_cast<int64_t>(actual);
  if (static_cast<uint64_t>(t->reserved + additional) <= t->room()) {
    t->reserved += additional;
    return QSX_OK;
  }
  const uint64_t new_capacity = capacity_for(t->key_type, 2 * (t->reserved + additional));
  void *bigger = nullptr;
  int rc = allocate_slots(t, new_capacity, &bigger);
  if (rc != QSX_OK) return rc;
  TableView src = t->view();
  void *old_slots = t->slots;
  t->slots = bigger;
  t->capacity = new_capacity;
  TableView dst = t->view();
  hipLaunchKernelGGL(rehash_kernel, dim3(grid_for(static_cast<int64_t>(src.num_slots()), kJBlock)), dim3(kJBlock), 0, nullptr,
                     t->key_type == QSX_LONG ? 1 : 0, src, dst);
  QSX_CHECK_LAUNCH();
  QSX_HIP_TRY(hipDeviceSynchronize());
  QSX_HIP_TRY(device_free(old_slots));
  t->reserved += additional;
  return QSX_OK;
}

extern "C" {

int qsx_join_table_create(int key_type, int64_t est_entries, qsx_join_table_t **out) {
  QSX_REQUIRE_DEVICE();
  if (out == nullptr || est_entries < 0) return QSX_ERR_INVALID_ARGUMENT;
  if (key_type != QSX_INT && key_type != QSX_LONG) return QSX_ERR_UNSUPPORTED;
  qsx_join_table *t = new qsx_join_table();
  t->key_type = key_type;
  t->capacity = capacity_for(key_type, est_entries);
  int rc = allocate_slots(t, t->capacity, &t->slots);
  if (rc != QSX_OK) { delete t; return rc; }
  hipError_t err = device_malloc(reinterpret_cast<void **>(&t->entries_dev), kControlBytes);
  if (err == hipSuccess) err = hipMemset(t->entries_dev, 0, kControlBytes);
  if (err != hipSuccess) {
    set_last_error("device_malloc(entries)", err);
    (void)device_free(t->slots);
    delete t;
    return QSX_ERR_HIP;
  }
  *out = t;
  return QSX_OK;
}

int qsx_join_table_create_dense(int key_type, int64_t min_key, int64_t max_key, int64_t key_stride,
                                int64_t est_entries, qsx_join_table_t **out) {
  QSX_REQUIRE_DEVICE();
  if (out == nullptr || est_entries < 0 || max_key < min_key || key_stride < 1) return QSX_ERR_INVALID_ARGUMENT;
  if (key_type != QSX_INT && key_type != QSX_LONG) return QSX_ERR_UNSUPPORTED;
  if ((key_stride & (key_stride - 1)) != 0) return QSX_ERR_UNSUPPORTED;  // shift addressing only
  int stride_shift = 0;
  while ((1ll << stride_shift) < key_stride) ++stride_shift;
  // unsigned: the span of [INT64_MIN, INT64_MAX] still fits
  const uint64_t span = static_cast<uint64_t>(max_key) - static_cast<uint64_t>(min_key);
  const uint64_t range = (span >> stride_shift) + 1;
  if (range > (1ull << 32)) return QSX_ERR_CAPACITY;  // 16 GiB of head words
  qsx_join_table *t = new qsx_join_table();
  t->key_type = key_type;
  t->dense = true;
  t->min_key = min_key;
  t->stride_shift = stride_shift;
  t->range = range;
  t->ov_capacity = static_cast<unsigned int>(est_entries < 1024 ? 1024 : (est_entries > 0x7FFFFFFF ? 0x7FFFFFFF : est_entries));
  hipError_t err = device_malloc(reinterpret_cast<void **>(&t->head), range * sizeof(uint32_t));
  if (err == hipSuccess) err = hipMemset(t->head, 0, range * sizeof(uint32_t));
  if (err == hipSuccess) err = device_malloc(reinterpret_cast<void **>(&t->ov), static_cast<size_t>(t->ov_capacity) * sizeof(uint2));
  if (err == hipSuccess) err = device_malloc(reinterpret_cast<void **>(&t->entries_dev), kControlBytes);
  if (err == hipSuccess) err = hipMemset(t->entries_dev, 0, kControlBytes);
  if (err != hipSuccess) {
    set_last_error("device_malloc(dense join table)", err);
    (void)device_free(t->head);
    (void)device_free(t->ov);
    (void)device_free(t->entries_dev);
    delete t;
    return err == hipErrorOutOfMemory ? QSX_ERR_OUT_OF_MEMORY : QSX_ERR_HIP;
  }
  *out = t;
  return QSX_OK;
}

int qsx_join_key_pack(int ncols, const void *const *cols, const int32_t *types, int64_t n, int64_t *out_dev,
                      int *out_exact, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (ncols < 1 || ncols > QSX_MAX_KEYS || cols == nullptr || types == nullptr || n < 0 || out_exact == nullptr ||
      (n > 0 && out_dev == nullptr)) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  KeyPackArgs a;
  a.ncols = ncols;
  int bits = 0;
  for (int k = 0; k < ncols; ++k) {
    if (types[k] != QSX_INT && types[k] != QSX_LONG) return QSX_ERR_UNSUPPORTED;
    if (n > 0 && cols[k] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
    a.cols[k] = cols[k];
    a.is_long[k] = types[k] == QSX_LONG;
    a.shift[k] = bits < 64 ? bits : 0;
    bits += types[k] == QSX_LONG ? 64 : 32;
  }
  a.exact = bits <= 64;
  *out_exact = a.exact;
  if (n == 0) return QSX_OK;
  hipLaunchKernelGGL(key_pack_kernel, dim3(grid_for(n, kJBlock * 4)), dim3(kJBlock), 0, as_stream(stream), a, n, out_dev);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_join_key_pack_char(const void *col_dev, int width, int64_t n, int64_t *out_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || (n > 0 && (col_dev == nullptr || out_dev == nullptr))) return QSX_ERR_INVALID_ARGUMENT;
  if (width < 1 || width > 8) return QSX_ERR_UNSUPPORTED;
  if (n == 0) return QSX_OK;
  hipLaunchKernelGGL(key_pack_char_kernel, dim3(grid_for(n, kJBlock * 4)), dim3(kJBlock), 0, as_stream(stream),
                     static_cast<const unsigned char *>(col_dev), width, n, out_dev);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

static int join_key_pack_blocks_impl(int ncols, const int32_t *types, int64_t num_blocks, const int64_t *block_rows,
                                     const void *const *block_cols, const int32_t *block_code_widths, const void *const *block_dictionaries,
                                     int64_t *out_dev, int *out_exact, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (ncols < 1 || ncols > QSX_MAX_KEYS || types == nullptr || num_blocks < 0 || out_exact == nullptr ||
      (num_blocks > 0 && (block_rows == nullptr || block_cols == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  KeyPackArgs a;
  a.ncols = ncols;
  int bits = 0;
  for (int k = 0; k < ncols; ++k) {
    if (types[k] != QSX_INT && types[k] != QSX_LONG) return QSX_ERR_UNSUPPORTED;
    a.cols[k] = nullptr;
    a.is_long[k] = types[k] == QSX_LONG;
    a.shift[k] = bits < 64 ? bits : 0;
    bits += types[k] == QSX_LONG ? 64 : 32;
  }
  a.exact = bits <= 64;
  *out_exact = a.exact;
  std::vector<int64_t> first(static_cast<size_t>(num_blocks));
  std::vector<const void *> none(static_cast<size_t>(num_blocks), nullptr);
  int64_t total = 0;
  for (int64_t b = 0; b < num_blocks; ++b) {
    if (block_rows[b] < 0) return QSX_ERR_INVALID_ARGUMENT;
    first[b] = total;
    total += block_rows[b];
  }
  if (total == 0) return QSX_OK;
  if (out_dev == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  std::vector<long long> table;
  const long long tiles = build_run_table(512, num_blocks, block_rows, none.data(), nullptr, nullptr, first.data(), &table);
  if (tiles < 0) return QSX_ERR_INVALID_ARGUMENT;
  const long long cols_offset = static_cast<long long>(table.size());
  for (int64_t b = 0; b < num_blocks; ++b) {
    for (int k = 0; k < ncols; ++k) {
      const void *col = block_cols[b * ncols + k];
      if (block_rows[b] > 0 && col == nullptr) return QSX_ERR_INVALID_ARGUMENT;
      table.push_back(static_cast<long long>(reinterpret_cast<uintptr_t>(col)));
    }
  }
  bool coded = false;
  for (int64_t e = 0; block_code_widths != nullptr && e < num_blocks * ncols; ++e) {
    const int w = block_code_widths[e];
    if (w != 0 && w != 1 && w != 2 && w != 4) return QSX_ERR_INVALID_ARGUMENT;
    coded = coded || w != 0;
  }
  if (coded) {
    for (int64_t e = 0; e < num_blocks * ncols; ++e) table.push_back(block_code_widths[e]);
    for (int64_t e = 0; e < num_blocks * ncols; ++e) {
      const void *d = block_dictionaries != nullptr && block_code_widths[e] != 0 ? block_dictionaries[e] : nullptr;
      table.push_back(static_cast<long long>(reinterpret_cast<uintptr_t>(d)));
    }
  }
  hipStream_t s = as_stream(stream);
  const size_t bytes = table.size() * sizeof(long long);
  const long long *runs_dev = static_cast<const long long *>(staged_device_buffer(s, bytes));
  if (runs_dev == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  const int rc = staged_upload(s, table.data(), bytes);
  if (rc != QSX_OK) return rc;
  hipLaunchKernelGGL(key_pack_runs_kernel, dim3(grid_for(tiles, kJBlock / kWave)), dim3(kJBlock), 0, s, a, runs_dev, cols_offset, out_dev, coded ? 1 : 0);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_join_key_pack_blocks(int ncols, const int32_t *types, int64_t num_blocks, const int64_t *block_rows,
                             const void *const *block_cols, int64_t *out_dev, int *out_exact, qsx_stream_t stream) {
  return join_key_pack_blocks_impl(ncols, types, num_blocks, block_rows, block_cols, nullptr, nullptr, out_dev, out_exact, stream);
}
int qsx_join_key_pack_blocks_coded(int ncols, const int32_t *types, int64_t num_blocks, const int64_t *block_rows,
                                   const void *const *block_cols, const int32_t *block_code_widths, const void *const *block_dictionaries,
                                   int64_t *out_dev, int *out_exact, qsx_stream_t stream) {
  return join_key_pack_blocks_impl(ncols, types, num_blocks, block_rows, block_cols, block_code_widths, block_dictionaries, out_dev, out_exact, stream);
}

static int destroy_table(qsx_join_table_t *t, bool wait_for_device) {
  if (t == nullptr) return QSX_OK;
  if (wait_for_device) {
    (void)synchronize_owner_device(t->slots != nullptr ? static_cast<const void *>(t->slots) : static_cast<const void *>(t->head));
  }
  if (t->shadow != nullptr) (void)destroy_table(t->shadow, wait_for_device);
  (void)device_free_idle(t->slots);
  (void)device_free_idle(t->head);
  (void)device_free_idle(t->head3);
  (void)device_free_idle(t->compact);
  (void)device_free_idle(t->cover);
  if (t->pack_event != nullptr) (void)hipEventDestroy(t->pack_event);
  if (t->seal_event != nullptr) (void)hipEventDestroy(t->seal_event);
  for (auto &m : t->marks) {
    if (m.second != nullptr) (void)hipEventDestroy(m.second);
  }
  if (t->control_host != nullptr) (void)hipHostFree(t->control_host);
  (void)device_free_idle(t->ov);
  (void)device_free_idle(t->entries_dev);
  delete t;
  return QSX_OK;
}
int qsx_join_table_destroy(qsx_join_table_t *t) { return destroy_table(t, true); }
int qsx_join_table_release(qsx_join_table_t *t) { return destroy_table(t, false); }

int qsx_join_table_clear(qsx_join_table_t *t, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (t == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  std::unique_lock<std::shared_mutex> lock(t->mutex);
  if (t->dense) {
    QSX_HIP_TRY(hipMemsetAsync(t->head, 0, t->range * sizeof(uint32_t), as_stream(stream)));
  } else {
    const int rc_fill = fill_empty(t, t->slots, t->capacity, as_stream(stream));
    if (rc_fill != QSX_OK) return rc_fill;
  }
  const int rc_control = reset_control_words(t->entries_dev, as_stream(stream));
  if (rc_control != QSX_OK) return rc_control;
  t->reserved = 0;
  t->max_tid.store(-1);
  t->seal_state.store(0);
  t->scanned.store(false);
  drop_cover(t);
  mark_stream(t, as_stream(stream));
  return QSX_OK;
}

int qsx_join_table_size(qsx_join_table_t *t, int64_t *out_entries, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (t == nullptr || out_entries == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  unsigned long long v[kControlWords] = {0, 0, 0, 0, 0, 0};
  const int rc_words = copy_control_words(t->entries_dev, as_stream(stream), v);
  if (rc_words != QSX_OK) return rc_words;
  *out_entries = static_cast<int64_t>(v[0]);
  // dense flavour: a build key outside [min_key, max_key] was skipped — the caller's statistics were not exact
  if (t->dense && static_cast<int>(v[3] & 0xFFFFFFFFu) != 0) return QSX_ERR_INVALID_ARGUMENT;
  return QSX_OK;
}

int qsx_join_build(qsx_join_table_t *t, const void *keys_dev, int64_t n, int32_t base_tid,
                   const uint64_t *filter_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (t == nullptr || n < 0 || base_tid < 0) return QSX_ERR_INVALID_ARGUMENT;
  if (n == 0) return QSX_OK;
  if (keys_dev == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  if (static_cast<int64_t>(base_tid) + n > INT32_MAX) return QSX_ERR_INVALID_ARGUMENT;
  int rc = ensure_room(t, n);
  if (rc != QSX_OK) return rc;
  t->seal_state.store(0);
  t->scanned.store(false);
  drop_cover(t);
  for (int64_t seen = t->max_tid.load(); seen < base_tid + n - 1 && !t->max_tid.compare_exchange_weak(seen, base_tid + n - 1);) {}
  std::shared_lock<std::shared_mutex> lock(t->mutex);
  const int grid = grid_for(n, kJBlock * 4);
  if (t->dense) {
    const int dgrid = grid_for((n + 63) >> 6, (kDBlock / kWave) * kBuildR);
    if (t->key_type == QSX_INT) {
      hipLaunchKernelGGL(dense_build_kernel<int32_t>, dim3(dgrid), dim3(kDBlock), 0, as_stream(stream), t->dense_view(),
                         static_cast<const int32_t *>(keys_dev), n, base_tid, filter_dev, t->entries_dev);
    } else {
      hipLaunchKernelGGL(dense_build_kernel<int64_t>, dim3(dgrid), dim3(kDBlock), 0, as_stream(stream), t->dense_view(),
                         static_cast<const int64_t *>(keys_dev), n, base_tid, filter_dev, t->entries_dev);
    }
    QSX_CHECK_LAUNCH();
    return QSX_OK;
  }
  if (t->key_type == QSX_INT) {
    hipLaunchKernelGGL(build_kernel<int32_t>, dim3(grid), dim3(kJBlock), 0, as_stream(stream), t->view(),
                       static_cast<const int32_t *>(keys_dev), n, base_tid, filter_dev, t->entries_dev);
  } else {
    hipLaunchKernelGGL(build_kernel<int64_t>, dim3(grid), dim3(kJBlock), 0, as_stream(stream), t->view(),
                       static_cast<const int64_t *>(keys_dev), n, base_tid, filter_dev, t->entries_dev);
  }
  QSX_CHECK_LAUNCH();
  mark_stream(t, as_stream(stream));
  return QSX_OK;
}

// A run's key coding as the kernels take it (block_runs.hpp): nullptr when there is none to speak of.
static int check_key_coding(const qsx_key_coding_t **coding, int64_t num_blocks) {
  if (*coding == nullptr) return QSX_OK;
  if ((*coding)->block_code_width == nullptr) {
    *coding = nullptr;
    return QSX_OK;
  }
  bool any = false;
  for (int64_t b = 0; b < num_blocks; ++b) {
    const int w = (*coding)->block_code_width[b];
    if (w != 0 && w != 1 && w != 2 && w != 4) return QSX_ERR_INVALID_ARGUMENT;
    if (w == 0 && (*coding)->block_dictionaries != nullptr && (*coding)->block_dictionaries[b] != nullptr) return QSX_ERR_INVALID_ARGUMENT;
    any = any || w != 0;
  }
  if (!any) *coding = nullptr;   // every stripe holds values: the plain run
  return QSX_OK;
}

static int join_build_blocks_impl(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                          const int32_t *block_base_tids, const uint64_t *const *block_filters, const qsx_key_coding_t *coding, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (t == nullptr || num_blocks < 0 || (num_blocks > 0 && (block_rows == nullptr || block_keys == nullptr || block_base_tids == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (check_key_coding(&coding, num_blocks) != QSX_OK) return QSX_ERR_INVALID_ARGUMENT;
  std::vector<int64_t> base(static_cast<size_t>(num_blocks));
  int64_t total = 0;
  for (int64_t b = 0; b < num_blocks; ++b) {
    if (block_rows[b] < 0 || (block_rows[b] > 0 && block_keys[b] == nullptr) || block_base_tids[b] < 0 ||
        static_cast<int64_t>(block_base_tids[b]) + block_rows[b] > INT32_MAX) {
      return QSX_ERR_INVALID_ARGUMENT;
    }
    base[b] = block_base_tids[b];
    total += block_rows[b];
  }
  if (total == 0) return QSX_OK;
  int rc = ensure_room(t, total);
  if (rc != QSX_OK) return rc;
  t->seal_state.store(0);
  t->scanned.store(false);
  drop_cover(t);
  for (int64_t b = 0; b < num_blocks; ++b) {
    const int64_t last = base[b] + block_rows[b] - 1;
    for (int64_t seen = t->max_tid.load(); seen < last && !t->max_tid.compare_exchange_weak(seen, last);) {}
  }
  hipStream_t s = as_stream(stream);
  std::vector<long long> table;
  const long long groups = build_run_table(kBuildTile, num_blocks, block_rows, block_keys,
                                           reinterpret_cast<const void *const *>(block_filters), nullptr, base.data(), &table,
                                           coding != nullptr ? coding->block_code_width : nullptr,
                                           coding != nullptr ? coding->block_dictionaries : nullptr);
  if (groups < 0) return QSX_ERR_INVALID_ARGUMENT;
  const size_t bytes = table.size() * sizeof(long long);
  const long long *runs_dev = static_cast<const long long *>(staged_device_buffer(s, bytes));
  if (runs_dev == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  rc = staged_upload(s, table.data(), bytes);
  if (rc != QSX_OK) return rc;
  std::shared_lock<std::shared_mutex> lock(t->mutex);
  if (t->dense) {
    const int dgrid = grid_for(groups, kDBlock / kWave);
    if (t->key_type == QSX_INT) {
      hipLaunchKernelGGL((dense_build_kernel<int32_t, true>), dim3(dgrid), dim3(kDBlock), 0, s, t->dense_view(),
                         static_cast<const int32_t *>(nullptr), total, 0, static_cast<const uint64_t *>(nullptr), t->entries_dev, runs_dev);
    } else {
      hipLaunchKernelGGL((dense_build_kernel<int64_t, true>), dim3(dgrid), dim3(kDBlock), 0, s, t->dense_view(),
                         static_cast<const int64_t *>(nullptr), total, 0, static_cast<const uint64_t *>(nullptr), t->entries_dev, runs_dev);
    }
  } else {
    const int grid = grid_for(groups, kJBlock / kWave);
    if (t->key_type == QSX_INT) {
      hipLaunchKernelGGL(build_runs_kernel<int32_t>, dim3(grid), dim3(kJBlock), 0, s, t->view(), runs_dev, t->entries_dev);
    } else {
      hipLaunchKernelGGL(build_runs_kernel<int64_t>, dim3(grid), dim3(kJBlock), 0, s, t->view(), runs_dev, t->entries_dev);
    }
  }
  QSX_CHECK_LAUNCH();
  mark_stream(t, s);
  return QSX_OK;
}

int qsx_join_build_blocks(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                          const int32_t *block_base_tids, const uint64_t *const *block_filters, qsx_stream_t stream) {
  return join_build_blocks_impl(t, num_blocks, block_rows, block_keys, block_base_tids, block_filters, nullptr, stream);
}
int qsx_join_build_blocks_coded(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                const qsx_key_coding_t *coding, const int32_t *block_base_tids, const uint64_t *const *block_filters,
                                qsx_stream_t stream) {
  return join_build_blocks_impl(t, num_blocks, block_rows, block_keys, block_base_tids, block_filters, coding, stream);
}

}  // extern "C"

// Tables the probe kernels copy into LDS (join_lds.hpp); QSX_JOIN_LDS=0 keeps every lookup in HBM / L2.
static std::atomic<long long> g_compact_probes{0};
// Test hook (not part of include/qsx.h): probe launches this process has issued against a table's compact plane.
extern "C" long long qsx_debug_join_compact_probes(void) { return g_compact_probes.load(std::memory_order_relaxed); }
static bool lds_tables_enabled() {
  const char *e = getenv("QSX_JOIN_LDS");
  return e == nullptr || e[0] != '0';
}

// The two-pass dense probe pays a second read of the keys and wins when the lookups are cheap and the tiles many: with
// a filter (LIP / predicate bitmap: few live rows) by default; QSX_JOIN_TWO_PASS=1 / 0 forces it on / off.
static bool dense_two_pass(const uint64_t *filter) {
  const char *e = getenv("QSX_JOIN_TWO_PASS");
  if (e != nullptr && e[0] != '\0') return e[0] != '0';
  return filter != nullptr;
}

// A build or clear: the covering array of the last projection describes another table now.
static void drop_cover(qsx_join_table *t) {
  std::lock_guard<std::mutex> lock(t->cover_mutex);
  t->cover_state = 0;
  t->cover_sig.clear();
}

// ---- hashed table -> directly addressed shadow ------------------------------------------------------------------------------
// The reference's hash of an INT / LONG key is the key itself (types/TypedValue.hpp:575-592) modulo a prime slot count, so
// over a dense key domain — every TPC-H key — its table is directly addressed in all but name.  This table hashes
// multiplicatively (sparse domains must not cluster) and pays for it with a 16 MiB table per million keys that no L2 holds.
// So the first probe after the builds looks at the bounds of the inserted keys (KeyBounds, kept by the build kernels): when
// they span at most 8 slots per entry the entries are copied into a directly addressed table (join_dense.hpp: one 4-byte word
// per key value, duplicates chained) and probes go there — no statistics from the optimizer needed.  A build or clear makes
// the next probe look again.  QSX_JOIN_ADAPTIVE=0 switches it off (the hashed kernels' own tests and measurements).
constexpr int64_t kAdaptiveMinRows = 1 << 16;
static bool adaptive_enabled() {
  const char *e = getenv("QSX_JOIN_ADAPTIVE");
  return e == nullptr || e[0] != '0';
}

// Directly addressed tables around the size of an XCD's L2: the first probe after the builds packs head[] to 3 bytes per key
// value when that makes it fit (4-byte words beyond ~3.25 MiB, the packed copy at most 3.6 MiB, every stored number below
// 2^23).  Builds keep working on head[]; a build or clear drops the copy.
static void sealed_pack(qsx_join_table *t, hipStream_t stream) {
  // packed on another stream a moment ago?  This stream's probe must not overtake the pack kernel.
  auto behind_the_pack = [&]() {
    if (t->pack_stream != stream && t->pack_event != nullptr && hipEventQuery(t->pack_event) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipStreamWaitEvent(stream, t->pack_event, 0);
    }
  };
  int state = t->seal_state.load(std::memory_order_acquire);
  if (state == 2) {
    behind_the_pack();
    return;
  }
  if (state != 0 || !adaptive_enabled()) return;
  const uint64_t bytes4 = t->range * 4, bytes3 = t->range * 3;
  // every number a head word can hold — tuple id + 1, overflow entry — stays below 2^23: both bounds are known on the host
  if (bytes4 <= (13ull << 18) || bytes3 > (36ull << 20) / 10 || t->max_tid.load() >= (1 << 23) - 2 || t->reserved >= (1 << 23) - 1) {
    return;   // (not marked: the test is four comparisons)
  }
  std::lock_guard<std::mutex> lock(t->seal_mutex);
  state = t->seal_state.load(std::memory_order_acquire);
  if (state != 0) {
    // another thread packed while this one waited for the mutex: its kernel may still be running on ITS stream, and this
    // thread's probe is about to read head3 (a probe that overtook the pack found zeros: 2.6 % of a join's rows lost once in
    // ~20 runs of 20 concurrent single-block work orders, tests/test_host_layer.py)
    if (state == 2) behind_the_pack();
    return;
  }
  // No host synchronisation: the pack kernel is ordered on the probing stream behind the builds the caller has ordered
  // before this probe (pipeline breaker), and other streams' probes wait for its event.
  if (t->head3 == nullptr && device_malloc(&t->head3, static_cast<size_t>(bytes3) + 4) != hipSuccess) {
    (void)hipGetLastError();
    t->head3 = nullptr;
    t->seal_state.store(1, std::memory_order_release);
    return;
  }
  if (t->pack_event == nullptr && hipEventCreateWithFlags(&t->pack_event, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    t->pack_event = nullptr;
    t->seal_state.store(1, std::memory_order_release);
    return;
  }
  hipLaunchKernelGGL(dense_pack_kernel, dim3(grid_for(static_cast<int64_t>(t->range), kJBlock * 4)), dim3(kJBlock), 0, stream, t->head, t->range,
                     t->head3);
  if (hipGetLastError() != hipSuccess || hipEventRecord(t->pack_event, stream) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(stream);
    t->seal_state.store(1, std::memory_order_release);
    return;
  }
  t->pack_stream = stream;
  t->seal_state.store(2, std::memory_order_release);
}

static bool compact_enabled() {
  const char *e = getenv("QSX_JOIN_COMPACT");   // (read per call: tests and tools compare the two forms)
  return e == nullptr || e[0] != '0';
}
// The compact plane of a sealed bucketed table (CompactView; called under seal_mutex with the device idle): true = built.
// What the host knows without the duplicate-key flag.
static bool compact_possible(const qsx_join_table *t) {
  const uint64_t buckets = t->capacity / kBucketSlots;
  return compact_enabled() && t->key_type == QSX_INT && buckets >= kCompactMinBuckets && buckets <= 0xFFFFFFFFull && t->max_tid.load() < (1 << 24);
}
static bool seal_compact(qsx_join_table *t, bool duplicate_keys, hipStream_t stream) {
  if (!compact_possible(t) || duplicate_keys) return false;
  if (t->compact_capacity != t->capacity) {
    (void)device_free(t->compact);
    t->compact = nullptr;
    t->compact_capacity = 0;
    if (device_malloc(&t->compact, static_cast<size_t>(t->capacity) * 5 + 16) != hipSuccess) {
      (void)hipGetLastError();
      t->compact = nullptr;
      return false;
    }
    t->compact_capacity = t->capacity;
  }
  hipLaunchKernelGGL(compact_build_kernel, dim3(static_cast<unsigned>((t->capacity + kJBlock - 1) / kJBlock)), dim3(kJBlock), 0, stream, t->view(),
                     static_cast<uint32_t *>(t->compact), static_cast<unsigned char *>(t->compact) + t->capacity * 4);
  // (no host wait: the plane is ordered on the sealing stream, probes of other streams wait for seal_event)
  if (hipGetLastError() != hipSuccess) return false;
  return true;
}

static bool seal_verify_enabled() {
  const char *e = getenv("QSX_JOIN_SEAL_VERIFY");   // tests: wait for the shadow's build and look at its error word
  return e != nullptr && e[0] == '1';
}
// Put `stream` behind every stream's last clear / build of this table (mark_stream; the calls themselves have returned:
// pipeline breaker) — device-side waits only.  false: some event could not be made or waited for.
static bool wait_for_builds(qsx_join_table *t, hipStream_t stream) {
  bool marked = true;
  std::lock_guard<std::mutex> lock(t->marks_mutex);
  for (const auto &m : t->marks) {
    if (m.second == nullptr) marked = false;
    else if (m.first != stream && hipStreamWaitEvent(stream, m.second, 0) != hipSuccess) marked = false;
  }
  if (!marked) (void)hipGetLastError();
  return marked;
}
// The control words of a hashed table as the builds (and the scan) left them, WITHOUT waiting for the device: `stream` is
// behind the builds (wait_for_builds), the words are copied behind them into pinned memory, and only that stream is
// synchronised — other Workers' streams keep running.
static bool read_control_words(qsx_join_table *t, hipStream_t stream, unsigned long long *v, bool *any_key) {
  if (t->control_host == nullptr && hipHostMalloc(reinterpret_cast<void **>(&t->control_host), kControlBytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    t->control_host = nullptr;
  }
  if (t->control_host == nullptr) return copy_control_words(t->entries_dev, stream, v, any_key) == QSX_OK;
  if (hipMemcpyAsync(t->control_host, t->entries_dev, kControlBytes, hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess) {
    return false;
  }
  *any_key = fold_control_words(t->control_host, v);
  return true;
}
// A probe on `stream` is about to read what another stream's seal made (shadow / compact plane): run behind its kernels.
static void behind_the_seal(qsx_join_table *t, hipStream_t stream) {
  if (t->seal_stream != stream && t->seal_event != nullptr && hipEventQuery(t->seal_event) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipStreamWaitEvent(stream, t->seal_event, 0);
  }
}
// Record seal_event behind what `stream` has queued for this table (scan, shadow, compact plane); false = no event and the
// stream could not be waited for either.
static bool record_seal(qsx_join_table *t, hipStream_t stream) {
  if (t->seal_event == nullptr && hipEventCreateWithFlags(&t->seal_event, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    t->seal_event = nullptr;
  }
  if (t->seal_event == nullptr || hipEventRecord(t->seal_event, stream) != hipSuccess) {
    (void)hipGetLastError();
    if (hipStreamSynchronize(stream) != hipSuccess) return false;
    t->seal_stream = nullptr;      // (finished: nobody has to wait)
    if (t->seal_event != nullptr) {
      (void)hipEventDestroy(t->seal_event);
      t->seal_event = nullptr;
    }
  } else {
    t->seal_stream = stream;
  }
  return true;
}
// Publish `state` behind the seal's kernels on `stream`.
static bool publish_seal(qsx_join_table *t, hipStream_t stream, int state) {
  if (!record_seal(t, stream)) return false;
  t->seal_state.store(state, std::memory_order_release);
  return true;
}

// probe_rows: the rows of the probe that asks (0: unknown).  A build side below kAdaptiveMinRows is only worth the look
// (a wait for the builds, a copy of the control words) in front of a probe of a million rows or more — and then its
// shadow, when the keys turn out dense, is a table the probe kernels hold in LDS (join_lds.hpp).
static qsx_join_table *sealed_shadow(qsx_join_table *t, hipStream_t stream, int64_t probe_rows = 0) {
  int state = t->seal_state.load(std::memory_order_acquire);
  if (state != 0) {
    behind_the_seal(t, stream);
    return state == 2 ? t->shadow : nullptr;
  }
  const bool small_build = t->reserved < kAdaptiveMinRows;
  const bool not_worth_a_look = small_build && (probe_rows < (1 << 20) || t->reserved < 1 || !lds_tables_enabled());
  if (not_worth_a_look && t->scanned.load(std::memory_order_acquire)) {   // (the short probes of a small table: no mutex)
    behind_the_seal(t, stream);
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(t->seal_mutex);
  state = t->seal_state.load(std::memory_order_acquire);
  if (state != 0) {
    behind_the_seal(t, stream);
    return state == 2 ? t->shadow : nullptr;
  }
  // The first probe since the last build: probes start after every build work order has finished (pipeline breaker), builds
  // issued on other streams included — this stream is put behind them (device-side waits), and every later probe (any stream)
  // runs behind seal_event.  A table that stays hashed gets its scan (fingerprint plane, duplicate keys) here, once; a table
  // that gets a directly addressed shadow needs neither.
  bool behind_builds = t->scanned.load(std::memory_order_acquire);
  if (behind_builds) behind_the_seal(t, stream);
  auto put_behind_builds = [&]() -> bool {
    if (behind_builds) return true;
    if (!wait_for_builds(t, stream) && hipDeviceSynchronize() != hipSuccess) return false;
    behind_builds = true;
    return true;
  };
  auto ensure_scanned = [&]() -> bool {
    if (t->scanned.load(std::memory_order_acquire)) return true;
    if (!put_behind_builds()) return false;
    const unsigned scan_grid = static_cast<unsigned>((t->capacity + kJBlock - 1) / kJBlock);
    if (t->key_type == QSX_INT) {
      hipLaunchKernelGGL(seal_scan_kernel<int32_t>, dim3(scan_grid), dim3(kJBlock), 0, stream, t->view());
    } else {
      hipLaunchKernelGGL(seal_scan_kernel<int64_t>, dim3(scan_grid), dim3(kJBlock), 0, stream, t->view());
    }
    if (hipGetLastError() != hipSuccess || !record_seal(t, stream)) return false;
    t->scanned.store(true, std::memory_order_release);
    return true;
  };
  auto stays_hashed = [&]() -> qsx_join_table * {   // (state 1 behind the scan; a failed scan leaves the table undecided and the probe fails)
    if (ensure_scanned()) t->seal_state.store(1, std::memory_order_release);
    return nullptr;
  };
  if (!adaptive_enabled()) return stays_hashed();
  // (undecided, not "stays hashed": a longer probe may still be worth the look)
  if (not_worth_a_look) {
    (void)ensure_scanned();
    return nullptr;
  }
  unsigned long long v[kControlWords];
  bool any_key = false;
  if (!put_behind_builds() || !read_control_words(t, stream, v, &any_key)) return stays_hashed();
  const uint64_t entries = v[0];
  const int64_t lo = static_cast<int64_t>(v[4] ^ 0x8000000000000000ull), hi = static_cast<int64_t>(v[5] ^ 0x8000000000000000ull);
  const uint64_t span = static_cast<uint64_t>(hi) - static_cast<uint64_t>(lo);
  // (a small build side: only when the whole key range fits the LDS table of the probe kernels)
  const bool fits_lds = small_build && entries >= 1 && span < static_cast<uint64_t>(kLdsDenseMaxWords);
  if (!any_key || (entries < static_cast<uint64_t>(kAdaptiveMinRows) && !fits_lds) || (span >= 8 * entries && !fits_lds) ||
      span >= (1ull << 32) || entries > 0x7FFFFFFFull) {
    // stays hashed — behind the compact plane of its slots when the table is one that can have it and the scan found no key twice
    // (the flag is the scan's: one more look at the control words, on this stream only)
    if (!ensure_scanned()) return nullptr;
    if (!compact_possible(t) || !read_control_words(t, stream, v, &any_key) || !seal_compact(t, v[2] != 0, stream) || !publish_seal(t, stream, 3)) {
      t->seal_state.store(1, std::memory_order_release);
    }
    return nullptr;
  }
  qsx_join_table *shadow = t->shadow;
  if (shadow != nullptr && (shadow->min_key != lo || shadow->range != span + 1 || shadow->ov_capacity < entries)) {
    (void)qsx_join_table_destroy(shadow);
    shadow = t->shadow = nullptr;
  }
  if (shadow == nullptr) {
    if (qsx_join_table_create_dense(t->key_type, lo, hi, 1, static_cast<int64_t>(entries), &shadow) != QSX_OK) {
      (void)hipGetLastError();
      return stays_hashed();   // no room for the shadow: the hashed table answers
    }
    t->shadow = shadow;
  } else if (qsx_join_table_clear(shadow, reinterpret_cast<qsx_stream_t>(stream)) != QSX_OK) {
    return stays_hashed();
  }
  shadow->reserved = static_cast<int64_t>(entries);
  shadow->max_tid.store(t->max_tid.load());
  const TableView src = t->view();
  hipLaunchKernelGGL(dense_build_from_slots_kernel, dim3(grid_for(static_cast<int64_t>(src.num_slots()), kJBlock * 4)), dim3(kJBlock), 0, stream,
                     t->key_type == QSX_LONG ? 1 : 0, src, shadow->dense_view());
  if (hipGetLastError() != hipSuccess) return stays_hashed();
  // The copy cannot fail on the device: the range is the exact min / max of the entries and the overflow list has room for
  // every one of them — so nobody waits for it on the host (QSX_JOIN_SEAL_VERIFY=1: the tests do, and read the error word).
  if (seal_verify_enabled()) {
    int error = 0;
    if (hipStreamSynchronize(stream) != hipSuccess ||
        hipMemcpy(&error, shadow->entries_dev + 3, sizeof(error), hipMemcpyDeviceToHost) != hipSuccess || error != 0) {
      set_last_error("QSX_JOIN_SEAL_VERIFY: the shadow's build reported an error", hipErrorUnknown);
      return stays_hashed();
    }
  }
  // other threads' probes (their own streams) may use it from here on, behind seal_event
  if (!publish_seal(t, stream, 2)) return stays_hashed();
  return shadow;
}

// The big-tile form of the dense probe (join_lds.hpp): one workgroup of 1024 threads per CU, one output reservation per 16 K
// rows; kLds: the table copied into the workgroup's LDS, else looked up where it lies; NLIP: LIP filters tested in the kernel.
template <int MODE, bool kRuns, bool kLds = true, int NLIP = 0>
static int launch_lds_dense_probe(qsx_join_table_t *t, const void *keys, int64_t n, int32_t probe_base_tid, const uint64_t *filter,
                                  int32_t *out_probe, int32_t *out_build, int64_t capacity, unsigned long long *count,
                                  uint64_t *out_bitmap, int anti, hipStream_t stream, const long long *runs_dev, int64_t tiles,
                                  const LipViews &lips = LipViews{}) {
  const size_t table_bytes = kLds ? ((static_cast<size_t>(t->range) * 4 + 15) & ~static_cast<size_t>(15)) : 0;
  auto launch = [&](auto key_tag) -> int {
    using KeyT = decltype(key_tag);
    auto kernel = &lds_dense_probe_kernel<KeyT, MODE, kRuns, kLds, NLIP>;
    if (table_bytes > 48 * 1024) {
      // (a property of (kernel, device); setting it again is a cheap host call)
      QSX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - kLdsStaticBytes));
    }
    const int sub = kLds ? kLdsSub : kLdsSubGlobal, per_cu = kLds ? 1 : 2;
    const int64_t units = (tiles + sub - 1) / sub;
    const int grid = static_cast<int>(units < static_cast<int64_t>(per_cu) * kCUs ? units : static_cast<int64_t>(per_cu) * kCUs);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kLdsBlock), table_bytes, stream, t->dense_view(), static_cast<const KeyT *>(keys), n, probe_base_tid,
                       filter, out_probe, out_build, capacity, count, out_bitmap, anti, runs_dev, lips);
    QSX_CHECK_LAUNCH();
    return QSX_OK;
  };
  return t->key_type == QSX_INT ? launch(int32_t{}) : launch(int64_t{});
}

// A pair-emitting probe under a filter in ONE pass (join_lds.hpp, kLds = false) instead of count / scan / write;
// QSX_JOIN_ONE_PASS=0 keeps the two passes.
static bool one_pass_enabled() {
  const char *e = getenv("QSX_JOIN_ONE_PASS");
  return e == nullptr || e[0] != '0';
}

// kRuns: the probe side is a run of blocks — runs_dev is its table (block_runs.hpp, tiles of 4096 rows), run_tiles its tile
// count, n the rows of all blocks together and `filter` non-NULL when any block has one; keys / probe_base_tid /
// out_bitmap come from the table.
template <int MODE, bool kRuns = false>
static int launch_probe(qsx_join_table_t *t, const void *keys, int64_t n, int32_t probe_base_tid,
                        const uint64_t *filter, int32_t *out_probe, int32_t *out_build,
                        int64_t capacity, int64_t *out_count, uint64_t *out_bitmap, int anti,
                        hipStream_t stream, const long long *runs_dev = nullptr, int64_t run_tiles = 0) {
  static_assert(kDenseTile == kProbeTile, "one run table serves both table kinds");
  if (!t->dense && n != 0) {
    qsx_join_table *shadow = sealed_shadow(t, stream, n);
    if (shadow != nullptr) {
      return launch_probe<MODE, kRuns>(shadow, keys, n, probe_base_tid, filter, out_probe, out_build, capacity, out_count, out_bitmap,
                                       anti, stream, runs_dev, run_tiles);
    }
    // the hashed kernels read the fingerprint plane, which the scan of the first probe writes: no scan, no probe
    if (!t->scanned.load(std::memory_order_acquire)) {
      set_last_error("the scan in front of a hashed table's first probe could not be issued", hipErrorUnknown);
      return QSX_ERR_HIP;
    }
  }
  if (out_count != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(int64_t), stream));
  if (n == 0) return QSX_OK;
  if (t->dense) sealed_pack(t, stream);
  std::shared_lock<std::shared_mutex> lock(t->mutex);
  if (t->dense) {
    const int64_t tiles = kRuns ? run_tiles : (n + kDenseTile - 1) / kDenseTile;
    const int64_t limit = 8 * kCUs;  // no LDS: 8 workgroups (32 waves) per CU
    const int dgrid = static_cast<int>(tiles < limit ? tiles : limit);
    unsigned long long *dcount = reinterpret_cast<unsigned long long *>(out_count);
    if constexpr (MODE == 0 || MODE == 1 || MODE == 2) {
      // A table of a few ten thousand key values: every workgroup holds it in LDS (join_lds.hpp).  Worth it when the probe
      // is long against the copies (a workgroup copies range words once and then walks its tiles).
      const bool two_pass = MODE == 0 && dense_two_pass(filter);
      if (!two_pass && lds_tables_enabled() && t->range <= static_cast<uint64_t>(kLdsDenseMaxWords) && n >= 32 * static_cast<int64_t>(t->range) &&
          n >= (1 << 18)) {
        return launch_lds_dense_probe<MODE, kRuns>(t, keys, n, probe_base_tid, filter, out_probe, out_build, capacity, dcount, out_bitmap, anti,
                                                   stream, runs_dev, tiles);
      }
    }
    if constexpr (MODE == 0) {
      if (dense_two_pass(filter) && one_pass_enabled() && n >= (1 << 20)) {
        return launch_lds_dense_probe<0, kRuns, false>(t, keys, n, probe_base_tid, filter, out_probe, out_build, capacity, dcount, out_bitmap, anti,
                                                       stream, runs_dev, tiles);
      }
    }
    if (MODE == 0 && dense_two_pass(filter)) {
      // count per (tile, wave) -> scan -> write: see join_dense.hpp
      const int64_t units = tiles * (kDBlock / kWave);
      CallScratch scratch(stream);
      const size_t bytes_counts = static_cast<size_t>(units) * 4, bytes_offsets = static_cast<size_t>(units + 1) * 8,
                   bytes_scan = scan_workspace_words(units) * 8;
      const int rc_scratch = scratch.reserve(CallScratch::padded(bytes_counts) + CallScratch::padded(bytes_offsets) + CallScratch::padded(bytes_scan));
      if (rc_scratch != QSX_OK) return rc_scratch;
      int32_t *unit_counts = static_cast<int32_t *>(scratch.take(bytes_counts));
      int64_t *unit_offsets = static_cast<int64_t *>(scratch.take(bytes_offsets)), *scan_ws = static_cast<int64_t *>(scratch.take(bytes_scan));
      if (t->key_type == QSX_INT) {
        hipLaunchKernelGGL((dense_probe_kernel<int32_t, 3, kRuns>), dim3(dgrid), dim3(kDBlock), 0, stream, t->dense_view(),
                           static_cast<const int32_t *>(keys), n, probe_base_tid, filter, out_probe, out_build, capacity,
                           dcount, out_bitmap, anti, unit_counts, unit_offsets, runs_dev);
      } else {
        hipLaunchKernelGGL((dense_probe_kernel<int64_t, 3, kRuns>), dim3(dgrid), dim3(kDBlock), 0, stream, t->dense_view(),
                           static_cast<const int64_t *>(keys), n, probe_base_tid, filter, out_probe, out_build, capacity,
                           dcount, out_bitmap, anti, unit_counts, unit_offsets, runs_dev);
      }
      QSX_CHECK_LAUNCH();
      QSX_HIP_TRY(launch_scan(unit_counts, units, unit_offsets, out_count, scan_ws, stream));
      if (t->key_type == QSX_INT) {
        hipLaunchKernelGGL((dense_probe_kernel<int32_t, 4, kRuns>), dim3(dgrid), dim3(kDBlock), 0, stream, t->dense_view(),
                           static_cast<const int32_t *>(keys), n, probe_base_tid, filter, out_probe, out_build, capacity,
                           dcount, out_bitmap, anti, unit_counts, unit_offsets, runs_dev);
      } else {
        hipLaunchKernelGGL((dense_probe_kernel<int64_t, 4, kRuns>), dim3(dgrid), dim3(kDBlock), 0, stream, t->dense_view(),
                           static_cast<const int64_t *>(keys), n, probe_base_tid, filter, out_probe, out_build, capacity,
                           dcount, out_bitmap, anti, unit_counts, unit_offsets, runs_dev);
      }
      QSX_CHECK_LAUNCH();
      return QSX_OK;
    }
    if (t->key_type == QSX_INT) {
      hipLaunchKernelGGL((dense_probe_kernel<int32_t, MODE, kRuns>), dim3(dgrid), dim3(kDBlock), 0, stream, t->dense_view(),
                         static_cast<const int32_t *>(keys), n, probe_base_tid, filter, out_probe, out_build, capacity,
                         dcount, out_bitmap, anti, nullptr, nullptr, runs_dev);
    } else {
      hipLaunchKernelGGL((dense_probe_kernel<int64_t, MODE, kRuns>), dim3(dgrid), dim3(kDBlock), 0, stream, t->dense_view(),
                         static_cast<const int64_t *>(keys), n, probe_base_tid, filter, out_probe, out_build, capacity,
                         dcount, out_bitmap, anti, nullptr, nullptr, runs_dev);
    }
    QSX_CHECK_LAUNCH();
    return QSX_OK;
  }
  const int64_t num_tiles = kRuns ? run_tiles : (n + kProbeTile - 1) / kProbeTile;
  if constexpr (MODE == 0 || MODE == 1 || MODE == 2) {
    // a bucketed table of a few thousand keys: fingerprint plane and slots copied into every workgroup's LDS
    // (join_lds_bucket.hpp), for a probe that is long against the copies
    if (lds_tables_enabled() && t->table_bytes() <= kLdsBucketMaxBytes && n >= (1 << 18) && n >= 8 * static_cast<int64_t>(t->capacity)) {
      const size_t table_bytes = (t->table_bytes() + 15) & ~static_cast<size_t>(15);
      auto launch = [&](auto key_tag) -> int {
        using KeyT = decltype(key_tag);
        auto kernel = &lds_bucket_probe_kernel<KeyT, MODE, kRuns>;
        if (table_bytes > 48 * 1024) {
          QSX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - kLdsStaticBytes));
        }
        const int64_t units = (num_tiles + kLdsSub - 1) / kLdsSub;
        const int lgrid = static_cast<int>(units < kCUs ? units : kCUs);
        hipLaunchKernelGGL(kernel, dim3(lgrid), dim3(kLdsBlock), table_bytes, stream, t->view(), static_cast<const KeyT *>(keys), n, probe_base_tid, filter,
                           out_probe, out_build, capacity, reinterpret_cast<unsigned long long *>(out_count), out_bitmap, anti, runs_dev);
        QSX_CHECK_LAUNCH();
        return QSX_OK;
      };
      return t->key_type == QSX_INT ? launch(int32_t{}) : launch(int64_t{});
    }
  }
  // 4 workgroups per CU keep 128 KiB of the 160 KiB LDS busy in pair mode.
  const int64_t max_grid = MODE == 0 ? 4 * kCUs : 8 * kCUs;
  const int grid = static_cast<int>(num_tiles < max_grid ? num_tiles : max_grid);
  unsigned long long *count = reinterpret_cast<unsigned long long *>(out_count);
  if constexpr (MODE == 0 || MODE == 1 || MODE == 2) {
    if (t->key_type == QSX_INT && t->seal_state.load(std::memory_order_acquire) == 3 && compact_enabled()) {
      g_compact_probes.fetch_add(1, std::memory_order_relaxed);
      hipLaunchKernelGGL((probe_fp_kernel<int32_t, MODE, kRuns, true>), dim3(grid), dim3(kJBlock), 0, stream, t->view(),
                         static_cast<const int32_t *>(keys), n, probe_base_tid, filter, out_probe,
                         out_build, capacity, count, out_bitmap, anti, runs_dev, t->compact_view());
      QSX_CHECK_LAUNCH();
      return QSX_OK;
    }
  }
  if (t->key_type == QSX_INT) {
    hipLaunchKernelGGL((probe_fp_kernel<int32_t, MODE, kRuns>), dim3(grid), dim3(kJBlock), 0, stream, t->view(),
                       static_cast<const int32_t *>(keys), n, probe_base_tid, filter, out_probe,
                       out_build, capacity, count, out_bitmap, anti, runs_dev);
  } else {
    hipLaunchKernelGGL((probe_fp_kernel<int64_t, MODE, kRuns>), dim3(grid), dim3(kJBlock), 0, stream, t->view(),
                       static_cast<const int64_t *>(keys), n, probe_base_tid, filter, out_probe,
                       out_build, capacity, count, out_bitmap, anti, runs_dev);
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

// The table of a run of probe blocks on the device (staged_upload: stream-ordered, pinned); *rows_total = rows of the run,
// *any_filter = some block has a filter.  Tuple ids: block_base_tids[b] + row, or run-global row numbers when it is NULL.
// The covering array of `view`'s build-side columns for table t (directly addressed), built on first use: returns its entry
// size (4 / 8 / 16) and sets *cover, or 0 — no build-side column, entries wider than 16 bytes, duplicate build keys, an entry
// that came out all-ones, QSX_JOIN_COVER=0 — and the probe reads head[] and the stripes.  `view.table` must be on the device
// in stream order (it is the calling probe's own table); the signature says what the array was built from.
static int cover_for(qsx_join_table *t, const ProjectionView &view, int entry_bytes, bool wanted, const std::vector<long long> &signature,
                     hipStream_t stream, const void **cover) {
  const char *env = getenv("QSX_JOIN_COVER");   // (read per call: tests and tools compare the two forms)
  if (!wanted || entry_bytes <= 0 || (env != nullptr && env[0] == '0') || !adaptive_enabled()) return 0;
  std::lock_guard<std::mutex> lock(t->cover_mutex);
  if (t->cover_state != 0 && t->cover_sig == signature) {
    *cover = t->cover;
    return t->cover_state == 2 ? t->cover_entry_bytes : 0;
  }
  // another projection's array may still be read by probes in flight
  if (t->cover_state == 2 && hipDeviceSynchronize() != hipSuccess) return 0;
  t->cover_state = 1;
  t->cover_sig = signature;
  const size_t bytes = static_cast<size_t>(t->range) * entry_bytes;
  if (bytes > (size_t(1) << 30)) return 0;
  if (t->cover_bytes < bytes) {
    (void)device_free(t->cover);
    t->cover = nullptr;
    t->cover_bytes = 0;
    if (device_malloc(&t->cover, bytes + 16) != hipSuccess) {
      (void)hipGetLastError();
      t->cover = nullptr;
      return 0;
    }
    t->cover_bytes = bytes;
  }
  unsigned int *flags = reinterpret_cast<unsigned int *>(static_cast<char *>(t->cover) + bytes);   // (16 spare bytes behind the array)
  if (hipMemsetAsync(flags, 0, sizeof(unsigned int), stream) != hipSuccess) return 0;
  const DenseTableView dv = t->dense_view();
  const unsigned grid = static_cast<unsigned>((t->range + kDBlock - 1) / kDBlock);
  switch (entry_bytes) {
    case 4: hipLaunchKernelGGL(cover_build_kernel<uint32_t>, dim3(grid), dim3(kDBlock), 0, stream, dv, view, static_cast<uint32_t *>(t->cover), flags); break;
    case 8: hipLaunchKernelGGL(cover_build_kernel<unsigned long long>, dim3(grid), dim3(kDBlock), 0, stream, dv, view,
                               static_cast<unsigned long long *>(t->cover), flags); break;
    default: hipLaunchKernelGGL(cover_build_kernel<ulonglong2>, dim3(grid), dim3(kDBlock), 0, stream, dv, view, static_cast<ulonglong2 *>(t->cover), flags); break;
  }
  unsigned int seen = 0;
  // (synchronous: once per table and projection; probes of other streams find a finished array)
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess ||
      hipMemcpy(&seen, flags, sizeof(seen), hipMemcpyDeviceToHost) != hipSuccess || seen != 0) {
    (void)hipGetLastError();
    return 0;
  }
  t->cover_entry_bytes = entry_bytes;
  t->cover_state = 2;
  *cover = t->cover;
  return entry_bytes;
}

static int upload_probe_run(int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                            const int32_t *block_base_tids, const uint64_t *const *block_filters, uint64_t *const *block_out,
                            hipStream_t stream, const long long **runs_dev, int64_t *tiles, int64_t *rows_total, bool *any_filter,
                            const std::vector<long long> *extra = nullptr, const long long **extra_dev = nullptr,
                            const qsx_key_coding_t *coding = nullptr) {
  std::vector<int64_t> base(static_cast<size_t>(num_blocks));
  int64_t total = 0;
  *any_filter = false;
  for (int64_t b = 0; b < num_blocks; ++b) {
    if (block_rows[b] < 0 || (block_rows[b] > 0 && block_keys[b] == nullptr)) return QSX_ERR_INVALID_ARGUMENT;
    if (block_rows[b] > 0 && block_out != nullptr && block_out[b] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
    base[b] = block_base_tids != nullptr ? block_base_tids[b] : total;
    if (base[b] < 0 || base[b] + block_rows[b] > INT32_MAX) return QSX_ERR_INVALID_ARGUMENT;
    total += block_rows[b];
    if (block_filters != nullptr && block_filters[b] != nullptr) *any_filter = true;
  }
  std::vector<long long> table;
  *tiles = build_run_table(kDenseTile, num_blocks, block_rows, block_keys, reinterpret_cast<const void *const *>(block_filters),
                           reinterpret_cast<void *const *>(block_out), base.data(), &table,
                           coding != nullptr ? coding->block_code_width : nullptr, coding != nullptr ? coding->block_dictionaries : nullptr);
  *rows_total = total;
  if (*tiles < 0) return QSX_ERR_INVALID_ARGUMENT;
  if (*tiles == 0) return QSX_OK;
  // (a second table of the same call rides behind the run table: one staging buffer per thread and stream)
  const size_t run_words = table.size();
  if (extra != nullptr) table.insert(table.end(), extra->begin(), extra->end());
  const size_t bytes = table.size() * sizeof(long long);
  *runs_dev = static_cast<const long long *>(staged_device_buffer(stream, bytes));
  if (*runs_dev == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  if (extra_dev != nullptr) *extra_dev = *runs_dev + run_words;
  return staged_upload(stream, table.data(), bytes);
}

namespace qsx {
// ---- an exact LIP filter from a directly addressed table (qsx_lip_build_from_join_table) ------------------------------------------
// The filter of a BuildHash work order is built over the very keys the table was (BuildHashOperator.cpp:187-203), and a
// directly addressed table already IS an existence map of its key range: head[k - min] != 0.  A wave reads 64 consecutive
// head words (coalesced), ballots "taken" into the filter word they stand for and ORs it in — 4 bytes streamed per key VALUE
// of the range instead of one atomic per key: Q3's 5.56 M qualifying orders in a range of 56 M: 0.05 ms against 0.25.
constexpr int kLipFromTableWords = kWave;   // filter words per wave and turn: lane r ends up holding word r's bits
__global__ __launch_bounds__(256) void lip_from_dense_kernel(DenseTableView t, LipView f, long long delta, long long first_word,
                                                             long long end_word) {
  const int lane = lane_id();
  const long long wave = static_cast<long long>(blockIdx.x) * (256 / kWave) + (threadIdx.x >> 6);
  const long long num_waves = static_cast<long long>(gridDim.x) * (256 / kWave);
  for (long long w0 = first_word + wave * kLipFromTableWords; w0 < end_word; w0 += num_waves * kLipFromTableWords) {
    // 64 words a turn, eight head reads in flight at a time; the ballot of word r stays in lane r, and ONE atomic instruction
    // of the wave sets all 64 words (a lane that looked its word up first and set it then kept every wave waiting on two
    // dependent round trips per word: 0.44 ms for 56 M key values; this way 0.06)
    unsigned long long mine = 0ull;
#pragma unroll
    for (int r0 = 0; r0 < kLipFromTableWords; r0 += 8) {
      uint32_t head[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long word = w0 + r0 + u;
        const long long bit = (word << 6) + lane;                 // bit of the filter = key - f.min_value
        const long long idx = bit - delta;                        // head word of that key: key - t.min_key
        const bool inside = word < end_word && bit < f.cardinality && idx >= 0 && idx < static_cast<long long>(t.range);
        head[u] = inside ? load_global_nt(&t.head[idx]) : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned long long mask = __ballot(head[u] != 0u);  // lane i <-> bit i of the word: the filter is LSB-first
        if (lane == r0 + u) mine = mask;
      }
    }
    if (mine != 0ull) atomicOr(&f.words[w0 + lane], mine);        // (a word behind end_word has no bits: nothing inside it was read)
  }
}
}  // namespace qsx

extern "C" {

int qsx_join_probe(qsx_join_table_t *t, const void *keys_dev, int64_t n, int32_t probe_base_tid,
                   const uint64_t *filter_dev, int32_t *out_probe_tid_dev, int32_t *out_build_tid_dev,
                   int64_t capacity, int64_t *out_count_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (t == nullptr || n < 0 || capacity < 0 || out_count_dev == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  if (n > 0 && (keys_dev == nullptr || (capacity > 0 && (out_probe_tid_dev == nullptr || out_build_tid_dev == nullptr)))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (static_cast<int64_t>(probe_base_tid) + n > INT32_MAX) return QSX_ERR_INVALID_ARGUMENT;
  return launch_probe<0>(t, keys_dev, n, probe_base_tid, filter_dev, out_probe_tid_dev,
                         out_build_tid_dev, capacity, out_count_dev, nullptr, 0, as_stream(stream));
}

int qsx_join_probe_lip(qsx_join_table_t *t, const void *keys_dev, int64_t n, int32_t probe_base_tid, const uint64_t *filter_dev,
                       int num_lip, const qsx_lip_filter_t *const *lip_filters, int32_t *out_probe_tid_dev, int32_t *out_build_tid_dev,
                       int64_t capacity, int64_t *out_count_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (t == nullptr || n < 0 || capacity < 0 || out_count_dev == nullptr || num_lip < 0 || (num_lip > 0 && lip_filters == nullptr)) return QSX_ERR_INVALID_ARGUMENT;
  if (n > 0 && (keys_dev == nullptr || (capacity > 0 && (out_probe_tid_dev == nullptr || out_build_tid_dev == nullptr)))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (static_cast<int64_t>(probe_base_tid) + n > INT32_MAX) return QSX_ERR_INVALID_ARGUMENT;
  for (int f = 0; f < num_lip; ++f) {
    if (lip_filters[f] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = as_stream(stream);
  if (num_lip == 0) {
    return launch_probe<0>(t, keys_dev, n, probe_base_tid, filter_dev, out_probe_tid_dev, out_build_tid_dev, capacity, out_count_dev, nullptr, 0, s);
  }
  // the table the probe will really read: a directly addressed one (or a hashed table's shadow) takes the filters into its kernel
  qsx_join_table *direct = t->dense ? t : (n != 0 ? sealed_shadow(t, s, n) : nullptr);
  if (direct != nullptr && num_lip <= kMaxFusedLip && n >= (1 << 16) && one_pass_enabled()) {
    QSX_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), s));
    sealed_pack(direct, s);
    std::shared_lock<std::shared_mutex> lock(direct->mutex);
    LipViews lips{};
    for (int f = 0; f < num_lip; ++f) lips.f[f] = lip_filter_view(lip_filters[f]);
    const int64_t tiles = (n + kDenseTile - 1) / kDenseTile;
    unsigned long long *count = reinterpret_cast<unsigned long long *>(out_count_dev);
    if (num_lip == 1) {
      return launch_lds_dense_probe<0, false, false, 1>(direct, keys_dev, n, probe_base_tid, filter_dev, out_probe_tid_dev, out_build_tid_dev, capacity, count,
                                                        nullptr, 0, s, nullptr, tiles, lips);
    }
    return launch_lds_dense_probe<0, false, false, 2>(direct, keys_dev, n, probe_base_tid, filter_dev, out_probe_tid_dev, out_build_tid_dev, capacity, count,
                                                      nullptr, 0, s, nullptr, tiles, lips);
  }
  // any other table (or more filters than the kernel takes): the filters one after the other into a bitmap of the call, then
  // the probe under it — what the separate entry points do
  CallScratch scratch(s);
  const size_t words = static_cast<size_t>((n + 63) / 64 + 1);
  int rc = scratch.reserve(CallScratch::padded(words * 8) * 2);
  if (rc != QSX_OK) return rc;
  uint64_t *bitmaps[2] = {static_cast<uint64_t *>(scratch.take(words * 8)), static_cast<uint64_t *>(scratch.take(words * 8))};
  const uint64_t *in = filter_dev;
  for (int f = 0; f < num_lip; ++f) {
    rc = qsx_lip_probe(lip_filters[f], t->key_type, keys_dev, n, in, bitmaps[f & 1], nullptr, stream);
    if (rc != QSX_OK) return rc;
    in = bitmaps[f & 1];
  }
  return launch_probe<0>(t, keys_dev, n, probe_base_tid, in, out_probe_tid_dev, out_build_tid_dev, capacity, out_count_dev, nullptr, 0, s);
}

int qsx_join_probe_exists_lip(qsx_join_table_t *t, const void *keys_dev, int64_t n, const uint64_t *filter_dev, int num_lip,
                              const qsx_lip_filter_t *const *lip_filters, uint64_t *out_bitmap_dev, int64_t *out_count_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (t == nullptr || n < 0 || num_lip < 0 || (num_lip > 0 && lip_filters == nullptr) || (n > 0 && (keys_dev == nullptr || out_bitmap_dev == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  for (int f = 0; f < num_lip; ++f) {
    if (lip_filters[f] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = as_stream(stream);
  if (num_lip == 0) return launch_probe<2>(t, keys_dev, n, 0, filter_dev, nullptr, nullptr, 0, out_count_dev, out_bitmap_dev, 0, s);
  qsx_join_table *direct = t->dense ? t : (n != 0 ? sealed_shadow(t, s, n) : nullptr);
  if (direct != nullptr && num_lip <= kMaxFusedLip && n >= (1 << 16) && one_pass_enabled()) {
    if (out_count_dev != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), s));
    sealed_pack(direct, s);
    std::shared_lock<std::shared_mutex> lock(direct->mutex);
    LipViews lips{};
    for (int f = 0; f < num_lip; ++f) lips.f[f] = lip_filter_view(lip_filters[f]);
    const int64_t tiles = (n + kDenseTile - 1) / kDenseTile;
    unsigned long long *count = reinterpret_cast<unsigned long long *>(out_count_dev);
    if (num_lip == 1) {
      return launch_lds_dense_probe<2, false, false, 1>(direct, keys_dev, n, 0, filter_dev, nullptr, nullptr, 0, count, out_bitmap_dev, 0, s, nullptr, tiles, lips);
    }
    return launch_lds_dense_probe<2, false, false, 2>(direct, keys_dev, n, 0, filter_dev, nullptr, nullptr, 0, count, out_bitmap_dev, 0, s, nullptr, tiles, lips);
  }
  // any other table: the filters one after the other into a bitmap of the call, then the existence probe under it
  CallScratch scratch(s);
  const size_t words = static_cast<size_t>((n + 63) / 64 + 1);
  int rc = scratch.reserve(CallScratch::padded(words * 8) * 2);
  if (rc != QSX_OK) return rc;
  uint64_t *bitmaps[2] = {static_cast<uint64_t *>(scratch.take(words * 8)), static_cast<uint64_t *>(scratch.take(words * 8))};
  const uint64_t *in = filter_dev;
  for (int f = 0; f < num_lip; ++f) {
    rc = qsx_lip_probe(lip_filters[f], t->key_type, keys_dev, n, in, bitmaps[f & 1], nullptr, stream);
    if (rc != QSX_OK) return rc;
    in = bitmaps[f & 1];
  }
  return launch_probe<2>(t, keys_dev, n, 0, in, nullptr, nullptr, 0, out_count_dev, out_bitmap_dev, 0, s);
}

int qsx_join_probe_count(qsx_join_table_t *t, const void *keys_dev, int64_t n,
                         const uint64_t *filter_dev, int64_t *out_count_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (t == nullptr || n < 0 || out_count_dev == nullptr || (n > 0 && keys_dev == nullptr)) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  return launch_probe<1>(t, keys_dev, n, 0, filter_dev, nullptr, nullptr, 0, out_count_dev, nullptr, 0,
                         as_stream(stream));
}

int qsx_join_probe_exists(qsx_join_table_t *t, const void *keys_dev, int64_t n,
                          const uint64_t *filter_dev, int anti, uint64_t *out_bitmap_dev,
                          int64_t *out_count_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (t == nullptr || n < 0 || (n > 0 && (keys_dev == nullptr || out_bitmap_dev == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  return launch_probe<2>(t, keys_dev, n, 0, filter_dev, nullptr, nullptr, 0, out_count_dev,
                         out_bitmap_dev, anti, as_stream(stream));
}

static int join_probe_blocks_impl(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                          const int32_t *block_base_tids, const uint64_t *const *block_filters, int32_t *out_probe_tid_dev,
                          int32_t *out_build_tid_dev, int64_t capacity, int64_t *out_count_dev, const qsx_key_coding_t *coding, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (t == nullptr || num_blocks < 0 || capacity < 0 || out_count_dev == nullptr ||
      (num_blocks > 0 && (block_rows == nullptr || block_keys == nullptr)) ||
      (capacity > 0 && (out_probe_tid_dev == nullptr || out_build_tid_dev == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = as_stream(stream);
  const long long *runs_dev = nullptr;
  int64_t tiles = 0, rows = 0;
  bool any_filter = false;
  if (check_key_coding(&coding, num_blocks) != QSX_OK) return QSX_ERR_INVALID_ARGUMENT;
  const int rc = upload_probe_run(num_blocks, block_rows, block_keys, block_base_tids, block_filters, nullptr, s, &runs_dev, &tiles,
                                  &rows, &any_filter, nullptr, nullptr, coding);
  if (rc != QSX_OK) return rc;
  const uint64_t *filter_mark = any_filter ? reinterpret_cast<const uint64_t *>(runs_dev) : nullptr;   // only tested against NULL
  return launch_probe<0, true>(t, nullptr, rows, 0, filter_mark, out_probe_tid_dev, out_build_tid_dev, capacity, out_count_dev,
                               nullptr, 0, s, runs_dev, tiles);
}

static int join_probe_count_blocks_impl(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                        const uint64_t *const *block_filters, int64_t *out_count_dev, const qsx_key_coding_t *coding, qsx_stream_t stream);
static int join_probe_project_blocks_impl(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                  const uint64_t *const *block_filters, const qsx_join_projection_t *proj, int64_t capacity,
                                  int64_t *out_count_dev, const qsx_key_coding_t *coding, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (t == nullptr || num_blocks < 0 || capacity < 0 || out_count_dev == nullptr || proj == nullptr ||
      (num_blocks > 0 && (block_rows == nullptr || block_keys == nullptr)) || proj->num_columns < 1 ||
      proj->num_columns > QSX_MAX_PROJECTED || proj->out_columns == nullptr || proj->num_build_segments < 0) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  const int nc = proj->num_columns;
  bool any_build = false, any_probe = false;
  for (int c = 0; c < nc; ++c) {
    const int w = proj->width[c];
    if (w != 1 && w != 2 && w != 4 && w != 8) return QSX_ERR_INVALID_ARGUMENT;
    if (capacity > 0 && proj->out_columns[c] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
    (proj->on_build[c] != 0 ? any_build : any_probe) = true;
  }
  if (any_probe && num_blocks > 0 && proj->probe_stripes == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  if (any_build && proj->num_build_segments > 0 && (proj->build_first_tids == nullptr || proj->build_stripes == nullptr)) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  for (int sg = 1; sg < proj->num_build_segments; ++sg) {
    if (proj->build_first_tids[sg] < proj->build_first_tids[sg - 1]) return QSX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = as_stream(stream);
  if (capacity == 0 || (any_build && proj->num_build_segments == 0)) {
    // nowhere to write, or a build side without tuples (nothing can match): the count only
    return join_probe_count_blocks_impl(t, num_blocks, block_rows, block_keys, block_filters, out_count_dev, coding, stream);
  }
  if (check_key_coding(&coding, num_blocks) != QSX_OK) return QSX_ERR_INVALID_ARGUMENT;
  qsx_join_table *direct = t->dense ? t : sealed_shadow(t, s);
  if (direct == nullptr && coding != nullptr) {
    // (the gather below reads values: a projected column that IS a coded key stripe has none — the caller presents the decoded stripe)
    for (int64_t b = 0; b < num_blocks; ++b) {
      for (int c = 0; c < nc && coding->block_code_width[b] != 0 && block_rows[b] > 0; ++c) {
        if (proj->on_build[c] == 0 && proj->probe_stripes[static_cast<size_t>(b) * nc + c] == block_keys[b]) return QSX_ERR_UNSUPPORTED;
      }
    }
  }
  if (direct == nullptr) {
    // No directly addressed form of this table: the pair list after all, in scratch of this call, and one gather per column.
    int64_t total_rows = 0;
    for (int64_t b = 0; b < num_blocks; ++b) total_rows += block_rows[b] > 0 ? block_rows[b] : 0;
    QSX_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), s));
    if (total_rows == 0) return QSX_OK;
    int32_t *pairs = nullptr;
    QSX_HIP_TRY(device_malloc(reinterpret_cast<void **>(&pairs), static_cast<size_t>(capacity) * 8 + 16));
    int32_t *probe_tids = pairs, *build_tids = pairs + capacity;
    int rc = join_probe_blocks_impl(t, num_blocks, block_rows, block_keys, nullptr, block_filters, probe_tids, build_tids, capacity,
                                    out_count_dev, coding, stream);
    // how many pairs there are to gather (this path synchronises anyway: the pair list is released before it returns)
    int64_t pairs_found = 0;
    if (rc == QSX_OK && (hipMemcpyAsync(&pairs_found, out_count_dev, sizeof(int64_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
                         hipStreamSynchronize(s) != hipSuccess)) {
      (void)hipGetLastError();
      rc = QSX_ERR_HIP;
    }
    const int64_t gathered = pairs_found < capacity ? pairs_found : capacity;
    std::vector<int64_t> probe_first(static_cast<size_t>(num_blocks));
    std::vector<const void *> segs;
    int64_t at = 0;
    for (int64_t b = 0; b < num_blocks; ++b) {
      probe_first[static_cast<size_t>(b)] = at;
      at += block_rows[b];
    }
    for (int c = 0; c < nc && rc == QSX_OK && gathered > 0; ++c) {
      segs.clear();
      if (proj->on_build[c] != 0) {
        for (int sg = 0; sg < proj->num_build_segments; ++sg) segs.push_back(proj->build_stripes[static_cast<size_t>(sg) * nc + c]);
        rc = qsx_gather_segmented(proj->width[c], proj->num_build_segments, segs.data(), proj->build_first_tids, build_tids, gathered,
                                  proj->out_columns[c], stream);
      } else {
        for (int64_t b = 0; b < num_blocks; ++b) segs.push_back(proj->probe_stripes[static_cast<size_t>(b) * nc + c]);
        rc = qsx_gather_segmented(proj->width[c], static_cast<int>(num_blocks), segs.data(), probe_first.data(), probe_tids, gathered,
                                  proj->out_columns[c], stream);
      }
    }
    (void)hipStreamSynchronize(s);   // the pair list is this call's
    (void)device_free_idle(pairs);
    return rc;
  }
  // the projection's table (join_dense.hpp ProjectionView) rides behind the run table
  const int nseg = proj->num_build_segments > 0 ? proj->num_build_segments : 1;
  const size_t head_words = static_cast<size_t>(kProjColumnWords) * nc;
  std::vector<long long> table(head_words + nseg + static_cast<size_t>(nseg) * nc + static_cast<size_t>(num_blocks) * nc, 0);
  auto word_of = [](const void *p) { return static_cast<long long>(reinterpret_cast<uintptr_t>(p)); };
  // entries of the covering array: the build-side columns at their natural alignment, 4 / 8 / 16 bytes in all
  int cover_bytes = 0;
  std::vector<long long> signature;
  for (int c = 0; c < nc; ++c) {
    table[static_cast<size_t>(c)] = proj->width[c];
    table[static_cast<size_t>(nc + c)] = proj->on_build[c] != 0 ? 1 : 0;
    table[static_cast<size_t>(2 * nc + c)] = word_of(proj->out_columns[c]);
    if (proj->on_build[c] != 0) {
      cover_bytes = (cover_bytes + proj->width[c] - 1) / proj->width[c] * proj->width[c];
      table[static_cast<size_t>(3 * nc + c)] = cover_bytes;
      signature.push_back(proj->width[c]);
      signature.push_back(cover_bytes);
      cover_bytes += proj->width[c];
    }
  }
  cover_bytes = cover_bytes == 0 ? 0 : (cover_bytes <= 4 ? 4 : (cover_bytes <= 8 ? 8 : (cover_bytes <= 16 ? 16 : -1)));
  long long seg_rows = 0;
  for (int sg = 0; sg < proj->num_build_segments; ++sg) {
    table[head_words + sg] = proj->build_first_tids[sg];
    signature.push_back(proj->build_first_tids[sg]);
    if (sg > 0) {
      const long long len = proj->build_first_tids[sg] - proj->build_first_tids[sg - 1];
      if (sg == 1) seg_rows = len;
      else if (len != seg_rows) seg_rows = -1;
    }
    for (int c = 0; c < nc; ++c) {
      if (proj->on_build[c] != 0) {
        table[head_words + nseg + static_cast<size_t>(sg) * nc + c] = word_of(proj->build_stripes[static_cast<size_t>(sg) * nc + c]);
        signature.push_back(table[head_words + nseg + static_cast<size_t>(sg) * nc + c]);
      }
    }
  }
  if (seg_rows <= 0 || seg_rows > INT32_MAX) seg_rows = 0;
  const int key_width = direct->key_type == QSX_INT ? 4 : 8;
  for (int c = 0; c < nc; ++c) table[static_cast<size_t>(4 * nc + c)] = proj->on_build[c] == 0 && proj->width[c] == key_width && num_blocks > 0 ? 1 : 0;
  for (int64_t b = 0; b < num_blocks; ++b) {
    for (int c = 0; c < nc; ++c) {
      if (proj->on_build[c] == 0) {
        const void *stripe = proj->probe_stripes[static_cast<size_t>(b) * nc + c];
        if (stripe == nullptr && block_rows[b] > 0) return QSX_ERR_INVALID_ARGUMENT;
        table[head_words + nseg + static_cast<size_t>(nseg) * nc + static_cast<size_t>(b) * nc + c] = word_of(stripe);
        // the probe key itself (SELECT ... the join attribute): every block's stripe of the column is its key stripe
        if (block_rows[b] > 0 && stripe != block_keys[b]) table[static_cast<size_t>(4 * nc + c)] = 0;
      }
    }
  }
  const long long *runs_dev = nullptr, *proj_dev = nullptr;
  int64_t tiles = 0, rows = 0;
  bool any_filter = false;
  const int rc = upload_probe_run(num_blocks, block_rows, block_keys, nullptr, block_filters, nullptr, s, &runs_dev, &tiles, &rows,
                                  &any_filter, &table, &proj_dev, coding);
  if (rc != QSX_OK) return rc;
  QSX_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), s));
  if (rows == 0) return QSX_OK;
  sealed_pack(direct, s);
  ProjectionView view{proj_dev, nc, nseg, static_cast<int>(seg_rows), nullptr};
  const int cover_entry = cover_for(direct, view, cover_bytes, any_build && proj->num_build_segments > 0, signature, s, &view.cover);
  std::shared_lock<std::shared_mutex> lock(direct->mutex);
  const int64_t limit = 8 * kCUs;
  const int dgrid = static_cast<int>(tiles < limit ? tiles : limit);
  const uint64_t *filter_mark = any_filter ? reinterpret_cast<const uint64_t *>(runs_dev) : nullptr;   // only tested against NULL
  unsigned long long *count = reinterpret_cast<unsigned long long *>(out_count_dev);
  const DenseTableView dv = direct->dense_view();
  // 512 threads x 8 rows per tile: measured 0.72 ms per 100 M rows (int key + int attribute) against 0.87 at 1024 x 4 and
  // 0.85 at 256 x 16 (tools/probe_project.py)
  constexpr int kCoverBlock = 512;
  const int cgrid = static_cast<int>(tiles < 4 * kCUs ? tiles : 4 * kCUs);
  auto by_entry = [&](auto key_tag) {
    using KeyT = decltype(key_tag);
    switch (cover_entry) {
      case 4: hipLaunchKernelGGL((cover_probe_kernel<KeyT, uint32_t, kCoverBlock>), dim3(cgrid), dim3(kCoverBlock), 0, s, dv, capacity, count, runs_dev, view); break;
      case 8: hipLaunchKernelGGL((cover_probe_kernel<KeyT, unsigned long long, kCoverBlock>), dim3(cgrid), dim3(kCoverBlock), 0, s, dv, capacity, count, runs_dev, view); break;
      case 16: hipLaunchKernelGGL((cover_probe_kernel<KeyT, ulonglong2, kCoverBlock>), dim3(cgrid), dim3(kCoverBlock), 0, s, dv, capacity, count, runs_dev, view); break;
      default:
        hipLaunchKernelGGL((dense_probe_kernel<KeyT, 5, true>), dim3(dgrid), dim3(kDBlock), 0, s, dv, static_cast<const KeyT *>(nullptr), rows, 0,
                           filter_mark, static_cast<int32_t *>(nullptr), static_cast<int32_t *>(nullptr), capacity, count,
                           static_cast<uint64_t *>(nullptr), 0, static_cast<int32_t *>(nullptr), static_cast<const int64_t *>(nullptr), runs_dev,
                           view);
        break;
    }
  };
  if (direct->key_type == QSX_INT) by_entry(int32_t{}); else by_entry(int64_t{});
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

static int join_probe_count_blocks_impl(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                const uint64_t *const *block_filters, int64_t *out_count_dev, const qsx_key_coding_t *coding, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (t == nullptr || num_blocks < 0 || out_count_dev == nullptr || (num_blocks > 0 && (block_rows == nullptr || block_keys == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = as_stream(stream);
  const long long *runs_dev = nullptr;
  int64_t tiles = 0, rows = 0;
  bool any_filter = false;
  if (check_key_coding(&coding, num_blocks) != QSX_OK) return QSX_ERR_INVALID_ARGUMENT;
  const int rc = upload_probe_run(num_blocks, block_rows, block_keys, nullptr, block_filters, nullptr, s, &runs_dev, &tiles, &rows,
                                  &any_filter, nullptr, nullptr, coding);
  if (rc != QSX_OK) return rc;
  return launch_probe<1, true>(t, nullptr, rows, 0, nullptr, nullptr, nullptr, 0, out_count_dev, nullptr, 0, s, runs_dev, tiles);
}

static int join_probe_exists_blocks_impl(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                 const uint64_t *const *block_filters, int anti, uint64_t *const *block_out_bitmaps,
                                 int64_t *out_count_dev, const qsx_key_coding_t *coding, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (t == nullptr || num_blocks < 0 ||
      (num_blocks > 0 && (block_rows == nullptr || block_keys == nullptr || block_out_bitmaps == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = as_stream(stream);
  const long long *runs_dev = nullptr;
  int64_t tiles = 0, rows = 0;
  bool any_filter = false;
  if (check_key_coding(&coding, num_blocks) != QSX_OK) return QSX_ERR_INVALID_ARGUMENT;
  const int rc = upload_probe_run(num_blocks, block_rows, block_keys, nullptr, block_filters, block_out_bitmaps, s, &runs_dev,
                                  &tiles, &rows, &any_filter, nullptr, nullptr, coding);
  if (rc != QSX_OK) return rc;
  return launch_probe<2, true>(t, nullptr, rows, 0, nullptr, nullptr, nullptr, 0, out_count_dev, nullptr, anti, s, runs_dev, tiles);
}

int qsx_join_probe_blocks(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                          const int32_t *block_base_tids, const uint64_t *const *block_filters, int32_t *out_probe_tid_dev,
                          int32_t *out_build_tid_dev, int64_t capacity, int64_t *out_count_dev, qsx_stream_t stream) {
  return join_probe_blocks_impl(t, num_blocks, block_rows, block_keys, block_base_tids, block_filters, out_probe_tid_dev, out_build_tid_dev,
                                capacity, out_count_dev, nullptr, stream);
}
int qsx_join_probe_blocks_coded(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                const qsx_key_coding_t *coding, const int32_t *block_base_tids, const uint64_t *const *block_filters,
                                int32_t *out_probe_tid_dev, int32_t *out_build_tid_dev, int64_t capacity, int64_t *out_count_dev,
                                qsx_stream_t stream) {
  return join_probe_blocks_impl(t, num_blocks, block_rows, block_keys, block_base_tids, block_filters, out_probe_tid_dev, out_build_tid_dev,
                                capacity, out_count_dev, coding, stream);
}
int qsx_join_probe_project_blocks(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                  const uint64_t *const *block_filters, const qsx_join_projection_t *proj, int64_t capacity,
                                  int64_t *out_count_dev, qsx_stream_t stream) {
  return join_probe_project_blocks_impl(t, num_blocks, block_rows, block_keys, block_filters, proj, capacity, out_count_dev, nullptr, stream);
}
int qsx_join_probe_project_blocks_coded(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                        const qsx_key_coding_t *coding, const uint64_t *const *block_filters,
                                        const qsx_join_projection_t *proj, int64_t capacity, int64_t *out_count_dev, qsx_stream_t stream) {
  return join_probe_project_blocks_impl(t, num_blocks, block_rows, block_keys, block_filters, proj, capacity, out_count_dev, coding, stream);
}
int qsx_join_probe_count_blocks(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                const uint64_t *const *block_filters, int64_t *out_count_dev, qsx_stream_t stream) {
  return join_probe_count_blocks_impl(t, num_blocks, block_rows, block_keys, block_filters, out_count_dev, nullptr, stream);
}
int qsx_join_probe_count_blocks_coded(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                      const qsx_key_coding_t *coding, const uint64_t *const *block_filters, int64_t *out_count_dev,
                                      qsx_stream_t stream) {
  return join_probe_count_blocks_impl(t, num_blocks, block_rows, block_keys, block_filters, out_count_dev, coding, stream);
}
int qsx_join_probe_exists_blocks(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                 const uint64_t *const *block_filters, int anti, uint64_t *const *block_out_bitmaps,
                                 int64_t *out_count_dev, qsx_stream_t stream) {
  return join_probe_exists_blocks_impl(t, num_blocks, block_rows, block_keys, block_filters, anti, block_out_bitmaps, out_count_dev, nullptr, stream);
}
int qsx_join_probe_exists_blocks_coded(qsx_join_table_t *t, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                       const qsx_key_coding_t *coding, const uint64_t *const *block_filters, int anti,
                                       uint64_t *const *block_out_bitmaps, int64_t *out_count_dev, qsx_stream_t stream) {
  return join_probe_exists_blocks_impl(t, num_blocks, block_rows, block_keys, block_filters, anti, block_out_bitmaps, out_count_dev, coding, stream);
}

int qsx_lip_build_from_join_table(qsx_lip_filter_t *filter, qsx_join_table_t *t, int64_t num_new_keys, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (filter == nullptr || t == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  const LipView f = lip_filter_view(filter);
  std::shared_lock<std::shared_mutex> lock(t->mutex);
  if (!t->dense || t->stride_shift != 0 || !f.exact || f.cardinality <= 0 || t->range == 0) return QSX_ERR_UNSUPPORTED;
  const DenseTableView view = t->dense_view();
  // the filter words that stand for key values of the table's range
  const long long delta = static_cast<long long>(t->min_key) - f.min_value;            // bit = head index + delta
  const long long first_bit = delta > 0 ? delta : 0;
  const long long end_bit = delta + static_cast<long long>(view.range) < f.cardinality ? delta + static_cast<long long>(view.range) : f.cardinality;
  if (end_bit <= first_bit) return QSX_OK;                                             // no key of the table can be in the filter
  const long long first_word = first_bit >> 6, end_word = (end_bit + 63) >> 6;
  // worth it?  (end_bit - first_bit) head words of 4 bytes at ~4 TB/s against num_new_keys atomics at 23.7 G/s
  if (num_new_keys >= 0 && static_cast<double>(end_bit - first_bit) * 4.0 / 4.0e12 > static_cast<double>(num_new_keys) / 23.7e9) return QSX_ERR_UNSUPPORTED;
  const long long turns = (end_word - first_word + kLipFromTableWords - 1) / kLipFromTableWords;
  const long long want = (turns + (256 / kWave) - 1) / (256 / kWave);
  const int grid = static_cast<int>(want < 16 * kCUs ? (want > 0 ? want : 1) : 16 * kCUs);
  hipLaunchKernelGGL(lip_from_dense_kernel, dim3(grid), dim3(256), 0, as_stream(stream), view, f, delta, first_word, end_word);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

}  // extern "C"
