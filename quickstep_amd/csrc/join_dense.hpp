// join_dense.hpp — the directly addressed join table (qsx_join_table_create_dense).
//
// Same contract as the hashed table of join.hip (K3 build / K4 probe; reference loops:
// storage/HashTable.hpp:1358-1461 and :2145-2181, :1979-2062), chosen by the caller when the
// build-side join attribute has exact min/max statistics in a bounded range — the condition of
// query_optimizer/rules/InjectJoinFilters.cpp:130-150.
//
// Layout in HBM:
//   head[range]   uint32 per key value (key - min_key):
//                   0                      no build row has this key
//                   t + 1      (bit 31 = 0) exactly one build row, tuple reference t
//                   0x80000000 | e          the chain of this key starts at overflow entry e
//   ov[e]         {tid, next}  next has the format of a head word (0 terminates the chain)
// A primary-key build side (TPC-H custkey / orderkey) never touches ov: a probe is ONE random
// 4-byte read, and for the C2 shape (1 M keys) head[] is 4 MiB — it stays resident in every
// XCD's L2 while the probe keys and the output pairs stream past it with non-temporal accesses.
//
// Match compaction (MODE 0): a wave owns 16 x 64 probe rows of a tile.  It ballots the 16 steps and
// reserves the space of all its first-level matches with ONE global atomic (the single output counter
// serialises same-address atomics in L2: one per 64-row step measured 1.25 ms / 100 M rows, of which
// 0.7 ms was the counter; one per 1024 rows is 1/16 of that and needs no workgroup barrier).
// Every step's matches are then written as one contiguous run (mbcnt rank) straight to the output
// arrays — no staging of the pairs.  Duplicate-key chains (rare) continue with one wave-aggregated
// reservation per chain step.
#ifndef QSX_CSRC_JOIN_DENSE_HPP_
#define QSX_CSRC_JOIN_DENSE_HPP_

#include <type_traits>

#include "block_runs.hpp"
#include "common.hpp"

namespace qsx {

constexpr int kDBlock = 256;
constexpr int kDenseRowsPerThread = 16;
constexpr int kDenseTile = kDBlock * kDenseRowsPerThread;
constexpr uint32_t kChainBit = 0x80000000u;
constexpr int kDenseSparsePairs = 256;   // a wave with at most this many first-level matches in its 1024 rows stages them in LDS

struct DenseTableView {
  uint32_t *head;
  // != nullptr: head[] packed to 3 bytes per key value for the probes (join.hip sealed_pack): bits 0-22 = tid + 1 or the
  // overflow entry, bit 23 = chain.  For ~1 M keys that is 3 MiB instead of 4: what fits an XCD's L2 next to the streamed
  // keys and pairs (tools/ubench/gather_floor.hip: 100 M lookups + pairs 0.48 ms at 3 MiB, 0.57 ms at 4 MiB).
  const unsigned char *head3;
  uint2 *ov;
  int64_t min_key;
  int stride_shift;          // keys are min_key + i * 2^stride_shift (one hash partition of a dense key domain)
  uint64_t range;            // number of head words: ((max_key - min_key) >> stride_shift) + 1
  unsigned int *ov_count;    // overflow entries handed out
  unsigned int ov_capacity;
  int *error;                // set when a build key is outside the range (or ov ran out: host bug)
};

// Head index of a key, or ~0 when the key is not a member of the table's progression.
template <typename KeyT>
__device__ __forceinline__ uint64_t dense_index(const DenseTableView &t, KeyT key) {
  const uint64_t d = static_cast<uint64_t>(static_cast<int64_t>(key) - t.min_key);
  const uint64_t idx = d >> t.stride_shift;
  return (idx << t.stride_shift) == d && idx < t.range ? idx : ~0ull;
}

// The head word of key value idx as the probe kernels read it.
__device__ __forceinline__ uint32_t dense_head_word(const DenseTableView &t, uint64_t idx) {
  if (t.head3 == nullptr) return t.head[idx];
  uint32_t w;
  __builtin_memcpy(&w, t.head3 + idx * 3, 4);       // (one unaligned 4-byte load; the array has a spare byte at its end)
  return (w & 0x7FFFFFu) | ((w & 0x800000u) << 8);   // chain bit 23 -> bit 31 (kChainBit)
}

__device__ __forceinline__ bool dense_row_in_filter(const uint64_t *filter, int64_t row) {
  return filter == nullptr || ((filter[row >> 6] >> (63 - (row & 63))) & 1u);
}

// A wave owns groups of kBuildR x 64 rows: filter words with one load per group, next group's keys requested before the
// current group's head words are claimed (the structure of dense_probe_kernel).
// kRuns: the build side is a run of blocks (qsx_join_build_blocks): a group belongs to one block and takes its key stripe,
// row count, filter and base tuple id from the run table.
constexpr int kBuildR = 8;
constexpr int kBuildTile = kBuildR * kWave;   // rows of a group
template <typename KeyT, bool kRuns = false>
__global__ __launch_bounds__(kDBlock) void dense_build_kernel(DenseTableView t, const KeyT *__restrict__ keys, int64_t n,
                                                             int32_t base_tid_arg, const uint64_t *__restrict__ filter,
                                                             unsigned long long *__restrict__ entries,
                                                             const long long *__restrict__ runs = nullptr) {
  constexpr int R = kBuildR;
  using Source = ProbeTileSource<KeyT>;
  const int lane = lane_id();
  const int64_t num_groups = kRuns ? runs[2] : (((n + 63) >> 6) + R - 1) / R;
  const int64_t wave = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * (kDBlock / kWave) + (threadIdx.x >> 6)));
  const int64_t num_waves = static_cast<int64_t>(gridDim.x) * (kDBlock / kWave);
  unsigned long long inserted = 0;
  KeyT key[R], next_key[R];
  uint64_t words = ~0ull, next_words = ~0ull;
  auto source_of = [&](int64_t group) {
    return probe_tile_source<KeyT, kBuildTile, kRuns>(runs, group, keys, n, base_tid_arg, filter, nullptr);
  };
  auto request = [&](const Source &src, KeyT (&k)[R], uint64_t &fw) {
    const int64_t sw0 = src.base >> 6;
    if (kRuns && src.code_width != 0) {   // a compressed key stripe (block_runs.hpp): read as it lies
      coded_keys(src, k, [&](int r) {
        const int64_t row = ((sw0 + r) << 6) + lane;
        return row < src.n ? row : src.n - 1;
      });
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t row = ((sw0 + r) << 6) + lane;
        k[r] = load_global(&src.keys[row < src.n ? row : src.n - 1]);   // clamped, not guarded: no branch around the read
      }
    }
    fw = ~0ull;
    if (src.filter != nullptr && lane < R && sw0 + lane < ((src.n + 63) >> 6)) fw = load_global(&src.filter[sw0 + lane]);
  };
  Source cur = Source(), next = Source();
  int64_t group = wave;
  if (group < num_groups) {
    cur = source_of(group);
    request(cur, key, words);
  }
  for (; group < num_groups; group += num_waves) {
    if (group + num_waves < num_groups) {
      next = source_of(group + num_waves);
      request(next, next_key, next_words);
    }
    const int64_t w0 = cur.base >> 6;
    const int64_t n = cur.n;               // (shadows the argument: the rows of this group's stripe)
    const int32_t base_tid = cur.base_tid;
    cur = next;
    // the R compare-and-swaps of a group are all issued before the first result is looked at
    uint64_t idx[R];
    uint32_t old[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t i = ((w0 + r) << 6) + lane;
      const uint64_t fw = __shfl(words, r, kWave);   // before any branch: every lane takes part
      const bool live = i < n && msb_bit(fw, lane);
      idx[r] = live ? dense_index(t, key[r]) : ~1ull;     // ~0: outside the range (an error), ~1: dead row
      if (idx[r] == ~0ull) atomicExch(t.error, 1);
      old[r] = 0u;
      if (idx[r] < ~1ull) old[r] = atomicCAS(&t.head[idx[r]], 0u, static_cast<uint32_t>(base_tid + i) + 1u);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (idx[r] >= ~1ull) continue;
      const int64_t i = ((w0 + r) << 6) + lane;
      const uint32_t tid = static_cast<uint32_t>(base_tid + i);
      if (old[r] != 0u) {
        // duplicate key: push an overflow entry in front of whatever the head holds now
        const unsigned int e = atomicAdd(t.ov_count, 1u);
        if (e >= t.ov_capacity) {
          atomicExch(t.error, 2);
          continue;
        }
        t.ov[e].x = tid;
        // next is only read by probe kernels launched after the build (pipeline breaker)
        t.ov[e].y = atomicExch(&t.head[idx[r]], kChainBit | e);
      }
      ++inserted;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) key[r] = next_key[r];
    words = next_words;
  }
  inserted = wave_reduce_add(inserted);
  // (the wave's copy of the control words, join.hip kControlReplicas x kControlStride: not one address for every wave)
  if (lane == 0 && inserted != 0) atomicAdd(entries + static_cast<size_t>(wave & 31) * 16, inserted);
}

// ---- probe + projection in one pass (qsx_join_probe_project_blocks) ------------------------------------------------------
// The output relation of an inner join — the reference's HashInnerJoinWorkOrder materialises it from the (probe, build)
// tuple-id pairs it collected (HashJoinOperator.cpp:494-560) — written by the probe itself: a matching lane reads the
// projected attributes of its probe row (coalesced: lanes are consecutive rows) and of the build tuple it found (random)
// and stores them at the output position it would have stored the pair at.  The pair list (8 bytes written and read
// again per match) and one kernel per attribute go away.
// Device table (8-byte words), kProjColumnWords arrays of nc words first: width | on_build | output stripe | byte offset of the
// column inside an entry of the covering array (build-side columns, when there is one) | 1 when the column is the probe key
// itself (its stripe in every block is the block's key stripe: the value is in a register already); then first tuple id of build segment
// s [nseg] | stripe of column c in build segment s [s * nc + c] | stripe of column c in probe block b [b * nc + c].
//
// Covering array (MODE 6 / 7 / 8: entries of 4 / 8 / 16 bytes): the projected build-side values of the tuple under key value
// k, packed into one entry at cover[k - min_key]; all bits set = no tuple.  Built once per (table, projection) by
// cover_build_kernel when the build keys are unique; the probe then reads ONE random entry per row — not head[] and then the
// attribute stripes: both together (3 + 4 MiB for a million keys and one INT attribute) do not share an XCD's L2.
constexpr int kProjColumnWords = 5;
struct ProjectionView {
  const long long *table;
  int nc;
  int nseg;
  int seg_rows;   // > 0: every build segment but the last holds this many tuples — segment of a tuple id by division
  const void *cover;
  __device__ __forceinline__ int width(int c) const { return static_cast<int>(table[c]); }
  __device__ __forceinline__ bool on_build(int c) const { return table[nc + c] != 0; }
  __device__ __forceinline__ char *out(int c) const { return as_global(reinterpret_cast<char *>(table[2 * nc + c])); }
  __device__ __forceinline__ int cover_offset(int c) const { return static_cast<int>(table[3 * nc + c]); }
  __device__ __forceinline__ bool is_probe_key(int c) const { return table[4 * nc + c] != 0; }
  __device__ __forceinline__ const long long *first_tids() const { return table + kProjColumnWords * nc; }
  __device__ __forceinline__ const char *build_stripe(int seg, int c) const {
    return as_global(reinterpret_cast<const char *>(table[kProjColumnWords * nc + nseg + seg * nc + c]));
  }
  __device__ __forceinline__ const char *probe_stripe(int block, int c) const {
    return as_global(reinterpret_cast<const char *>(table[kProjColumnWords * nc + nseg + nseg * nc + block * nc + c]));
  }
};
__device__ __forceinline__ unsigned long long load_value(const char *src, int width) {
  switch (width) {
    case 1: return load_global(reinterpret_cast<const uint8_t *>(src));
    case 2: return load_global(reinterpret_cast<const uint16_t *>(src));
    case 4: return load_global(reinterpret_cast<const uint32_t *>(src));
    default: return load_global(reinterpret_cast<const unsigned long long *>(src));
  }
}
__device__ __forceinline__ void store_value(char *dst, unsigned long long v, int width) {
  switch (width) {
    case 1: store_global(static_cast<uint8_t>(v), reinterpret_cast<uint8_t *>(dst)); break;
    case 2: store_global(static_cast<uint16_t>(v), reinterpret_cast<uint16_t *>(dst)); break;
    case 4: store_global_nt(static_cast<uint32_t>(v), reinterpret_cast<uint32_t *>(dst)); break;
    default: store_global_nt(v, reinterpret_cast<unsigned long long *>(dst)); break;
  }
}
// Where column c of build tuple `tid` lives.
__device__ __forceinline__ const char *projected_build_value(const ProjectionView &p, int c, int width, uint32_t tid) {
  const long long *first = p.first_tids();
  int seg = 0;
  if (p.nseg > 1) {
    if (p.seg_rows > 0) {
      seg = static_cast<int>((tid - static_cast<uint32_t>(first[0])) / static_cast<uint32_t>(p.seg_rows));
      seg = seg < p.nseg ? seg : p.nseg - 1;
    } else {
      int lo = 0, hi = p.nseg - 1;   // last segment whose first tuple id <= tid
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first[mid] <= static_cast<long long>(tid)) lo = mid; else hi = mid - 1;
      }
      seg = lo;
    }
  }
  return p.build_stripe(seg, c) + static_cast<size_t>(tid - static_cast<uint32_t>(first[seg])) * width;
}
// One projected tuple (the duplicate chains' path: one match at a time).
// (tile: the probe tile's source — a projected column that is the coded key stripe itself comes out as the key's value)
template <typename KeyT>
__device__ __forceinline__ void project_one(const ProjectionView &p, const ProbeTileSource<KeyT> &tile, int64_t probe_row, uint32_t tid,
                                            unsigned long long o) {
  for (int c = 0; c < p.nc; ++c) {
    const int width = p.width(c);
    if (!p.on_build(c) && tile.code_width != 0 && p.is_probe_key(c)) {
      store_value(p.out(c) + o * width, static_cast<unsigned long long>(coded_key(tile, probe_row)), width);
      continue;
    }
    const char *src = p.on_build(c) ? projected_build_value(p, c, width, tid) : p.probe_stripe(tile.block, c) + static_cast<size_t>(probe_row) * width;
    store_value(p.out(c) + o * width, load_value(src, width), width);
  }
}

// The covering array of a projection: entry i = the build-side values of the tuple whose key is min_key + i.
// flags: bit 0 = some key has several tuples (no covering array for this table), bit 1 = a real entry came out all-ones
// (indistinguishable from "no tuple").
template <typename EntryT>
__global__ __launch_bounds__(kDBlock) void cover_build_kernel(DenseTableView t, ProjectionView p, EntryT *__restrict__ cover,
                                                             unsigned int *__restrict__ flags) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kDBlock + threadIdx.x;
  if (i >= t.range) return;
  const uint32_t hw = t.head[i];
  unsigned long long words[2] = {~0ull, ~0ull};
  if (hw & kChainBit) {
    atomicOr(flags, 1u);
  } else if (hw != 0u) {
    words[0] = words[1] = 0ull;
    for (int c = 0; c < p.nc; ++c) {
      if (!p.on_build(c)) continue;
      const int width = p.width(c), off = p.cover_offset(c);
      const unsigned long long v = load_value(projected_build_value(p, c, width, hw - 1u), width);
      if (off < 8) words[0] |= v << (8 * off); else words[1] |= v << (8 * (off - 8));
    }
    const bool ones = sizeof(EntryT) == 4 ? static_cast<uint32_t>(words[0]) == ~0u
                                          : (sizeof(EntryT) == 8 ? words[0] == ~0ull : (words[0] & words[1]) == ~0ull);
    if (ones) atomicOr(flags, 2u);
  }
  if constexpr (sizeof(EntryT) == 4) {
    cover[i] = static_cast<uint32_t>(words[0]);
  } else if constexpr (sizeof(EntryT) == 8) {
    cover[i] = words[0];
  } else {
    cover[i] = EntryT{words[0], words[1]};
  }
}

// One wave-aggregated append of the matching lanes straight to the global output.
__device__ __forceinline__ void dense_emit_direct(bool match, int32_t probe_tid, int32_t build_tid,
                                                  int32_t *__restrict__ out_probe, int32_t *__restrict__ out_build,
                                                  unsigned long long capacity, unsigned long long *out_count) {
  const uint64_t m = __ballot(match);
  if (m == 0) return;
  unsigned long long base = 0;
  if (lane_id() == __ffsll(static_cast<long long>(m)) - 1) {
    base = atomicAdd(out_count, static_cast<unsigned long long>(__popcll(m)));
  }
  base = __shfl(base, __ffsll(static_cast<long long>(m)) - 1, kWave);
  const unsigned long long o = base + rank_below(m);
  if (match && o < capacity) {
    __builtin_nontemporal_store(probe_tid, &out_probe[o]);
    __builtin_nontemporal_store(build_tid, &out_build[o]);
  }
}

// MODE 0: emit pairs (one reservation per tile on the output counter), 1: count only, 2: existence bitmap (as
// probe_kernel in join.hip); 3 + 4: the two-pass form of 0 — 3 counts the matches of every (tile, wave) unit, a scan turns
// the counts into offsets, 4 writes every unit's pairs at its offset.  No atomics on the output counter (which takes
// ~12 ns each: with sparse matches over clustered keys the one-pass form is bound by them, 146 K tiles = 1.7 ms for
// 600 M rows) and the pairs come out in probe-row order.
//
// MODE 5: MODE 0 with the projected tuples written instead of the pairs (ProjectionView above).
// (with a covering array of the build-side columns: cover_probe_kernel below)
//
// kRuns: the probe side is a run of blocks (qsx_join_probe_blocks, block_runs.hpp) — `runs` is the table, a tile belongs to
// one block and takes that block's key stripe, row count, filter, base tuple id and (MODE 2) output bitmap; the output
// pair list and its counter are the run's.
template <typename KeyT, int MODE, bool kRuns = false>
__global__ __launch_bounds__(kDBlock) __attribute__((amdgpu_waves_per_eu(MODE == 5 ? 3 : 1))) void dense_probe_kernel(
    DenseTableView t, const KeyT *__restrict__ keys, int64_t n, int32_t probe_base_tid,
    const uint64_t *__restrict__ filter, int32_t *__restrict__ out_probe, int32_t *__restrict__ out_build,
    int64_t capacity_signed, unsigned long long *__restrict__ out_count, uint64_t *__restrict__ out_bitmap, int anti,
    int32_t *__restrict__ unit_counts = nullptr, const int64_t *__restrict__ unit_offsets = nullptr,
    const long long *__restrict__ runs = nullptr, ProjectionView proj = ProjectionView{}) {
  constexpr int R = kDenseRowsPerThread;
  constexpr bool kProject = MODE == 5;                  // emits projected tuples
  using Source = ProbeTileSource<KeyT>;
  const unsigned long long capacity = static_cast<unsigned long long>(capacity_signed);
  const int64_t num_tiles = kRuns ? runs[2] : (n + kDenseTile - 1) / kDenseTile;
  auto source_of = [&](int64_t tile) {
    return probe_tile_source<KeyT, kDenseTile, kRuns>(runs, tile, keys, n, probe_base_tid, filter, out_bitmap);
  };
  unsigned long long local_count = 0;
  __shared__ int2 s_sparse[(MODE == 0 || MODE == 4) ? kDBlock / kWave : 1][(MODE == 0 || MODE == 4) ? kDenseSparsePairs : 1];
  __shared__ int s_wave_total[2][kDBlock / kWave];
  __shared__ unsigned long long s_tile_base;
  int parity = 0;
  const int wave = threadIdx.x >> 6;

  // The keys and the filter words of the NEXT tile are requested before the head words of the current one are read:
  // one memory round trip per tile instead of filter -> keys -> head.  The 16 filter words a wave needs for a tile
  // (rows r * 256 + wave * 64 ..) come with one load, lane r holding word r.
  const int lane = lane_id();
  KeyT key[R], next_key[R];
  uint64_t filter_words = ~0ull, next_filter_words = ~0ull;
  auto request = [&](const Source &src, KeyT (&k)[R], uint64_t &words) {
    if (kRuns && src.code_width != 0) {   // a compressed key stripe (block_runs.hpp): read as it lies
      coded_keys(src, k, [&](int r) {
        const int64_t row = src.base + r * kDBlock + threadIdx.x;
        return row < src.n ? row : src.n - 1;
      });
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t row = src.base + r * kDBlock + threadIdx.x;
        k[r] = load_global_nt(&src.keys[row < src.n ? row : src.n - 1]);   // clamped, not guarded
      }
    }
    words = ~0ull;
    if (src.filter != nullptr && lane < R) {
      const int64_t w = (src.base >> 6) + lane * (kDBlock / kWave) + wave;
      if (w < ((src.n + 63) >> 6)) words = load_global(&src.filter[w]);
    }
  };
  Source cur = Source(), next = Source();
  if (static_cast<int64_t>(blockIdx.x) < num_tiles) {
    cur = source_of(blockIdx.x);
    request(cur, key, filter_words);
  }

  for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x, parity ^= 1) {
    if (tile + gridDim.x < num_tiles) {
      next = source_of(tile + gridDim.x);
      request(next, next_key, next_filter_words);
    }
    const int64_t tile_base = cur.base;
    const int64_t n_rows = cur.n;
    const int32_t base_tid = cur.base_tid;
    uint64_t *const tile_bitmap = cur.out_bitmap;
    const int tile_block = cur.block;
    const Source tile_src = cur;   // (MODE 5: a projected column that IS a coded key stripe is read through the tile's coding)
    cur = next;
    // Row r of this thread: tile_base + r * 256 + tid; a wave owns 64 consecutive rows per r (the workgroup reads 1 KiB
    // contiguous per step: the wave-contiguous mapping measured 8 % slower).
    uint32_t h[R];
    uint32_t live_mask = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {  // R independent 4-byte reads in flight per lane
      const int64_t row = tile_base + r * kDBlock + threadIdx.x;
      const uint64_t filter_word = __shfl(filter_words, r, kWave);   // before the branch: every lane must take part
      const bool live = row < n_rows && msb_bit(filter_word, lane);
      live_mask |= live ? (1u << r) : 0u;
      const uint64_t idx = dense_index(t, key[r]);
      // unconditional read (dead lanes read word 0): a guarded read compiles to branch + load + wait per step, which
      // serialises the 16 reads of a tile (seen in the existence variant: 1.45 ms instead of 0.6 ms per 600 M rows)
      const bool lookup = live && idx != ~0ull;
      const uint32_t word = dense_head_word(t, lookup ? idx : 0);
      h[r] = lookup ? word : 0u;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) key[r] = next_key[r];
    filter_words = next_filter_words;

    if (MODE == 2) {
      uint64_t mine = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        // live is folded into h (dead rows read 0); anti needs it back
        const bool bit = ((live_mask >> r) & 1u) && ((h[r] != 0u) != (anti != 0));
        const uint64_t word = msb_first(__ballot(bit));
        if (lane == r) mine = word;
        if (lane == 0) local_count += __popcll(word);
      }
      // lane r holds the word of step r: one store instruction per tile and wave
      const int64_t w = (tile_base >> 6) + lane * (kDBlock / kWave) + wave;
      if (lane < R && w < ((n_rows + 63) >> 6)) store_global(mine, &tile_bitmap[w]);
      continue;
    }

    // ---- first level: at most one match per row ---------------------------------------
    uint64_t m[R];
    int total = 0;
    bool any_chain = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      m[r] = __ballot(h[r] != 0u);
      total += __popcll(m[r]);
      any_chain = any_chain || (h[r] & kChainBit) != 0u;
    }
    const bool wave_has_chain = __any(any_chain);
    uint32_t next[R];
    unsigned long long base = 0;   // MODE 0 / 4: where this wave's next pair goes
    if (MODE == 1 || MODE == 3) {
      if (lane == 0) local_count += total;
      if (wave_has_chain) {
#pragma unroll
        for (int r = 0; r < R; ++r) next[r] = (h[r] & kChainBit) ? t.ov[h[r] & ~kChainBit].y : 0u;
      }
    } else {
      if (MODE == 0 || kProject) {
        // s_wave_total is double-buffered by tile parity, s_tile_base is rewritten only after the next
        // tile's first barrier: two barriers per tile suffice.
        if (lane == 0) s_wave_total[parity][wave] = total;
        __syncthreads();
        if (threadIdx.x == 0) {
          int all = 0;
#pragma unroll
          for (int w = 0; w < kDBlock / kWave; ++w) all += s_wave_total[parity][w];
          s_tile_base = all != 0 ? atomicAdd(out_count, static_cast<unsigned long long>(all)) : 0ull;
        }
        __syncthreads();
        base = s_tile_base;
        for (int w = 0; w < wave; ++w) base += s_wave_total[parity][w];
      } else {
        base = static_cast<unsigned long long>(unit_offsets[tile * (kDBlock / kWave) + wave]);   // from the counting pass
      }
      // Few matches in the wave's 1024 rows (a selective filter in front of the probe): 16 steps of stores with a
      // handful of active lanes each would write 16-byte fragments.  The pairs meet in a wave-private LDS strip and
      // leave as full-wave contiguous stores.
      const bool sparse = !kProject && total <= kDenseSparsePairs;   // wave-uniform
      int staged = 0;
      if (kProject) {
        // positions and build tuple ids of the first-level matches, then column by column: the R reads of a column are all
        // issued before its first store (a store between them would order them: the compiler cannot tell the stripes apart)
        // (registers: h[r] becomes the matched build tuple id + 1, the output position a 32-bit offset from the wave's base)
        int off[R];
        int emitted = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          next[r] = 0u;
          if (wave_has_chain && (h[r] & kChainBit)) {
            const uint2 e = t.ov[h[r] & ~kChainBit];
            h[r] = e.x + 1u;
            next[r] = e.y;
          }
          off[r] = emitted + rank_below(m[r]);
          emitted += __popcll(m[r]);
        }
        const unsigned long long wave_base = base;
        base += emitted;
        for (int c = 0; c < proj.nc; ++c) {
          const int width = proj.width(c);
          const bool on_build = proj.on_build(c);
          char *dst = proj.out(c);
          const char *probe_stripe = proj.probe_stripe(tile_block, c);   // (0 for a build-side column: never read)
          // (the width is chosen once per column, outside the R reads: a switch per read puts every read into a basic block of
          // its own and the tile pays R memory round trips per column instead of one — 2.1 instead of 0.8 ms per 100 M rows)
          auto column = [&](auto tag) __attribute__((always_inline)) {
            using V = decltype(tag);
            constexpr int H = R / 2;   // (two half tiles: 16 values and their addresses in registers cost two waves per SIMD)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              V v[H];
              if (on_build) {
#pragma unroll
                for (int i = 0; i < H; ++i) {
                  const int r = half * H + i;
                  // unconditional read, no branch around it: rows without a match read the first word of the output stripe
                  const char *src = h[r] != 0u ? projected_build_value(proj, c, sizeof(V), h[r] - 1u) : dst;
                  v[i] = load_global(reinterpret_cast<const V *>(src));
                }
              } else if (kRuns && sizeof(V) == sizeof(KeyT) && tile_src.code_width != 0 && proj.is_probe_key(c)) {
                // the join attribute itself over a block that holds it compressed (qsx_join_probe_project_blocks_coded): the
                // column's stripe is the key's code stripe, the value what the code stands for
#pragma unroll
                for (int i = 0; i < H; ++i) {
                  const int64_t row = tile_base + (half * H + i) * kDBlock + threadIdx.x;
                  v[i] = static_cast<V>(coded_key(tile_src, row < n_rows ? row : n_rows - 1));
                }
              } else {
#pragma unroll
                for (int i = 0; i < H; ++i) {
                  const int64_t row = tile_base + (half * H + i) * kDBlock + threadIdx.x;
                  v[i] = load_global_nt(&reinterpret_cast<const V *>(probe_stripe)[row < n_rows ? row : n_rows - 1]);
                }
              }
#pragma unroll
              for (int i = 0; i < H; ++i) {
                const int r = half * H + i;
                const unsigned long long o = wave_base + off[r];
                if (h[r] != 0u && o < capacity) store_global_nt(v[i], reinterpret_cast<V *>(dst) + o);
              }
            }
          };
          switch (width) {
            case 1: column(uint8_t{}); break;
            case 2: column(uint16_t{}); break;
            case 4: column(uint32_t{}); break;
            default: column(static_cast<unsigned long long>(0)); break;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (kProject) break;
        const int64_t row = tile_base + r * kDBlock + threadIdx.x;
        uint32_t tid = h[r] - 1u;
        next[r] = 0u;
        if (wave_has_chain && (h[r] & kChainBit)) {
          const uint2 e = t.ov[h[r] & ~kChainBit];
          tid = e.x;
          next[r] = e.y;
        }
        if (sparse) {
          if (m[r] != 0) {
            if (h[r] != 0u) s_sparse[wave][staged + rank_below(m[r])] = make_int2(static_cast<int32_t>(base_tid + row), static_cast<int32_t>(tid));
            staged += __popcll(m[r]);
          }
        } else {
          const unsigned long long o = base + rank_below(m[r]);
          if (h[r] != 0u && o < capacity) {
            __builtin_nontemporal_store(static_cast<int32_t>(base_tid + row), &out_probe[o]);
            __builtin_nontemporal_store(static_cast<int32_t>(tid), &out_build[o]);
          }
          base += __popcll(m[r]);
        }
      }
      if (sparse) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = lane; i < total; i += kWave) {
          const int2 pair = s_sparse[wave][i];
          const unsigned long long o = base + i;
          if (o < capacity) {
            __builtin_nontemporal_store(pair.x, &out_probe[o]);
            __builtin_nontemporal_store(pair.y, &out_build[o]);
          }
        }
        base += total;
        __builtin_amdgcn_wave_barrier();   // the strip is rewritten by the next tile
      }
    }
    // ---- duplicate build keys: walk the chains ------------------------------------------
    if (wave_has_chain) {
#pragma unroll 1
      for (int r = 0; r < R; ++r) {
        const int64_t row = tile_base + r * kDBlock + threadIdx.x;
        uint32_t cur = next[r];
        while (__any(cur != 0u)) {
          uint32_t tid = cur - 1u, nxt = 0u;
          if (cur & kChainBit) {
            const uint2 e = t.ov[cur & ~kChainBit];
            tid = e.x;
            nxt = e.y;
          }
          if (MODE == 1 || MODE == 3) {
            local_count += cur != 0u ? 1u : 0u;
          } else if (MODE == 0) {
            dense_emit_direct(cur != 0u, static_cast<int32_t>(base_tid + row), static_cast<int32_t>(tid), out_probe,
                              out_build, capacity, out_count);
          } else if (kProject) {
            const uint64_t cm = __ballot(cur != 0u);
            unsigned long long at = 0;
            const int leader = __ffsll(static_cast<long long>(cm)) - 1;
            if (lane == leader) at = atomicAdd(out_count, static_cast<unsigned long long>(__popcll(cm)));
            at = __shfl(at, leader, kWave) + rank_below(cm);
            if (cur != 0u && at < capacity) project_one(proj, tile_src, row, tid, at);
          } else {   // the wave's run continues behind its first-level matches, in the order the counting pass saw
            const uint64_t cm = __ballot(cur != 0u);
            const unsigned long long o = base + rank_below(cm);
            if (cur != 0u && o < capacity) {
              __builtin_nontemporal_store(static_cast<int32_t>(base_tid + row), &out_probe[o]);
              __builtin_nontemporal_store(static_cast<int32_t>(tid), &out_build[o]);
            }
            base += __popcll(cm);
          }
          cur = nxt;
        }
      }
    }
    if (MODE == 3) {   // matches of this wave in this tile, for the scan between the two passes
      const unsigned long long unit = wave_reduce_add(local_count);
      if (lane == 0) unit_counts[tile * (kDBlock / kWave) + wave] = static_cast<int32_t>(unit);
      local_count = 0;
    }
  }
  if (MODE == 1 || MODE == 2) {
    local_count = wave_reduce_add(local_count);
    if (lane == 0 && local_count != 0 && out_count != nullptr) atomicAdd(out_count, local_count);
  }
}

// Probe + projection through the covering array (ProjectionView::cover), over a run of probe blocks: one random read per
// probe row brings the match AND the build-side values.  Tiles, reservation (one atomic on the output counter per tile) and
// output order as dense_probe_kernel MODE 0; unique build keys by construction (cover_build_kernel), so no chains.
// BLOCK threads take a tile of kDenseTile rows, kDenseTile / BLOCK rows per thread: launched as 512 x 8 instead of the pair
// kernel's 256 x 16 — the entries and values of 16 rows per thread cost 180 registers, i.e. two workgroups per CU and most of a
// tile's dependent memory round trips exposed.
template <typename KeyT, typename CoverT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void cover_probe_kernel(DenseTableView t, int64_t capacity_signed,
                                                             unsigned long long *__restrict__ out_count,
                                                             const long long *__restrict__ runs, ProjectionView proj) {
  constexpr int R = kDenseTile / BLOCK;
  static_assert(R * BLOCK == kDenseTile && R <= kWave, "a tile is R steps of BLOCK rows; lane r holds the filter word of step r");
  const unsigned long long capacity = static_cast<unsigned long long>(capacity_signed);
  const int64_t num_tiles = runs[2];
  __shared__ int s_wave_total[2][BLOCK / kWave];
  __shared__ unsigned long long s_tile_base;
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const CoverT *cover = reinterpret_cast<const CoverT *>(proj.cover);
  int parity = 0;
  for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x, parity ^= 1) {
    const ProbeTileSource<KeyT> src = probe_tile_source<KeyT, kDenseTile, true>(runs, tile, nullptr, 0, 0, nullptr, nullptr);
    uint64_t filter_words = ~0ull;
    if (src.filter != nullptr && lane < R) {
      const int64_t w = (src.base >> 6) + lane * (BLOCK / kWave) + wave;
      if (w < ((src.n + 63) >> 6)) filter_words = load_global(&src.filter[w]);
    }
    CoverT e[R];
    KeyT key[R];   // (kept for projected columns that ARE the probe key)
    {
      if (src.code_width != 0) {   // a compressed key stripe (block_runs.hpp): read as it lies
        coded_keys(src, key, [&](int r) {
          const int64_t row = src.base + r * BLOCK + threadIdx.x;
          return row < src.n ? row : src.n - 1;
        });
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int64_t row = src.base + r * BLOCK + threadIdx.x;
          key[r] = load_global_nt(&src.keys[row < src.n ? row : src.n - 1]);
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {   // R independent reads in flight per lane; dead rows read entry 0 and are masked below
        const uint64_t idx = dense_index(t, key[r]);
        if constexpr (sizeof(CoverT) == 16) {
          using Pair = unsigned long long __attribute__((ext_vector_type(2)));   // (one 16-byte load; a class type cannot be read through an address-space pointer)
          const Pair both = load_global(reinterpret_cast<const Pair *>(&cover[idx != ~0ull ? idx : 0]));
          e[r] = CoverT{both.x, both.y};
        } else {
          e[r] = load_global(&cover[idx != ~0ull ? idx : 0]);
        }
        if (idx == ~0ull) {
          if constexpr (sizeof(CoverT) == 16) e[r] = CoverT{~0ull, ~0ull}; else e[r] = static_cast<CoverT>(~static_cast<CoverT>(0));
        }
      }
    }
    uint64_t m[R];
    int total = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = src.base + r * BLOCK + threadIdx.x;
      const uint64_t filter_word = __shfl(filter_words, r, kWave);
      bool present;
      if constexpr (sizeof(CoverT) == 16) present = (e[r].x & e[r].y) != ~0ull; else present = e[r] != static_cast<CoverT>(~static_cast<CoverT>(0));
      m[r] = __ballot(present && row < src.n && msb_bit(filter_word, lane));
      total += __popcll(m[r]);
    }
    if (lane == 0) s_wave_total[parity][wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
      int all = 0;
#pragma unroll
      for (int w = 0; w < BLOCK / kWave; ++w) all += s_wave_total[parity][w];
      s_tile_base = all != 0 ? atomicAdd(out_count, static_cast<unsigned long long>(all)) : 0ull;
    }
    __syncthreads();
    unsigned long long base = s_tile_base;
    for (int w = 0; w < wave; ++w) base += s_wave_total[parity][w];
    if (total == 0) continue;   // (wave-uniform; the barriers are behind us)
    // Per row and column-invariant, in 32 bits (what the compiler hoists out of the column loop stays in registers: 64-bit
    // positions and row numbers for 16 rows were 64 of them): the row's offset from the wave's first output tuple, its
    // offset from the tile's first row, and one bit "matched and inside the capacity".
    int off[R];
    unsigned int emit_mask = 0;
    {
      int at = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        off[r] = at + rank_below(m[r]);
        if (((m[r] >> lane) & 1ull) && base + static_cast<unsigned long long>(off[r]) < capacity) emit_mask |= 1u << r;
        at += __popcll(m[r]);
      }
    }
    const int tile_rows = static_cast<int>(src.n - src.base < kDenseTile ? src.n - src.base : kDenseTile);
    for (int c = 0; c < proj.nc; ++c) {
      const int width = proj.width(c);
      const bool on_build = proj.on_build(c);
      const int shift = 8 * proj.cover_offset(c);
      const bool from_key = proj.is_probe_key(c);   // (host: only for columns as wide as the key)
      auto column = [&](auto tag) __attribute__((always_inline)) {
        using V = decltype(tag);
        constexpr int H = R / 2;
        // (wave-uniform; as_global at the point of use: a pointer that went through a select with nullptr or a struct is
        // generic again and its accesses become flat_load / flat_store)
        V *tile_dst = reinterpret_cast<V *>(proj.out(c)) + base;
        const V *tile_src = reinterpret_cast<const V *>(proj.probe_stripe(src.block, c)) + src.base;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          V v[H];
#pragma unroll
          for (int i = 0; i < H; ++i) {
            const int r = half * H + i;
            if (on_build) {
              if constexpr (sizeof(CoverT) == 16) {
                v[i] = static_cast<V>(shift < 64 ? e[r].x >> shift : e[r].y >> (shift - 64));
              } else {
                v[i] = static_cast<V>(static_cast<unsigned long long>(e[r]) >> shift);
              }
            } else if (from_key && sizeof(V) == sizeof(KeyT)) {
              v[i] = static_cast<V>(key[r]);
            } else {
              const int in_tile = r * BLOCK + static_cast<int>(threadIdx.x);
              v[i] = load_global_nt(&tile_src[in_tile < tile_rows ? in_tile : tile_rows - 1]);   // (a stream: keep it out of the covering array's L2)
            }
          }
#pragma unroll
          for (int i = 0; i < H; ++i) {
            const int r = half * H + i;
            if ((emit_mask >> r) & 1u) store_global_nt(v[i], &tile_dst[off[r]]);
          }
        }
      };
      switch (width) {
        case 1: column(uint8_t{}); break;
        case 2: column(uint16_t{}); break;
        case 4: column(uint32_t{}); break;
        default: column(static_cast<unsigned long long>(0)); break;
      }
    }
  }
}

}  // namespace qsx

#endif  // QSX_CSRC_JOIN_DENSE_HPP_
