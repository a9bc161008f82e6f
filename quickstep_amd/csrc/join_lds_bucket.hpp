// join_lds_bucket.hpp — K4 against the BUCKETED join table (join.hip: 16-slot buckets behind a fingerprint plane) copied into
// LDS: the form join_lds.hpp's header announces for build sides WITHOUT a dense key domain (sparse keys, composite keys
// packed into a LONG).  Included by join.hip behind the table's definitions (TableView, home_bucket, fingerprint,
// bucket_masks, SlotOf).
//
// Reference loop: HashTable::getAllFromValueAccessorImpl (storage/HashTable.hpp:2145-2181) over getNextEntryForKey
// (storage/SimpleScalarSeparateChainingHashTable.hpp:751-781); composite keys :1835-1880.
//
// LDS: [fingerprint plane: 16 B per bucket][slots: 16 x 8 B (INT) or 16 x 16 B (LONG) per bucket] — the device table byte for
// byte, so every rule of the global kernel holds (a probe reads its home bucket's fingerprint word, touches slots only
// under a matching fingerprint, walks on only from a bucket without an empty byte, and ends at its first match when the
// table's duplicate flag is clear).  Up to 144 KiB: 1024 buckets of INT keys (~13 K keys at load 0.8), 512 of LONG keys.
// Work distribution, reservation and emission as lds_dense_probe_kernel: 1024 threads, super tiles of 4 x 4096 rows, the
// FIRST match of every row goes out through the per-unit reservation; further matches of a row (duplicate build keys) are
// appended wave by wave.
#ifndef QSX_CSRC_JOIN_LDS_BUCKET_HPP_
#define QSX_CSRC_JOIN_LDS_BUCKET_HPP_

namespace qsx {

constexpr size_t kLdsBucketMaxBytes = 144 * 1024;

template <typename KeyT, int MODE, bool kRuns = false>
__global__ __launch_bounds__(kLdsBlock) void lds_bucket_probe_kernel(TableView t, const KeyT *__restrict__ keys, int64_t n,
                                                                    int32_t probe_base_tid, const uint64_t *__restrict__ filter,
                                                                    int32_t *__restrict__ out_probe, int32_t *__restrict__ out_build,
                                                                    int64_t capacity_signed, unsigned long long *__restrict__ out_count,
                                                                    uint64_t *__restrict__ out_bitmap, int anti,
                                                                    const long long *__restrict__ runs = nullptr) {
  constexpr int BLOCK = kLdsBlock, S = kLdsSub;
  constexpr int R = kLdsTile / BLOCK;
  constexpr int kWaves = BLOCK / kWave;
  using Slot = SlotOf<KeyT>;
  using Raw = typename Slot::Raw;
  using Source = ProbeTileSource<KeyT>;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ u32x4 l_table[];          // fingerprint words, then the slots
  __shared__ int s_wave_total[2][kWaves];
  __shared__ unsigned long long s_tile_base;
  const unsigned long long capacity = static_cast<unsigned long long>(capacity_signed);
  const int64_t num_tiles = kRuns ? runs[2] : (n + kLdsTile - 1) / kLdsTile;
  const int64_t num_super = (num_tiles + S - 1) / S;
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int buckets = static_cast<int>(t.buckets);
  const u32x4 *l_fp = l_table;
  const Raw *l_slots = reinterpret_cast<const Raw *>(l_table + buckets);
  const bool unique = sizeof(KeyT) == 4 && *t.dup_flag == 0u;

  auto source_of = [&](int64_t tile) {
    return probe_tile_source<KeyT, kLdsTile, kRuns>(runs, tile, keys, n, probe_base_tid, filter, out_bitmap);
  };
  KeyT key[S][R], next_key[S][R];
  uint64_t filter_words[S], next_filter_words[S];
  auto request = [&](int64_t super, KeyT (&k)[S][R], uint64_t (&words)[S]) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int64_t tile = super * S + s;
      words[s] = ~0ull;
      if (tile >= num_tiles) {
#pragma unroll
        for (int r = 0; r < R; ++r) k[s][r] = KeyT(0);
        continue;
      }
      const Source src = source_of(tile);
      if (kRuns && src.code_width != 0) {   // a compressed key stripe (block_runs.hpp): read as it lies
        coded_keys(src, k[s], [&](int r) {
          const int64_t row = src.base + r * BLOCK + threadIdx.x;
          return row < src.n ? row : src.n - 1;
        });
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int64_t row = src.base + r * BLOCK + threadIdx.x;
          k[s][r] = load_global_nt(&src.keys[row < src.n ? row : src.n - 1]);
        }
      }
      if (src.filter != nullptr && lane < R) {
        const int64_t w = (src.base >> 6) + lane * kWaves + wave;
        if (w < ((src.n + 63) >> 6)) words[s] = load_global(&src.filter[w]);
      }
    }
  };
  if (static_cast<int64_t>(blockIdx.x) < num_super) request(blockIdx.x, key, filter_words);
  {
    const u32x4 *fp_src = reinterpret_cast<const u32x4 *>(t.fp);
    for (int i = threadIdx.x; i < buckets; i += BLOCK) l_table[i] = load_global(&fp_src[i]);
    const int slot_words = static_cast<int>(static_cast<size_t>(buckets) * kBucketSlots * sizeof(Raw) / 16);
    const u32x4 *slot_src = reinterpret_cast<const u32x4 *>(t.slots);
    for (int i = threadIdx.x; i < slot_words; i += BLOCK) l_table[buckets + i] = load_global(&slot_src[i]);
  }
  __syncthreads();

  unsigned long long local_count = 0;
  int parity = 0;
  for (int64_t super = blockIdx.x; super < num_super; super += gridDim.x, parity ^= 1) {
    if (super + gridDim.x < num_super) request(super + gridDim.x, next_key, next_filter_words);
    uint32_t h[S][R];            // tuple id + 1 of the row's FIRST match, 0 = none
    uint32_t live_mask = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int64_t tile = super * S + s;
      const bool present = tile < num_tiles;
      const Source src = present ? source_of(tile) : Source();
      const int64_t n_rows = present ? src.n : 0;
      // the fingerprint words and the first slots of the sub-tile's R rows: independent ds_reads, issued together
      uint32_t bucket[R], masks[R];
      bool live[R];
      {
        u32x4 w[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int64_t row = src.base + r * BLOCK + threadIdx.x;
          const uint64_t filter_word = __shfl(filter_words[s], r, kWave);   // before any branch: every lane takes part
          live[r] = row < n_rows && msb_bit(filter_word, lane);
          live_mask |= live[r] ? (1u << (s * R + r)) : 0u;
          bucket[r] = static_cast<uint32_t>(home_bucket(key[s][r], t));
          w[r] = l_fp[live[r] ? bucket[r] : 0u];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          masks[r] = live[r] ? bucket_masks(uint4{w[r].x, w[r].y, w[r].z, w[r].w}, fingerprint(key[s][r])) : 0x10000u;   // dead: nothing, "empty seen"
        }
      }
      Raw first[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t same = masks[r] & 0xFFFFu;
        first[r] = l_slots[same != 0u ? bucket[r] * kBucketSlots + (__ffs(same) - 1) : 0u];
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t row = src.base + r * BLOCK + threadIdx.x;
        const int32_t probe_tid = static_cast<int32_t>(src.base_tid + row);
        uint32_t same = masks[r] & 0xFFFFu, empty = masks[r] >> 16;
        bool hit = same != 0u && Slot::holds(first[r], key[s][r]);
        h[s][r] = hit ? static_cast<uint32_t>(Slot::tid(first[r])) + 1u : 0u;
        if (MODE == 1) local_count += hit ? 1u : 0u;
        same &= same - 1u;
        bool walking = live[r] && !(hit && (unique || MODE == 2)) && !(same == 0u && empty != 0u);
        uint32_t b = bucket[r];
        const uint32_t f = fingerprint(key[s][r]);
        while (__any(walking)) {   // (wave-uniform: the direct emission ballots)
          if (walking && same == 0u) {   // this bucket is used up and was full: the next one
            b = b + 1u == static_cast<uint32_t>(buckets) ? 0u : b + 1u;
            const u32x4 w = l_fp[b];
            const uint32_t m = bucket_masks(uint4{w.x, w.y, w.z, w.w}, f);
            same = m & 0xFFFFu;
            empty = m >> 16;
          }
          const bool have = walking && same != 0u;
          const Raw e = l_slots[have ? b * kBucketSlots + (__ffs(same) - 1) : 0u];
          if (have) same &= same - 1u;
          hit = have && Slot::holds(e, key[s][r]);
          const bool is_first = hit && h[s][r] == 0u;
          if (is_first) h[s][r] = static_cast<uint32_t>(Slot::tid(e)) + 1u;
          if (MODE == 1) local_count += hit ? 1u : 0u;
          if (MODE == 0) dense_emit_direct(hit && !is_first, probe_tid, Slot::tid(e), out_probe, out_build, capacity, out_count);
          if (hit && (unique || MODE == 2)) walking = false;
          if (same == 0u && empty != 0u) walking = false;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int r = 0; r < R; ++r) key[s][r] = next_key[s][r];
      filter_words[s] = next_filter_words[s];
    }

    if (MODE == 2) {
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int64_t tile = super * S + s;
        if (tile >= num_tiles) break;
        const Source src = source_of(tile);
        uint64_t mine = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const bool bit = ((live_mask >> (s * R + r)) & 1u) && ((h[s][r] != 0u) != (anti != 0));
          const uint64_t word = msb_first(__ballot(bit));
          if (lane == r) mine = word;
          if (lane == 0) local_count += __popcll(word);
        }
        const int64_t w = (src.base >> 6) + lane * kWaves + wave;
        if (lane < R && w < ((src.n + 63) >> 6)) store_global(mine, &src.out_bitmap[w]);
      }
      continue;
    }
    if (MODE == 1) continue;

    uint64_t m[S][R];
    int total = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        m[s][r] = __ballot(h[s][r] != 0u);
        total += __popcll(m[s][r]);
      }
    }
    if (lane == 0) s_wave_total[parity][wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
      int all = 0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) all += s_wave_total[parity][w];
      s_tile_base = all != 0 ? atomicAdd(out_count, static_cast<unsigned long long>(all)) : 0ull;
    }
    __syncthreads();
    unsigned long long base = s_tile_base;
    for (int w = 0; w < wave; ++w) base += s_wave_total[parity][w];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int64_t tile = super * S + s;
      const Source src = tile < num_tiles ? source_of(tile) : Source();
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t row = src.base + r * BLOCK + threadIdx.x;
        const unsigned long long o = base + rank_below(m[s][r]);
        if (h[s][r] != 0u && o < capacity) {
          __builtin_nontemporal_store(static_cast<int32_t>(src.base_tid + row), &out_probe[o]);
          __builtin_nontemporal_store(static_cast<int32_t>(h[s][r] - 1u), &out_build[o]);
        }
        base += __popcll(m[s][r]);
      }
    }
  }
  if (MODE == 1 || MODE == 2) {
    local_count = wave_reduce_add(local_count);
    if (lane == 0 && local_count != 0 && out_count != nullptr) atomicAdd(out_count, local_count);
  }
}

}  // namespace qsx

#endif  // QSX_CSRC_JOIN_LDS_BUCKET_HPP_
