// join_lds.hpp — K4 against a join table held in LDS (north_star: "LDS-staged open-addressing hash tables"; SURVEY.md §7
// step 4: "LDS-staged variant when the per-partition build side <= LDS budget").
//
// Reference loop: HashTable::getAllFromValueAccessorImpl (storage/HashTable.hpp:2145-2181) over
// SimpleScalarSeparateChainingHashTable::getNextEntryForKey (storage/SimpleScalarSeparateChainingHashTable.hpp:751-781).
//
// Why: a lookup of a table in HBM / L2 goes through the CU's vector memory pipeline, which takes a wave's 64 distinct lines
// one after the other — 0.42 ms per 100 M lookups of ANY table up to 3.5 MiB (tools/ubench/gather_floor.hip, DESIGN.md §4
// "What bounds a probe").  A build side of a few ten thousand keys (nation, region, a filtered dimension, one partition of a
// radix split) fits the 160 KiB of LDS of one CU: every workgroup copies the sealed table in once (a few µs: <= 144 KiB from
// L2) and then answers its probe rows with ds_read — 64 lanes over 64 banks, a handful of cycles per wave — so the kernel
// is left with its streams: 4 B of key in, 8 B of pair out per row.
//
// Two tables, both copies of the device tables of join.hip / join_dense.hpp made per workgroup:
//   lds_dense_probe_kernel   the directly addressed table (or the shadow of a hashed table over a dense key domain):
//                            head words of 4 bytes, key range <= kLdsDenseMaxWords;
//   lds_bucket_probe_kernel  the bucketed table behind its fingerprint plane (sparse keys): slots + fingerprints of up
//                            to kLdsBucketMaxSlots slots.
// Tiles are the 4096 rows of every probe kernel here (the run tables of block_runs.hpp serve them unchanged); a workgroup of
// 1024 threads works on four of them at a time.  Match compaction as dense_probe_kernel MODE 0: ballots per step, the
// waves' totals meet in LDS, ONE global atomic per unit reserves the output, every step's matches leave as one contiguous run.
#ifndef QSX_CSRC_JOIN_LDS_HPP_
#define QSX_CSRC_JOIN_LDS_HPP_

#include "join_dense.hpp"
#include "lip_view.hpp"

namespace qsx {

constexpr int kLdsTile = kDenseTile;                      // 4096 rows
constexpr int kLdsDenseMaxWords = 36 * 1024;              // 144 KiB of head words; + the waves' totals: one workgroup per CU
constexpr int kLdsStaticBytes = 1024;                     // what the kernels' static LDS may take next to the table

// MODE 0: pairs, 1: count, 2: existence bitmap (anti: the complement among the live rows).
//
// One workgroup of 1024 threads per CU; its unit of work is a SUPER TILE of kLdsSub tiles (4 x 4096 rows, 16 rows per
// thread).  The reservation of output space is one same-address atomic per unit, and those complete one at a time,
// ~12 ns each, device-wide: per 4096-row tile that is 24 K atomics = 0.29 ms per 100 M rows — invisible next to lookups
// that take 0.5 ms, the whole kernel once the lookups are ds_reads (first version: 0.33 ms whatever the table).  Per
// 16 K rows it is 0.07 ms and hides under the streams.  A sub-tile is a tile of the run tables (block_runs.hpp): each has
// its own stripe, row count, filter and base tuple id, looked up from the table when it is needed (scalar loads).
//
// kLds = false: the same kernel over a table that stays in HBM / L2 (head words through dense_head_word) — the ONE-PASS form
// of a pair-emitting probe under a filter.  With a reservation per 4096-row tile that probe had to run in two passes (count
// per unit, scan, write: the keys are read twice) because 146 K same-address atomics cost 1.7 ms per 600 M rows; one
// reservation per 16 K rows makes it 37 K = 0.4 ms, hidden under the streams, and the keys are read once.
// NLIP > 0: LIP filters tested inside the probe, between the input bitmap and the table lookup — what
// HashInnerJoinWorkOrder::execute does with its LIPFilterAdaptiveProber before it probes
// (relational_operators/HashJoinOperator.cpp:450-470, utility/lip_filter/LIPFilterAdaptiveProber.hpp:113-228): no bitmap
// pass of its own, no second read of the keys.
constexpr int kLdsBlock = 1024;
constexpr int kLdsSub = 4;
constexpr int kMaxFusedLip = 2;
struct LipViews {
  LipView f[kMaxFusedLip];
};
// (a table that stays in HBM takes units of 2 sub-tiles: 8 rows per thread fit 64 registers, so two workgroups share a CU and
// twice as many lookups are in flight — what a latency-bound probe wants; the LDS table admits one workgroup per CU anyway)
constexpr int kLdsSubGlobal = 2;
template <typename KeyT, int MODE, bool kRuns = false, bool kLds = true, int NLIP = 0>
__global__ __launch_bounds__(kLdsBlock, kLds ? 1 : 8) void lds_dense_probe_kernel(DenseTableView t, const KeyT *__restrict__ keys, int64_t n,
                                                                   int32_t probe_base_tid, const uint64_t *__restrict__ filter,
                                                                   int32_t *__restrict__ out_probe, int32_t *__restrict__ out_build,
                                                                   int64_t capacity_signed, unsigned long long *__restrict__ out_count,
                                                                   uint64_t *__restrict__ out_bitmap, int anti,
                                                                   const long long *__restrict__ runs = nullptr, LipViews lips = LipViews{}) {
  constexpr int BLOCK = kLdsBlock, S = kLds ? kLdsSub : kLdsSubGlobal;
  constexpr int R = kLdsTile / BLOCK;
  constexpr int kWaves = BLOCK / kWave;
  static_assert(R * BLOCK == kLdsTile && R <= kWave, "a tile is R steps of BLOCK rows; lane r holds the filter word of step r");
  using Source = ProbeTileSource<KeyT>;
  extern __shared__ uint32_t l_head[];
  __shared__ int s_wave_total[2][kWaves];
  __shared__ unsigned long long s_tile_base;
  const unsigned long long capacity = static_cast<unsigned long long>(capacity_signed);
  const int64_t num_tiles = kRuns ? runs[2] : (n + kLdsTile - 1) / kLdsTile;
  const int64_t num_super = (num_tiles + S - 1) / S;
  const int wave = threadIdx.x >> 6, lane = lane_id();

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
      if (tile >= num_tiles) {        // (workgroup-uniform: the run's last super tile may be short)
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
          k[s][r] = load_global_nt(&src.keys[row < src.n ? row : src.n - 1]);   // clamped, not guarded
        }
      }
      if (src.filter != nullptr && lane < R) {
        const int64_t w = (src.base >> 6) + lane * kWaves + wave;
        if (w < ((src.n + 63) >> 6)) words[s] = load_global(&src.filter[w]);
      }
    }
  };
  // the first super tile's keys are requested before the table is copied: both travel together
  if (static_cast<int64_t>(blockIdx.x) < num_super) request(blockIdx.x, key, filter_words);
  if constexpr (kLds) {
    // head[] -> LDS, 16 bytes per lane and step (a device allocation: 256-byte aligned; the tail goes word by word)
    const int range = static_cast<int>(t.range);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *src4 = reinterpret_cast<const u32x4 *>(t.head);
    for (int i = threadIdx.x; i < (range >> 2); i += BLOCK) {
      const u32x4 v = load_global(&src4[i]);
      reinterpret_cast<u32x4 *>(l_head)[i] = v;
    }
    for (int i = (range & ~3) + threadIdx.x; i < range; i += BLOCK) l_head[i] = load_global(&t.head[i]);
  }
  __syncthreads();

  unsigned long long local_count = 0;
  int parity = 0;
  for (int64_t super = blockIdx.x; super < num_super; super += gridDim.x, parity ^= 1) {
    if (super + gridDim.x < num_super) request(super + gridDim.x, next_key, next_filter_words);
    uint32_t h[S][R];
    uint32_t live_mask = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int64_t tile = super * S + s;
      const bool present = tile < num_tiles;
      const Source src = present ? source_of(tile) : Source();
      const int64_t n_rows = present ? src.n : 0;
      bool live[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t row = src.base + r * BLOCK + threadIdx.x;
        const uint64_t filter_word = __shfl(filter_words[s], r, kWave);   // before any branch: every lane takes part
        live[r] = row < n_rows && msb_bit(filter_word, lane);
      }
      if constexpr (NLIP > 0) {
        // the filter bits of the sub-tile's live rows: independent reads, issued together (dead rows read word 0)
#pragma unroll
        for (int f = 0; f < NLIP; ++f) {
          long long bit[R];
          unsigned long long w[R];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            bit[r] = live[r] ? lip_bit_index(lips.f[f], static_cast<long long>(key[s][r])) : -2;
            w[r] = load_global(&lips.f[f].words[bit[r] >= 0 ? bit[r] >> 6 : 0]);
          }
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const bool set = bit[r] >= 0 && ((w[r] >> (bit[r] & 63)) & 1ull) != 0;
            // (outside an exact filter's range: a miss — a hit for an anti filter, BitVectorExactFilter.hpp:158-172)
            live[r] = live[r] && (bit[r] == -1 ? lips.f[f].is_anti != 0 : (set != (lips.f[f].is_anti != 0)));
          }
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        live_mask |= live[r] ? (1u << (s * R + r)) : 0u;
        const uint64_t idx = dense_index(t, key[s][r]);
        const bool lookup = live[r] && idx != ~0ull;
        uint32_t word;
        if constexpr (kLds) word = l_head[lookup ? static_cast<int>(idx) : 0];   // unconditional ds_read (dead lanes read word 0)
        else word = dense_head_word(t, lookup ? idx : 0);
        h[s][r] = lookup ? word : 0u;
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
        const int64_t w = (src.base >> 6) + lane * kWaves + wave;   // lane r holds the word of step r
        if (lane < R && w < ((src.n + 63) >> 6)) store_global(mine, &src.out_bitmap[w]);
      }
      continue;
    }

    uint64_t m[S][R];
    int total = 0;
    bool any_chain = false;
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        m[s][r] = __ballot(h[s][r] != 0u);
        total += __popcll(m[s][r]);
        any_chain = any_chain || (h[s][r] & kChainBit) != 0u;
      }
    }
    const bool wave_has_chain = __any(any_chain);
    uint32_t chain_next[S][R];
    if (MODE == 1) {
      if (lane == 0) local_count += total;
      if (wave_has_chain) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
          for (int r = 0; r < R; ++r) chain_next[s][r] = (h[s][r] & kChainBit) ? t.ov[h[s][r] & ~kChainBit].y : 0u;
        }
      }
    } else {
      // s_wave_total is double-buffered by parity, s_tile_base is rewritten only behind the next unit's first barrier
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
          uint32_t tid = h[s][r] - 1u;
          chain_next[s][r] = 0u;
          if (wave_has_chain && (h[s][r] & kChainBit)) {
            const uint2 e = t.ov[h[s][r] & ~kChainBit];
            tid = e.x;
            chain_next[s][r] = e.y;
          }
          const unsigned long long o = base + rank_below(m[s][r]);
          if (h[s][r] != 0u && o < capacity) {
            __builtin_nontemporal_store(static_cast<int32_t>(src.base_tid + row), &out_probe[o]);
            __builtin_nontemporal_store(static_cast<int32_t>(tid), &out_build[o]);
          }
          base += __popcll(m[s][r]);
        }
      }
    }
    // duplicate build keys: the chains live in the overflow list in HBM (rare: a wave-uniform loop)
    if (wave_has_chain) {
      // (unrolled: chain_next[][] indexed by a loop variable would live in scratch, and its stores sit in the hot loop)
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int64_t tile = super * S + s;
        if (tile >= num_tiles) break;
        const Source src = source_of(tile);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int64_t row = src.base + r * BLOCK + threadIdx.x;
          uint32_t cur_word = chain_next[s][r];
          while (__any(cur_word != 0u)) {
            uint32_t tid = cur_word - 1u, nxt = 0u;
            if (cur_word & kChainBit) {
              const uint2 e = t.ov[cur_word & ~kChainBit];
              tid = e.x;
              nxt = e.y;
            }
            if (MODE == 1) {
              local_count += cur_word != 0u ? 1u : 0u;
            } else {
              dense_emit_direct(cur_word != 0u, static_cast<int32_t>(src.base_tid + row), static_cast<int32_t>(tid), out_probe, out_build,
                                capacity, out_count);
            }
            cur_word = nxt;
          }
        }
      }
    }
  }
  if (MODE == 1 || MODE == 2) {
    local_count = wave_reduce_add(local_count);
    if (lane == 0 && local_count != 0 && out_count != nullptr) atomicAdd(out_count, local_count);
  }
}

}  // namespace qsx

#endif  // QSX_CSRC_JOIN_LDS_HPP_
