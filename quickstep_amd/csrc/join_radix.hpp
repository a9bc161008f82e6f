// join_radix.hpp — radix-partitioned probe with LDS-resident hash tables.
//
// Why: probing a table that does not fit the 4 MiB per-XCD L2 costs one 64-byte
// fabric fetch per probe row (measured: 100 M probes of a 16 MiB table move
// 6-12 GB for 0.4 GB of keys, profiles/r01_pmc_summary_agg_v2.txt), i.e. the table
// lines, not the key stream, are the traffic.  Partitioning the probe rows on the
// top bits of the table's own hash makes every partition's share of the table a
// contiguous slot range small enough for LDS, so the probe touches HBM only for
// coalesced streams: keys in, (key, tid) partitions out and back in, table ranges in,
// pairs out.  The reference's partitioned joins do the same at relation granularity
// (per-partition hash tables, BuildHashOperator.cpp:82-91, HashJoinOperator.cpp:220-231).
//
//   build side: nothing extra.  Partition p of P owns the home slots
//     [p * cap / P, (p + 1) * cap / P) of the open-addressing table; linear probing can
//     displace an entry past that range by at most `max_disp` slots, which the build
//     kernels track (atomicMax), so the range plus that margin holds every entry of p.
//   probe side (per qsx_join_probe call)
//     R1 radix_probe_hist      per-workgroup histogram of its contiguous row chunk
//     R2 launch_scan           start of (partition p, workgroup b), partition-major
//     R3 radix_probe_scatter   tile-local counting sort in LDS, then runs of one partition
//                              go out contiguously: out[start(p, b) + ...] = (key, tid)
//     R4 radix_join            one workgroup per (partition, slice): copy the partition's
//                              table range into an LDS table (ds_cmpst_b64), stream the probe
//                              slice through it, stage pairs in LDS, one global atomic per tile.
// A partition with more entries than the LDS table accepts (duplicate-heavy keys) makes R4
// probe the global table for that partition instead (same result, slower).
#ifndef QSX_CSRC_JOIN_RADIX_HPP_
#define QSX_CSRC_JOIN_RADIX_HPP_

#include "common.hpp"

namespace qsx {

constexpr int kRBlock = 256;
constexpr int kRadixLdsSlots = 4096;        // 32 KiB LDS table per workgroup
constexpr int kRadixMaxBuild = 3072;        // entries an LDS table accepts (load <= 0.75)
constexpr int kRadixTile = 4096;            // rows per scatter tile (16 per thread)
constexpr int kRadixJoinTile = 2048;        // rows per join tile (8 per thread)
constexpr int kRadixMaxPartitions = 1024;     // scatter LDS: (2 * 4096 + 4 * P) * 4 B <= 48 KiB

// Home slot of a key in the open-addressing table (join.hip: slot_of(key) & ~1) and the
// partition that owns it: the top log_p bits of the home slot.
struct RadixGeom {
  int table_shift;   // 64 - log2(capacity)
  int part_shift;    // log2(capacity) - log_p
  int log_p;
};
__device__ __forceinline__ uint64_t radix_home(int32_t key, const RadixGeom &g) {
  return ((static_cast<uint64_t>(static_cast<uint32_t>(key)) * 0x9E3779B97F4A7C15ull) >> g.table_shift) & ~1ull;
}
__device__ __forceinline__ int radix_partition(int32_t key, const RadixGeom &g) {
  return static_cast<int>(radix_home(key, g) >> g.part_shift);
}
__device__ __forceinline__ int radix_slot(int32_t key) {
  // independent of the partition bits: a second multiplicative hash
  return static_cast<int>((static_cast<uint32_t>(key) * 0x85EBCA6Bu) >> 20) & (kRadixLdsSlots - 1);
}

__device__ __forceinline__ bool radix_row_selected(const uint64_t *filter, int64_t row) {
  return filter == nullptr || ((filter[row >> 6] >> (63 - (row & 63))) & 1u);
}

// ---- R1: probe-side histogram, one contiguous row chunk per workgroup -------------------------
__global__ __launch_bounds__(kRBlock) void radix_probe_hist(const int32_t *__restrict__ keys, int64_t n,
                                                           const uint64_t *__restrict__ filter, RadixGeom geom,
                                                           int64_t rows_per_block, int32_t *__restrict__ hist_t) {
  extern __shared__ int32_t s_hist[];
  const int P = 1 << geom.log_p;
  for (int i = threadIdx.x; i < P; i += kRBlock) s_hist[i] = 0;
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * rows_per_block;
  const int64_t end = begin + rows_per_block < n ? begin + rows_per_block : n;
  constexpr int U = 8;  // independent key loads in flight per thread
  for (int64_t base = begin; base < end; base += kRBlock * U) {
    int part[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * kRBlock + threadIdx.x;
      part[u] = (i < end && radix_row_selected(filter, i)) ? radix_partition(keys[i], geom) : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (part[u] >= 0) atomicAdd(&s_hist[part[u]], 1);
    }
  }
  __syncthreads();
  const int64_t G = gridDim.x;
  for (int i = threadIdx.x; i < P; i += kRBlock) hist_t[static_cast<int64_t>(i) * G + blockIdx.x] = s_hist[i];
}

// ---- R3: probe-side scatter ----------------------------------------------------------------------
// Dynamic LDS: stage_key[T] | stage_tid[T] | tile_cnt[P] | tile_start[P] | cursor[P] (int64 -> 2P ints)
__global__ __launch_bounds__(kRBlock) void radix_probe_scatter(const int32_t *__restrict__ keys, int64_t n,
                                                              const uint64_t *__restrict__ filter,
                                                              int32_t probe_base_tid, RadixGeom geom, int64_t rows_per_block,
                                                              const int64_t *__restrict__ starts,
                                                              int32_t *__restrict__ out_keys,
                                                              int32_t *__restrict__ out_tids) {
  extern __shared__ int32_t s_mem[];
  const int P = 1 << geom.log_p;
  int32_t *stage_key = s_mem;
  int32_t *stage_tid = stage_key + kRadixTile;
  int32_t *tile_cnt = stage_tid + kRadixTile;
  int32_t *tile_start = tile_cnt + P;
  int64_t *cursor = reinterpret_cast<int64_t *>(tile_start + P + (P & 1));
  __shared__ int32_t s_wave_total[kRBlock / kWave];
  const int64_t G = gridDim.x;
  for (int i = threadIdx.x; i < P; i += kRBlock) {
    tile_cnt[i] = 0;
    cursor[i] = starts[static_cast<int64_t>(i) * G + blockIdx.x];
  }
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * rows_per_block;
  const int64_t end = begin + rows_per_block < n ? begin + rows_per_block : n;
  constexpr int R = kRadixTile / kRBlock;
  for (int64_t tile = begin; tile < end; tile += kRadixTile) {
    int32_t key[R];
    int part[R], rank[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int64_t row = tile + j * kRBlock + threadIdx.x;
      part[j] = -1;
      if (row < end && radix_row_selected(filter, row)) {
        key[j] = keys[row];
        part[j] = radix_partition(key[j], geom);
      }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      if (part[j] >= 0) rank[j] = atomicAdd(&tile_cnt[part[j]], 1);
    }
    __syncthreads();
    // exclusive scan of tile_cnt over P partitions (P / 256 consecutive counters per thread)
    {
      const int per = (P + kRBlock - 1) / kRBlock;
      const int first = threadIdx.x * per;
      int local = 0;
      for (int k = 0; k < per; ++k) {
        if (first + k < P) local += tile_cnt[first + k];
      }
      int incl = local;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const int up = __shfl_up(incl, off, kWave);
        if (lane_id() >= off) incl += up;
      }
      if (lane_id() == kWave - 1) s_wave_total[threadIdx.x >> 6] = incl;
      __syncthreads();
      int base = incl - local;
      for (int w = 0; w < (threadIdx.x >> 6); ++w) base += s_wave_total[w];
      for (int k = 0; k < per; ++k) {
        if (first + k < P) {
          tile_start[first + k] = base;
          base += tile_cnt[first + k];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < R; ++j) {
      if (part[j] >= 0) {
        const int pos = tile_start[part[j]] + rank[j];
        stage_key[pos] = key[j];
        stage_tid[pos] = static_cast<int32_t>(probe_base_tid + tile + j * kRBlock + threadIdx.x);
      }
    }
    __syncthreads();
    const int valid = tile_start[P - 1] + tile_cnt[P - 1];
    for (int i = threadIdx.x; i < valid; i += kRBlock) {
      const int32_t k = stage_key[i];
      const int p = radix_partition(k, geom);
      const int64_t g = cursor[p] + (i - tile_start[p]);
      out_keys[g] = k;
      out_tids[g] = stage_tid[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += kRBlock) {
      cursor[i] += tile_cnt[i];
      tile_cnt[i] = 0;
    }
    __syncthreads();
  }
}

// ---- R4: join of one (partition, slice) ------------------------------------------------------------
struct RadixJoinArgs {
  const unsigned long long *table_slots;  // the open-addressing table: {tid:32 | key:32} entries, ~0 = empty
  uint64_t table_mask;                    // capacity - 1
  const unsigned int *max_disp;           // largest displacement any build saw (device word)
  RadixGeom geom;
  const int32_t *probe_keys;              // partitioned probe keys / tids
  const int32_t *probe_tids;
  const int64_t *probe_starts;            // start of (p, b): [p * G + b]; [P * G] = total
  int64_t probe_blocks;                   // G
  int slices;
};

template <int MODE>  // 0: emit pairs, 1: count only
__global__ __launch_bounds__(kRBlock) void radix_join(RadixJoinArgs a, int32_t *__restrict__ out_probe,
                                                     int32_t *__restrict__ out_build, int64_t capacity,
                                                     unsigned long long *__restrict__ out_count) {
  __shared__ unsigned long long s_table[kRadixLdsSlots];
  __shared__ int32_t s_probe[MODE == 0 ? kRadixJoinTile : 1];
  __shared__ int32_t s_build[MODE == 0 ? kRadixJoinTile : 1];
  __shared__ int s_fill;
  __shared__ int s_nbuild;
  __shared__ unsigned long long s_base;
  const int p = blockIdx.x / a.slices;
  const int slice = blockIdx.x % a.slices;
  const int64_t part_lo = a.probe_starts[static_cast<int64_t>(p) * a.probe_blocks];
  const int64_t part_hi = a.probe_starts[static_cast<int64_t>(p + 1) * a.probe_blocks];
  const int64_t len = part_hi - part_lo;
  const int64_t per = (len + a.slices - 1) / a.slices;
  const int64_t lo = part_lo + slice * per;
  const int64_t hi = lo + per < part_hi ? lo + per : part_hi;
  if (lo >= hi) return;  // workgroup-uniform

  // ---- the partition's share of the table -> LDS ----------------------------------------------
  for (int i = threadIdx.x; i < kRadixLdsSlots; i += kRBlock) s_table[i] = ~0ull;
  if (threadIdx.x == 0) s_nbuild = 0;
  __syncthreads();
  {
    const uint64_t width = 1ull << a.geom.part_shift;
    const uint64_t first = static_cast<uint64_t>(p) << a.geom.part_shift;
    const uint64_t span = width + *a.max_disp + 2;  // entries displaced past the range by linear probing
    for (uint64_t i = threadIdx.x; i < span; i += kRBlock) {
      const unsigned long long e = a.table_slots[(first + i) & a.table_mask];
      if (e == ~0ull) continue;
      const int32_t key = static_cast<int32_t>(static_cast<uint32_t>(e));
      if (radix_partition(key, a.geom) != p) continue;  // a neighbour's displaced entry
      if (atomicAdd(&s_nbuild, 1) >= kRadixMaxBuild) continue;  // too many for LDS: global fallback below
      int s = radix_slot(key);
      while (atomicCAS(&s_table[s], ~0ull, e) != ~0ull) s = (s + 1) & (kRadixLdsSlots - 1);
    }
  }
  __syncthreads();
  const bool use_lds = s_nbuild <= kRadixMaxBuild;
  const ulonglong2 *global_units = reinterpret_cast<const ulonglong2 *>(a.table_slots);
  const uint64_t global_unit_mask = a.table_mask >> 1;

  unsigned long long local_count = 0;
  constexpr int R = kRadixJoinTile / kRBlock;
  for (int64_t tile = lo; tile < hi; tile += kRadixJoinTile) {
    if (MODE == 0 && threadIdx.x == 0) s_fill = 0;
    __syncthreads();  // previous tile flushed
    int32_t key[R], ptid[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {  // all loads of the tile in flight before the first use
      const int64_t i = tile + j * kRBlock + threadIdx.x;
      key[j] = i < hi ? a.probe_keys[i] : 0;
      ptid[j] = i < hi ? a.probe_tids[i] : 0;
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int64_t i = tile + j * kRBlock + threadIdx.x;
      bool walking = i < hi;
      int s = radix_slot(key[j]);
      uint64_t gu = radix_home(key[j], a.geom) >> 1;
      while (__any(walking)) {
        bool m0 = false, m1 = false;
        int32_t t0 = 0, t1 = 0;
        if (walking) {
          if (use_lds) {
            const unsigned long long e = s_table[s];
            if (e == ~0ull) {
              walking = false;
            } else {
              m0 = static_cast<uint32_t>(e) == static_cast<uint32_t>(key[j]);
              t0 = static_cast<int32_t>(e >> 32);
              s = (s + 1) & (kRadixLdsSlots - 1);
            }
          } else {
            // same walk as probe_kernel<IntUnits> over the global table (join.hip)
            const ulonglong2 u = global_units[gu];
            const bool e0 = u.x == ~0ull, e1 = u.y == ~0ull;
            m0 = !e0 && static_cast<uint32_t>(u.x) == static_cast<uint32_t>(key[j]);
            m1 = !e0 && !e1 && static_cast<uint32_t>(u.y) == static_cast<uint32_t>(key[j]);
            t0 = static_cast<int32_t>(u.x >> 32);
            t1 = static_cast<int32_t>(u.y >> 32);
            if (e0 || e1) walking = false;
            gu = (gu + 1) & global_unit_mask;
          }
        }
        if (MODE == 1) {
          local_count += (m0 ? 1u : 0u) + (m1 ? 1u : 0u);
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const bool match = h == 0 ? m0 : m1;
            const int32_t bt = h == 0 ? t0 : t1;
            const uint64_t m = __ballot(match);
            if (m == 0) continue;  // wave-uniform
            const int leader = __ffsll(static_cast<long long>(m)) - 1;
            int base = 0;
            if (lane_id() == leader) base = atomicAdd(&s_fill, __popcll(m));
            base = __shfl(base, leader, kWave);
            const int pos = base + rank_below(m);
            const bool over = match && pos >= kRadixJoinTile;
            if (match && !over) {
              s_probe[pos] = ptid[j];
              s_build[pos] = bt;
            }
            const uint64_t mo = __ballot(over);
            if (mo != 0) {  // more matches than the stage holds (duplicate-heavy keys): straight to HBM
              const int leader2 = __ffsll(static_cast<long long>(mo)) - 1;
              unsigned long long gbase = 0;
              if (lane_id() == leader2) gbase = atomicAdd(out_count, static_cast<unsigned long long>(__popcll(mo)));
              gbase = __shfl(gbase, leader2, kWave);
              if (over) {
                const unsigned long long o = gbase + rank_below(mo);
                if (o < static_cast<unsigned long long>(capacity)) {
                  out_probe[o] = ptid[j];
                  out_build[o] = bt;
                }
              }
            }
          }
        }
      }
    }
    if (MODE == 0) {
      __syncthreads();
      const int produced = s_fill;
      const int staged = produced < kRadixJoinTile ? produced : kRadixJoinTile;
      if (threadIdx.x == 0) s_base = atomicAdd(out_count, static_cast<unsigned long long>(staged));
      __syncthreads();
      const unsigned long long base = s_base;
      for (int i = threadIdx.x; i < staged; i += kRBlock) {
        const unsigned long long o = base + i;
        if (o < static_cast<unsigned long long>(capacity)) {
          out_probe[o] = s_probe[i];
          out_build[o] = s_build[i];
        }
      }
    }
  }
  if (MODE == 1) {
    local_count = wave_reduce_add(local_count);
    if (lane_id() == 0 && local_count != 0) atomicAdd(out_count, local_count);
  }
}

}  // namespace qsx

#endif  // QSX_CSRC_JOIN_RADIX_HPP_
