// block_runs.hpp — a run of storage blocks presented to ONE launch of a tile-walking kernel (K1 select, K4 probe).
//
// The reference hands an operator one work order per block (RelationalOperator::getAllWorkOrders,
// relational_operators/RelationalOperator.hpp:117-119; SelectOperator.cpp:83-150, HashJoinOperator.cpp:203-260); at its
// 2-4 MB blocks a launch per block is launch-bound by an order of magnitude on this device (DESIGN.md "Work-order
// granularity").  A run keeps every block's own stripes / bitmaps where they are and gives the kernel a table to find
// them: the kernel walks the tiles of the whole run, a tile never straddles two blocks.
//
// Table (64-bit words, device memory, read through the kernel's `const __restrict__` argument with workgroup- or
// wave-uniform indices, i.e. scalar loads):
//   [0] number of blocks   [1] tiles per block when every block but the last has the same number of tiles, else 0
//   [2] tiles in the run   [3] unused
//   then arrays of one word per block:  first_tile[nb + 1] | rows[nb] | in[nb] | filter[nb] | out[nb] | base[nb]
// (in / filter / out are device addresses, 0 = absent; base is a row number added to emitted tuple ids).
#ifndef QSX_CSRC_BLOCK_RUNS_HPP_
#define QSX_CSRC_BLOCK_RUNS_HPP_

#include "device_common.hpp"

namespace qsx {

constexpr int kRunHeaderWords = 4;

struct RunTile {
  int block;
  int tile_in_block;
};

// `tile` must be uniform over the lanes that call this (make it so with readfirstlane where the compiler cannot see it).
__device__ __forceinline__ RunTile run_locate(const long long *__restrict__ table, int tile) {
  const int nb = static_cast<int>(table[0]);
  const int uniform = static_cast<int>(table[1]);
  int b;
  if (uniform > 0) {
    b = static_cast<int>(static_cast<unsigned>(tile) / static_cast<unsigned>(uniform));
    return RunTile{b, tile - b * uniform};
  }
  int lo = 0, hi = nb - 1;   // last block whose first tile <= tile
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[kRunHeaderWords + mid] <= tile) lo = mid; else hi = mid - 1;
  }
  return RunTile{lo, tile - static_cast<int>(table[kRunHeaderWords + lo])};
}
__device__ __forceinline__ long long run_rows(const long long *__restrict__ t, int b) {
  return t[kRunHeaderWords + (t[0] + 1) + b];
}
template <typename T>
__device__ __forceinline__ const T *run_in(const long long *__restrict__ t, int b) {
  return as_global(reinterpret_cast<const T *>(t[kRunHeaderWords + (t[0] + 1) + t[0] + b]));
}
__device__ __forceinline__ const uint64_t *run_filter(const long long *__restrict__ t, int b) {
  return as_global(reinterpret_cast<const uint64_t *>(t[kRunHeaderWords + (t[0] + 1) + 2 * t[0] + b]));
}
template <typename T>
__device__ __forceinline__ T *run_out(const long long *__restrict__ t, int b) {
  return as_global(reinterpret_cast<T *>(t[kRunHeaderWords + (t[0] + 1) + 3 * t[0] + b]));
}
__device__ __forceinline__ long long run_base(const long long *__restrict__ t, int b) {
  return t[kRunHeaderWords + (t[0] + 1) + 4 * t[0] + b];
}

// ---- a tile of a stripe, or of a run of blocks (block_runs.hpp) -----------------------------------------------------------
template <typename KeyT>
struct ProbeTileSource {
  const KeyT *keys;
  int64_t n;                 // rows of the block
  int64_t base;              // first row of the tile within the block
  int32_t base_tid;
  const uint64_t *filter;
  uint64_t *out_bitmap;
  int block;                 // position of the block in the run (0 without a run)
};
template <typename KeyT, int kTileRows, bool kRuns>
__device__ __forceinline__ ProbeTileSource<KeyT> probe_tile_source(const long long *__restrict__ runs, int64_t tile,
                                                                   const KeyT *keys, int64_t n, int32_t base_tid,
                                                                   const uint64_t *filter, uint64_t *out_bitmap) {
  if (!kRuns) return ProbeTileSource<KeyT>{keys, n, tile * kTileRows, base_tid, filter, out_bitmap, 0};
  const RunTile at = run_locate(runs, static_cast<int>(tile));
  return ProbeTileSource<KeyT>{run_in<KeyT>(runs, at.block), run_rows(runs, at.block),
                               static_cast<int64_t>(at.tile_in_block) * kTileRows, static_cast<int32_t>(run_base(runs, at.block)),
                               run_filter(runs, at.block), run_out<uint64_t>(runs, at.block), at.block};
}

}  // namespace qsx

#ifndef __HIPCC_RTC__
#include <vector>

namespace qsx {

// Host side: the table of a run whose tiles hold tile_rows rows.  Returns the number of tiles (0: nothing to do);
// -1 when the run is too long for 32-bit tile numbers.
inline long long build_run_table(long long tile_rows, long long num_blocks, const int64_t *rows, const void *const *in,
                                 const void *const *filters, void *const *out, const int64_t *base,
                                 std::vector<long long> *table) {
  const size_t nb = static_cast<size_t>(num_blocks);
  table->assign(kRunHeaderWords + (nb + 1) + 5 * nb, 0);
  long long *first = table->data() + kRunHeaderWords;
  long long *t_rows = first + nb + 1, *t_in = t_rows + nb, *t_filter = t_in + nb, *t_out = t_filter + nb, *t_base = t_out + nb;
  long long tiles = 0, uniform = -1;
  bool same = true;
  for (size_t b = 0; b < nb; ++b) {
    const long long bt = (rows[b] + tile_rows - 1) / tile_rows;
    first[b] = tiles;
    tiles += bt;
    if (b + 1 < nb) {                 // every block but the last
      if (uniform < 0) uniform = bt;
      else if (bt != uniform) same = false;
    } else if (uniform >= 0 && bt > uniform) {
      same = false;                   // a longer last block would spill into a block that does not exist
    } else if (uniform < 0) {
      uniform = bt;                   // a single block
    }
    t_rows[b] = rows[b];
    t_in[b] = static_cast<long long>(reinterpret_cast<uintptr_t>(in[b]));
    t_filter[b] = filters != nullptr ? static_cast<long long>(reinterpret_cast<uintptr_t>(filters[b])) : 0;
    t_out[b] = out != nullptr ? static_cast<long long>(reinterpret_cast<uintptr_t>(out[b])) : 0;
    t_base[b] = base != nullptr ? base[b] : 0;
  }
  first[nb] = tiles;
  if (tiles > 0x7FFFFFF0ll) return -1;
  (*table)[0] = num_blocks;
  (*table)[1] = (same && uniform > 0) ? uniform : 0;
  (*table)[2] = tiles;
  return tiles;
}

}  // namespace qsx
#endif  // __HIPCC_RTC__

#endif  // QSX_CSRC_BLOCK_RUNS_HPP_
