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
//   [2] tiles in the run   [3] 1 when the coding arrays below are present, else 0
//   then arrays of one word per block:  first_tile[nb + 1] | rows[nb] | in[nb] | filter[nb] | out[nb] | base[nb]
//   and, for a run of JOIN KEY stripes some of which are compressed (header word 3):  code_width[nb] | dictionary[nb]
// (in / filter / out are device addresses, 0 = absent; base is a row number added to emitted tuple ids).
//
// Coded key stripes (storage/CompressedColumnStoreTupleStorageSubBlock.cpp, CompressedTupleStorageSubBlock.hpp:225-300
// getAttributeValue): the reference compresses every block on its own — an INT / LONG attribute of one block may lie as
// values, as 1 / 2 / 4-byte truncated values, or as 1 / 2 / 4-byte codes into the block's own sorted dictionary.  A block of
// the run with code_width != 0 presents that stripe as it lies; the kernels read the code and widen it (truncation) or look
// it up in the block's dictionary (a few KB, read through L2), per tile — a tile never straddles two blocks.
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
__device__ __forceinline__ bool run_has_coding(const long long *__restrict__ t) { return t[3] != 0; }
__device__ __forceinline__ int run_code_width(const long long *__restrict__ t, int b) {
  return static_cast<int>(t[kRunHeaderWords + (t[0] + 1) + 5 * t[0] + b]);
}
__device__ __forceinline__ const void *run_dictionary(const long long *__restrict__ t, int b) {
  return as_global(reinterpret_cast<const unsigned char *>(t[kRunHeaderWords + (t[0] + 1) + 6 * t[0] + b]));
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
  int code_width;            // 0: `keys` are values; 1 / 2 / 4: `keys` is a stripe of codes that wide (coded_key below)
  const void *dictionary;    // the block's dictionary of KeyT values, or nullptr: the codes are truncated values
};
// Row `row` of a coded stripe (src.code_width != 0) as the key it stands for.
template <typename KeyT>
__device__ __forceinline__ KeyT coded_key(const ProbeTileSource<KeyT> &src, int64_t row) {
  const void *codes = src.keys;
  uint32_t code;
  if (src.code_width == 1) code = load_global_nt(&static_cast<const uint8_t *>(codes)[row]);
  else if (src.code_width == 2) code = load_global_nt(&static_cast<const uint16_t *>(codes)[row]);
  else code = load_global_nt(&static_cast<const uint32_t *>(codes)[row]);
  if (src.dictionary == nullptr) return static_cast<KeyT>(code);   // (truncation keeps non-negative values only: zero-extension)
  return load_global(&static_cast<const KeyT *>(src.dictionary)[code]);
}
// R rows of a coded stripe at once — row_of(r) = the (clamped) row of step r: the width is chosen once, outside the reads (a
// choice per read puts every read into a basic block of its own and the tile pays R memory round trips instead of one), and a
// dictionary's lookups follow the codes as a second batch.
template <typename KeyT, int R, typename RowOf>
__device__ __forceinline__ void coded_keys(const ProbeTileSource<KeyT> &src, KeyT (&k)[R], RowOf row_of) {
  const void *codes = src.keys;
  uint32_t code[R];
  if (src.code_width == 1) {
#pragma unroll
    for (int r = 0; r < R; ++r) code[r] = load_global_nt(&static_cast<const uint8_t *>(codes)[row_of(r)]);
  } else if (src.code_width == 2) {
#pragma unroll
    for (int r = 0; r < R; ++r) code[r] = load_global_nt(&static_cast<const uint16_t *>(codes)[row_of(r)]);
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) code[r] = load_global_nt(&static_cast<const uint32_t *>(codes)[row_of(r)]);
  }
  if (src.dictionary == nullptr) {
#pragma unroll
    for (int r = 0; r < R; ++r) k[r] = static_cast<KeyT>(code[r]);
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) k[r] = load_global(&static_cast<const KeyT *>(src.dictionary)[code[r]]);
  }
}
template <typename KeyT, int kTileRows, bool kRuns>
__device__ __forceinline__ ProbeTileSource<KeyT> probe_tile_source(const long long *__restrict__ runs, int64_t tile,
                                                                   const KeyT *keys, int64_t n, int32_t base_tid,
                                                                   const uint64_t *filter, uint64_t *out_bitmap) {
  if (!kRuns) return ProbeTileSource<KeyT>{keys, n, tile * kTileRows, base_tid, filter, out_bitmap, 0, 0, nullptr};
  const RunTile at = run_locate(runs, static_cast<int>(tile));
  const bool coded = run_has_coding(runs);
  return ProbeTileSource<KeyT>{run_in<KeyT>(runs, at.block), run_rows(runs, at.block),
                               static_cast<int64_t>(at.tile_in_block) * kTileRows, static_cast<int32_t>(run_base(runs, at.block)),
                               run_filter(runs, at.block), run_out<uint64_t>(runs, at.block), at.block,
                               coded ? run_code_width(runs, at.block) : 0, coded ? run_dictionary(runs, at.block) : nullptr};
}

}  // namespace qsx

#ifndef __HIPCC_RTC__
#include <vector>

namespace qsx {

// Host side: the table of a run whose tiles hold tile_rows rows.  Returns the number of tiles (0: nothing to do);
// -1 when the run is too long for 32-bit tile numbers.
// code_widths != nullptr: a run of key stripes with their coding (code_widths[b] = 0 / 1 / 2 / 4; dictionaries may be nullptr, and
// so may its entries: truncated values).
inline long long build_run_table(long long tile_rows, long long num_blocks, const int64_t *rows, const void *const *in,
                                 const void *const *filters, void *const *out, const int64_t *base,
                                 std::vector<long long> *table, const int32_t *code_widths = nullptr,
                                 const void *const *dictionaries = nullptr) {
  const size_t nb = static_cast<size_t>(num_blocks);
  table->assign(kRunHeaderWords + (nb + 1) + (code_widths != nullptr ? 7 : 5) * nb, 0);
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
    if (code_widths != nullptr) {
      t_base[nb + b] = code_widths[b];
      t_base[2 * nb + b] = dictionaries != nullptr ? static_cast<long long>(reinterpret_cast<uintptr_t>(dictionaries[b])) : 0;
    }
  }
  first[nb] = tiles;
  if (tiles > 0x7FFFFFF0ll) return -1;
  (*table)[0] = num_blocks;
  (*table)[1] = (same && uniform > 0) ? uniform : 0;
  (*table)[2] = tiles;
  (*table)[3] = code_widths != nullptr ? 1 : 0;
  return tiles;
}

}  // namespace qsx
#endif  // __HIPCC_RTC__

#endif  // QSX_CSRC_BLOCK_RUNS_HPP_
