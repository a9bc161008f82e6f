// select.hip — K1 (predicate -> TupleIdSequence bitmap), bitmap algebra,
// K2 (order-preserving compaction / projection) and K5 (gather by tuple id).
//
// Reference loops replaced (paths in the Quickstep tree):
//   K1  types/operations/comparisons/LiteralComparators-inl.hpp:317-388
//   K2  storage/StorageBlock.cpp:363-399 ->
//       storage/BasicColumnStoreTupleStorageSubBlock.cpp:339-425
//   K5  expressions/scalar/ScalarAttribute.cpp:185-225
//
// Layout: a column is a dense stripe of n values in HBM (what
// BasicColumnStoreTupleStorageSubBlock keeps per attribute).  A wave owns 64
// consecutive rows per bitmap word: lane l loads row 64*w + l (coalesced
// 256/512-byte wave loads), the comparison result of the whole word is one
// v_cmp -> SGPR-pair ballot, and s_brev_b64 turns the LSB-first ballot into the
// MSB-first word BitVector uses.  No shuffles, no LDS for K1.

#include <type_traits>
#include <vector>

#include "common.hpp"
#include "block_runs.hpp"
#include "scan.hpp"

namespace qsx {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;

// ---------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------
template <typename T, int OP>
__device__ __forceinline__ bool cmp_static(T a, T b) {
  if (OP == QSX_EQ) return a == b;
  if (OP == QSX_NE) return a != b;
  if (OP == QSX_LT) return a < b;
  if (OP == QSX_LE) return a <= b;
  if (OP == QSX_GT) return a > b;
  return a >= b;
}

// R = bitmap words (64-row groups) a wave keeps in flight per iteration.
// COLS: the right operand is a second column (attribute OP attribute,
// ComparisonPredicate.cpp:214-334 -> compareColumnVectors / compareValueAccessors) instead of the literal.
template <typename T, int OP, int R, bool COLS>
__global__ __launch_bounds__(kBlock) void select_cmp_kernel(
    const T *__restrict__ col, const T *__restrict__ rhs, int64_t n, T lit, const uint64_t *__restrict__ filter,
    uint64_t *__restrict__ out, unsigned long long *__restrict__ out_count) {
  const int lane = lane_id();
  const int64_t num_words = (n + 63) >> 6;
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t num_waves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  unsigned long long count = 0;

  for (int64_t w0 = wave * R; w0 < num_words; w0 += num_waves * R) {
    T v[R], u[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = ((w0 + r) << 6) + lane;
      v[r] = load_global_nt(&col[row < n ? row : n - 1]);   // clamped, not guarded (rows past n are masked out of the ballot)
      u[r] = COLS ? load_global_nt(&rhs[row < n ? row : n - 1]) : lit;
    }
    uint64_t mine = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = ((w0 + r) << 6) + lane;
      bool pred = row < n && cmp_static<T, OP>(v[r], u[r]);
      if (filter != nullptr && w0 + r < num_words) {
        // short-circuit semantics (:344-356): only rows already selected can match.
        const uint64_t fw = filter[w0 + r];  // wave-uniform address: one scalar load
        pred = pred && msb_bit(fw, lane);
      }
      const uint64_t word = msb_first(__ballot(pred));
      count += __popcll(word);
      if (lane == r) mine = word;
    }
    if (lane < R && w0 + lane < num_words) out[w0 + lane] = mine;
  }

  if (out_count != nullptr) {
    __shared__ unsigned long long block_count;
    if (threadIdx.x == 0) block_count = 0;
    __syncthreads();
    if (lane == 0 && count != 0) atomicAdd(&block_count, count);
    __syncthreads();
    if (threadIdx.x == 0 && block_count != 0) atomicAdd(out_count, block_count);
  }
}

template <typename T, typename Pred>
static int launch_select_packed(const void *col, int64_t n, Pred pred, const uint64_t *filter, uint64_t *out, int64_t *out_count,
                                hipStream_t stream);
static bool aligned16(const void *p);
template <typename T, int OP>
struct LiteralPred;

// rhs_col == nullptr: compare with *literal; else with the second column.
template <typename T, int OP>
static int launch_select_cmp(const void *col, const void *rhs_col, int64_t n, const void *literal, const uint64_t *filter,
                             uint64_t *out, int64_t *out_count, hipStream_t stream) {
  constexpr int R = sizeof(T) == 4 ? 8 : 4;
  T lit = T();
  if (literal != nullptr) std::memcpy(&lit, literal, sizeof(T));
  const int64_t num_words = (n + 63) >> 6;
  if (rhs_col == nullptr && aligned16(col)) {
    // 16 bytes per lane per load (select_packed_kernel below); unaligned slices keep the row-per-lane kernel
    return launch_select_packed<T>(col, n, LiteralPred<T, OP>{lit}, filter, out, out_count, stream);
  }
  if (rhs_col == nullptr) {
    const int grid = grid_for(num_words, kWavesPerBlock * R);
    hipLaunchKernelGGL((select_cmp_kernel<T, OP, R, false>), dim3(grid), dim3(kBlock), 0, stream,
                       static_cast<const T *>(col), static_cast<const T *>(nullptr), n, lit, filter, out,
                       reinterpret_cast<unsigned long long *>(out_count));
  } else {
    constexpr int R2 = R / 2;  // two loads per row
    const int grid = grid_for(num_words, kWavesPerBlock * R2);
    hipLaunchKernelGGL((select_cmp_kernel<T, OP, R2, true>), dim3(grid), dim3(kBlock), 0, stream,
                       static_cast<const T *>(col), static_cast<const T *>(rhs_col), n, lit, filter, out,
                       reinterpret_cast<unsigned long long *>(out_count));
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

template <typename T>
static int dispatch_select_op(int op, const void *col, const void *rhs_col, int64_t n, const void *literal,
                              const uint64_t *filter, uint64_t *out, int64_t *out_count,
                              hipStream_t stream) {
  switch (op) {
    case QSX_EQ: return launch_select_cmp<T, QSX_EQ>(col, rhs_col, n, literal, filter, out, out_count, stream);
    case QSX_NE: return launch_select_cmp<T, QSX_NE>(col, rhs_col, n, literal, filter, out, out_count, stream);
    case QSX_LT: return launch_select_cmp<T, QSX_LT>(col, rhs_col, n, literal, filter, out, out_count, stream);
    case QSX_LE: return launch_select_cmp<T, QSX_LE>(col, rhs_col, n, literal, filter, out, out_count, stream);
    case QSX_GT: return launch_select_cmp<T, QSX_GT>(col, rhs_col, n, literal, filter, out, out_count, stream);
    case QSX_GE: return launch_select_cmp<T, QSX_GE>(col, rhs_col, n, literal, filter, out, out_count, stream);
    default: return QSX_ERR_INVALID_ARGUMENT;
  }
}

// ---------------------------------------------------------------------------
// bitmap algebra
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void bitmap_combine_kernel(
    int op, const uint64_t *__restrict__ a, const uint64_t *__restrict__ b, int64_t n,
    uint64_t *__restrict__ out) {
  const int64_t num_words = (n + 63) >> 6;
  const int tail = static_cast<int>(n & 63);
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; w < num_words;
       w += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint64_t x = a[w];
    uint64_t r;
    switch (op) {
      case 0: r = x & b[w]; break;
      case 1: r = x | b[w]; break;
      case 2: r = x & ~b[w]; break;
      default: r = ~x; break;
    }
    if (w == num_words - 1 && tail != 0) r &= ~0ull << (64 - tail);  // trailing bits stay zero
    out[w] = r;
  }
}

__global__ __launch_bounds__(kBlock) void bitmap_count_kernel(const uint64_t *__restrict__ bitmap,
                                                              int64_t num_words,
                                                              unsigned long long *__restrict__ out) {
  unsigned long long c = 0;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; w < num_words;
       w += static_cast<int64_t>(gridDim.x) * kBlock) {
    c += __popcll(bitmap[w]);
  }
  c = wave_reduce_add(c);
  __shared__ unsigned long long block_count;
  if (threadIdx.x == 0) block_count = 0;
  __syncthreads();
  if (lane_id() == 0 && c != 0) atomicAdd(&block_count, c);
  __syncthreads();
  if (threadIdx.x == 0 && block_count != 0) atomicAdd(out, block_count);
}

// ---------------------------------------------------------------------------
// K2: order-preserving compaction.  Tile = 64 bitmap words = 4096 rows, one
// wave per tile.  Pass 1 popcounts tiles, pass 2 scans the tile counts, pass 3
// writes: within a tile the wave walks the 64 words; for a word, lane l owns
// row l, its output slot is tile_offset + prefix(word) + popcount(bits before l).
// ---------------------------------------------------------------------------
constexpr int kTileWords = 64;
constexpr int kSmallTileWords = 16;   // 1024-row tiles: four times the waves for an input of few tiles (a 2-4 MB block)

// tile_words: bitmap words of a tile — kTileWords, or kSmallTileWords for inputs of few tiles (run_compaction).
__global__ __launch_bounds__(kBlock) void tile_count_kernel(const uint64_t *__restrict__ bitmap,
                                                            int64_t num_words, int64_t num_tiles,
                                                            int32_t *__restrict__ tile_counts, int tile_words = kTileWords) {
  const int lane = lane_id();
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
       tile < num_tiles; tile += static_cast<int64_t>(gridDim.x) * kWavesPerBlock) {
    const int64_t w = tile * tile_words + lane;
    int c = lane < tile_words && w < num_words ? __popcll(bitmap[w]) : 0;
    c = wave_reduce_add(c);
    if (lane == 0) tile_counts[tile] = c;
  }
}

struct GatherArgs {
  int ncols;
  int width[QSX_MAX_COLUMNS];
  const void *src[QSX_MAX_COLUMNS];
  void *dst[QSX_MAX_COLUMNS];
};

__device__ __forceinline__ void copy_value(const void *src, int64_t si, void *dst, int64_t di, int width) {
  switch (width) {
    case 1: static_cast<uint8_t *>(dst)[di] = static_cast<const uint8_t *>(src)[si]; break;
    case 2: static_cast<uint16_t *>(dst)[di] = static_cast<const uint16_t *>(src)[si]; break;
    case 4: static_cast<uint32_t *>(dst)[di] = static_cast<const uint32_t *>(src)[si]; break;
    case 8: static_cast<uint64_t *>(dst)[di] = static_cast<const uint64_t *>(src)[si]; break;
    default: {   // CHAR(n): byte by byte
      const uint8_t *from = static_cast<const uint8_t *>(src) + si * width;
      uint8_t *to = static_cast<uint8_t *>(dst) + di * width;
      for (int b = 0; b < width; ++b) to[b] = from[b];
      break;
    }
  }
}

// One wave per 4096-row tile.  Every lane expands the set bits of its bitmap word into a tile-local position list in LDS
// (at its prefix offset: the list is in row order); the wave then handles 64 selected rows per step with every lane busy
// and contiguous stores — at 1 % selectivity the previous form (8 words per step, lanes = rows of a word) had ~1 of
// 64 lanes working.  Reads are unconditional and batched (4 steps in flight per lane).
//
// compact_tile is one such tile: `tile` counts within the stripe the bitmap belongs to, tile_off is where the tile's rows
// go in the output.  block_cols (kRuns): the column addresses of the block the tile belongs to, instead of args.src.
template <bool kRuns>
__device__ __forceinline__ void compact_tile(const GatherArgs &args, const long long *__restrict__ block_cols,
                                             const uint64_t *__restrict__ bitmap, int64_t num_words, int64_t tile,
                                             int64_t tile_off, int32_t *__restrict__ out_tids, int32_t base_tid,
                                             uint16_t *__restrict__ s_pos_wave, int tile_words = kTileWords) {
  const int lane = lane_id();
  {
    const int64_t w = tile * tile_words + lane;
    uint64_t my_word = lane < tile_words && w < num_words ? bitmap[w] : 0;
    const int pc = __popcll(my_word);
    int incl = pc;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int up = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += up;
    }
    const int total = __shfl(incl, kWave - 1, kWave);
    if (total == 0) return;                      // wave-uniform: 4096 unselected rows
    int at = incl - pc;
    while (my_word != 0) {                       // MSB-first: bit 63 is row 0 of the word
      const int row_in_word = __clzll(static_cast<long long>(my_word));
      s_pos_wave[at++] = static_cast<uint16_t>(lane * 64 + row_in_word);
      my_word &= ~(1ull << (63 - row_in_word));
    }
    // Same-wave LDS hand-off: DS ops of one wave complete in order.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int64_t tile_row0 = tile * tile_words * 64;
    constexpr int kBatch = 4;
    for (int i0 = lane; i0 < total; i0 += kWave * kBatch) {
      int64_t si[kBatch];
      bool live[kBatch];
#pragma unroll
      for (int b = 0; b < kBatch; ++b) {
        const int i = i0 + b * kWave;
        live[b] = i < total;
        si[b] = tile_row0 + (live[b] ? s_pos_wave[i] : s_pos_wave[0]);   // a valid row either way
      }
      if (out_tids != nullptr) {
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
          if (live[b]) out_tids[tile_off + i0 + b * kWave] = static_cast<int32_t>(base_tid + si[b]);
        }
      }
      for (int c = 0; c < args.ncols; ++c) {
        const void *src = kRuns ? as_global(reinterpret_cast<const void *>(block_cols[c])) : args.src[c];
        void *dst = args.dst[c];
        switch (args.width[c]) {               // wave-uniform
          case 4: {
            uint32_t v[kBatch];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) v[b] = static_cast<const uint32_t *>(src)[si[b]];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
              if (live[b]) static_cast<uint32_t *>(dst)[tile_off + i0 + b * kWave] = v[b];
            }
            break;
          }
          case 8: {
            uint64_t v[kBatch];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) v[b] = static_cast<const uint64_t *>(src)[si[b]];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
              if (live[b]) static_cast<uint64_t *>(dst)[tile_off + i0 + b * kWave] = v[b];
            }
            break;
          }
          default:
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
              if (live[b]) copy_value(src, si[b], dst, tile_off + i0 + b * kWave, args.width[c]);
            }
            break;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();             // the list is rewritten by the next tile
  }
}

__global__ __launch_bounds__(kBlock) void compact_gather_kernel(
    GatherArgs args, const uint64_t *__restrict__ bitmap, int64_t num_words, int64_t num_tiles,
    const int64_t *__restrict__ tile_offsets, int32_t *__restrict__ out_tids, int32_t base_tid, int tile_words = kTileWords) {
  __shared__ uint16_t s_pos[kWavesPerBlock][kTileWords * 64];
  const int wave = threadIdx.x >> 6;
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave; tile < num_tiles;
       tile += static_cast<int64_t>(gridDim.x) * kWavesPerBlock) {
    compact_tile<false>(args, nullptr, bitmap, num_words, tile, tile_offsets[tile], out_tids, base_tid, s_pos[wave], tile_words);
  }
}

// K2 over a run of blocks (qsx_compact_gather_blocks): the tiles of all blocks are counted and scanned together, so the
// selected rows of the run land in ONE output stripe per column, block after block, in row order — the way consecutive
// SelectWorkOrders fill an InsertDestination block.  The run table's `filter` column holds each block's bitmap; the
// blocks' column addresses follow the table at word cols_offset ([block * ncols + column]).
__global__ __launch_bounds__(kBlock) void tile_count_runs_kernel(const long long *__restrict__ runs, int32_t *__restrict__ tile_counts) {
  const int lane = lane_id();
  const int num_tiles = static_cast<int>(runs[2]);
  for (int tile = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * kWavesPerBlock + static_cast<int>(threadIdx.x >> 6));
       tile < num_tiles; tile += static_cast<int>(gridDim.x) * kWavesPerBlock) {
    const RunTile at = run_locate(runs, tile);
    const int64_t num_words = (run_rows(runs, at.block) + 63) >> 6;
    const int64_t w = static_cast<int64_t>(at.tile_in_block) * kTileWords + lane;
    int c = w < num_words ? __popcll(run_filter(runs, at.block)[w]) : 0;
    c = wave_reduce_add(c);
    if (lane == 0) tile_counts[tile] = c;
  }
}
__global__ __launch_bounds__(kBlock) void compact_gather_runs_kernel(
    GatherArgs args, const long long *__restrict__ runs, long long cols_offset, const int64_t *__restrict__ tile_offsets,
    int32_t *__restrict__ out_tids) {
  __shared__ uint16_t s_pos[kWavesPerBlock][kTileWords * 64];
  const int wave = threadIdx.x >> 6;
  const int num_tiles = static_cast<int>(runs[2]);
  for (int tile = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * kWavesPerBlock + wave); tile < num_tiles;
       tile += static_cast<int>(gridDim.x) * kWavesPerBlock) {
    const RunTile at = run_locate(runs, tile);
    compact_tile<true>(args, runs + cols_offset + static_cast<long long>(at.block) * args.ncols, run_filter(runs, at.block),
                       (run_rows(runs, at.block) + 63) >> 6, at.tile_in_block, tile_offsets[tile], out_tids,
                       static_cast<int32_t>(run_base(runs, at.block)), s_pos[wave]);
  }
}


// ---------------------------------------------------------------------------
// K1 on the sort column of a sorted column store: the matches of `col OP literal` are one row range (its complement for
// !=), found by two searches instead of a scan — SortColumnPredicateEvaluator::EvaluatePredicateForUncompressedSortColumn
// (storage/ColumnStoreUtil.cpp:40-280: lower_bound / upper_bound on the stripe, then a range of the TupleIdSequence).
// One wave searches 64-ary (lane l probes the l-th of 64 cut points of the current interval: 5 steps for 1 G rows), a
// second kernel writes the bitmap words of the range ANDed with the filter.
// ---------------------------------------------------------------------------
struct SortedBounds {
  long long lower;   // first row with value >= literal
  long long upper;   // first row with value >  literal
};

template <typename T>
__global__ __launch_bounds__(kWave) void sorted_bounds_kernel(const T *__restrict__ col, int64_t n, T literal,
                                                             SortedBounds *__restrict__ out) {
  const int lane = lane_id();
  long long result[2];
  for (int which = 0; which < 2; ++which) {   // 0: lower bound (value < literal goes left), 1: upper bound (value <= literal)
    long long lo = 0, hi = n;                 // the bound lies in [lo, hi]
    while (hi - lo > 0) {
      const long long span = hi - lo;
      const long long step = (span + kWave) / (kWave + 1);              // >= 1
      const long long at = lo + step * (lane + 1) - 1;                   // cut point of this lane (may pass hi - 1)
      bool left = false;                                                 // col[at] sorts before the bound
      if (at < hi) {
        const T v = col[at];
        left = which == 0 ? v < literal : v <= literal;
      }
      const uint64_t m = __ballot(left);
      const int passed = __popcll(m);                                    // cut points left of the bound: a prefix of the lanes
      const long long new_lo = passed == 0 ? lo : lo + step * passed;
      const long long new_hi = passed == kWave ? hi : (lo + step * (passed + 1) - 1 < hi ? lo + step * (passed + 1) - 1 : hi);
      lo = new_lo < hi ? new_lo : hi;
      hi = new_hi;
    }
    result[which] = lo;
  }
  if (lane == 0) {
    out->lower = result[0];
    out->upper = result[1];
  }
}

// The same search on the code stripe of a compressed SORT column (codes ascend with the values: truncation keeps the
// order, a dictionary is sorted): rows with lo <= code < hi are [lower_bound(lo), lower_bound(hi)).
template <typename T>
__global__ __launch_bounds__(kWave) void sorted_code_bounds_kernel(const T *__restrict__ codes, int64_t n, unsigned long long lo_code,
                                                                  unsigned long long hi_code, SortedBounds *__restrict__ out) {
  const int lane = lane_id();
  long long result[2];
  for (int which = 0; which < 2; ++which) {
    const unsigned long long target = which == 0 ? lo_code : hi_code;   // first row with code >= target
    long long lo = 0, hi = n;
    while (hi - lo > 0) {
      const long long span = hi - lo;
      const long long step = (span + kWave) / (kWave + 1);
      const long long at = lo + step * (lane + 1) - 1;
      const bool left = at < hi && static_cast<unsigned long long>(codes[at < hi ? at : hi - 1]) < target;
      const int passed = __popcll(__ballot(left));
      const long long new_lo = passed == 0 ? lo : lo + step * passed;
      const long long new_hi = passed == kWave ? hi : (lo + step * (passed + 1) - 1 < hi ? lo + step * (passed + 1) - 1 : hi);
      lo = new_lo < hi ? new_lo : hi;
      hi = new_hi;
    }
    result[which] = lo;
  }
  if (lane == 0) {
    out->lower = result[0];
    out->upper = result[1] > result[0] ? result[1] : result[0];
  }
}

__global__ __launch_bounds__(kBlock) void sorted_range_bitmap_kernel(const SortedBounds *__restrict__ bounds, int64_t n, int op,
                                                                    const uint64_t *__restrict__ filter,
                                                                    uint64_t *__restrict__ out,
                                                                    unsigned long long *__restrict__ out_count) {
  // [begin, end) = rows matching, or (QSX_NE) rows NOT matching
  long long begin = 0, end = n;
  switch (op) {
    case QSX_EQ: case QSX_NE: begin = bounds->lower; end = bounds->upper; break;
    case QSX_LT: end = bounds->lower; break;
    case QSX_LE: end = bounds->upper; break;
    case QSX_GT: begin = bounds->upper; break;
    default: begin = bounds->lower; break;   // QSX_GE
  }
  const int64_t num_words = (n + 63) >> 6;
  unsigned long long count = 0;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; w < num_words; w += static_cast<int64_t>(gridDim.x) * kBlock) {
    const long long first = w << 6;
    // rows first .. first + 63 of this word that lie in [begin, end), MSB-first
    uint64_t word = 0;
    const long long a = begin > first ? begin - first : 0, b = end < first + 64 ? end - first : 64;
    if (a < b) word = (b - a == 64 ? ~0ull : ((~0ull) >> (64 - (b - a))) << (64 - b));
    if (op == QSX_NE) word = ~word;
    const long long valid = n - first >= 64 ? 64 : n - first;            // trailing bits of the last word stay zero
    if (valid < 64) word &= ~0ull << (64 - valid);
    if (filter != nullptr) word &= filter[w];
    out[w] = word;
    count += __popcll(word);
  }
  if (out_count != nullptr) {   // one atomic per workgroup (same-address atomics are ~12 ns each)
    __shared__ unsigned long long s_part[kBlock / kWave];
    count = wave_reduce_add(count);
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = count;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long all = 0;
      for (int w = 0; w < kBlock / kWave; ++w) all += s_part[w];
      if (all != 0) atomicAdd(out_count, all);
    }
  }
}

// ---- the same over a run of blocks (qsx_select_cmp_sorted_blocks / qsx_select_codes_sorted_blocks) ---------------------------
// Every block of a sorted column store is sorted on its own, so every block has its own bounds: one wave per block searches
// them (64-ary, as above), a second kernel walks the bitmap tiles of the run (64 words per wave, block_runs.hpp) and writes
// each block's range.  For code stripes the per-block code range and comparison (the dictionaries differ from block to
// block: the host rewrites the predicate per block) follow the run table at word `extra`: [3 b] lo, [3 b + 1] hi, [3 b + 2] op.
template <typename T>
__device__ __forceinline__ long long sorted_first_not_before(const T *__restrict__ col, long long n, T literal, bool or_equal) {
  // first row whose value is >= literal (or_equal = false) / > literal (or_equal = true)
  const int lane = lane_id();
  long long lo = 0, hi = n;
  while (hi - lo > 0) {
    const long long span = hi - lo;
    const long long step = (span + kWave) / (kWave + 1);
    const long long at = lo + step * (lane + 1) - 1;
    bool left = false;
    if (at < hi) {
      const T v = col[at];
      left = or_equal ? v <= literal : v < literal;
    }
    const int passed = __popcll(__ballot(left));
    const long long new_lo = passed == 0 ? lo : lo + step * passed;
    const long long new_hi = passed == kWave ? hi : (lo + step * (passed + 1) - 1 < hi ? lo + step * (passed + 1) - 1 : hi);
    lo = new_lo < hi ? new_lo : hi;
    hi = new_hi;
  }
  return lo;
}
template <typename T, bool kCodes>
__global__ __launch_bounds__(kBlock) void sorted_bounds_runs_kernel(const long long *__restrict__ runs, T literal, long long extra,
                                                                   SortedBounds *__restrict__ bounds) {
  const int b = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * kWavesPerBlock + static_cast<int>(threadIdx.x >> 6));
  if (b >= static_cast<int>(runs[0])) return;
  const T *col = run_in<T>(runs, b);
  const long long n = run_rows(runs, b);
  long long lower, upper;
  if constexpr (kCodes) {
    // rows with lo <= code < hi: [first code >= lo, first code >= hi)
    const unsigned long long lo_code = static_cast<unsigned long long>(runs[extra + 3 * b]);
    const unsigned long long hi_code = static_cast<unsigned long long>(runs[extra + 3 * b + 1]);
    const unsigned long long max_code = static_cast<T>(~static_cast<T>(0));
    lower = lo_code > max_code ? n : sorted_first_not_before<T>(col, n, static_cast<T>(lo_code), false);
    upper = hi_code > max_code ? n : sorted_first_not_before<T>(col, n, static_cast<T>(hi_code), false);
    if (upper < lower) upper = lower;
  } else {
    lower = sorted_first_not_before<T>(col, n, literal, false);
    upper = sorted_first_not_before<T>(col, n, literal, true);
  }
  if (lane_id() == 0) {
    bounds[b].lower = lower;
    bounds[b].upper = upper;
  }
}
__global__ __launch_bounds__(kBlock) void sorted_range_bitmap_runs_kernel(const long long *__restrict__ runs, const SortedBounds *__restrict__ bounds,
                                                                         int op_all, long long extra,
                                                                         unsigned long long *__restrict__ out_counts) {
  const int lane = lane_id();
  const int num_tiles = static_cast<int>(runs[2]);
  for (int tile = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * kWavesPerBlock + static_cast<int>(threadIdx.x >> 6));
       tile < num_tiles; tile += static_cast<int>(gridDim.x) * kWavesPerBlock) {
    const RunTile at = run_locate(runs, tile);
    const long long n = run_rows(runs, at.block);
    const int op = extra != 0 ? (static_cast<int>(runs[extra + 3 * at.block + 2]) == QSX_CODE_NE ? QSX_NE : QSX_EQ) : op_all;
    long long begin = 0, end = n;
    switch (op) {
      case QSX_EQ: case QSX_NE: begin = bounds[at.block].lower; end = bounds[at.block].upper; break;
      case QSX_LT: end = bounds[at.block].lower; break;
      case QSX_LE: end = bounds[at.block].upper; break;
      case QSX_GT: begin = bounds[at.block].upper; break;
      default: begin = bounds[at.block].lower; break;   // QSX_GE
    }
    const long long w = static_cast<long long>(at.tile_in_block) * kTileWords + lane;
    unsigned long long count = 0;
    if (w < ((n + 63) >> 6)) {
      const long long first = w << 6;
      uint64_t word = 0;
      const long long a = begin > first ? begin - first : 0, b = end < first + 64 ? end - first : 64;
      if (a < b) word = (b - a == 64 ? ~0ull : ((~0ull) >> (64 - (b - a))) << (64 - b));
      if (op == QSX_NE) word = ~word;
      const long long valid = n - first >= 64 ? 64 : n - first;            // trailing bits of the last word stay zero
      if (valid < 64) word &= ~0ull << (64 - valid);
      const uint64_t *filter = run_filter(runs, at.block);
      if (filter != nullptr) word &= filter[w];
      run_out<uint64_t>(runs, at.block)[w] = word;
      count = __popcll(word);
    }
    if (out_counts != nullptr) {
      count = wave_reduce_add(count);
      if (lane == 0 && count != 0) atomicAdd(&out_counts[at.block], count);
    }
  }
}

// ---------------------------------------------------------------------------
// K5: gather by tuple id
// ---------------------------------------------------------------------------
// 8 row numbers per thread and step: the 8 tid loads are issued together, then the 8 source reads (unconditional: a
// negative tid reads row 0 and is zeroed afterwards), then the 8 stores — one dependent pair per thread and step left
// the gather latency-bound (0.36 ms for 30 M 8-byte values, now 0.2 ms).
template <typename T>
__global__ __launch_bounds__(kBlock) void gather_kernel(const T *__restrict__ src,
                                                        const int32_t *__restrict__ tids, int64_t n,
                                                        T *__restrict__ dst) {
  constexpr int R = 8;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i0 < n; i0 += stride * R) {
    int32_t t[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t i = i0 + r * stride;
      t[r] = i < n ? load_global_nt(&tids[i]) : -1;   // the two streams pass the caches by: what is read again is src
    }
    T v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = src[t[r] < 0 ? 0 : t[r]];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t i = i0 + r * stride;
      if (i < n) store_global_nt(t[r] < 0 ? T() : v[r], &dst[i]);
    }
  }
}

// Attributes of any other width (CHAR(n)): one thread per output byte — consecutive threads write consecutive bytes and
// read the bytes of one source row.
__global__ __launch_bounds__(kBlock) void gather_bytes_kernel(const uint8_t *__restrict__ src, int width, const int32_t *__restrict__ tids,
                                                              int64_t n, uint8_t *__restrict__ dst) {
  const int64_t total = n * width;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; j < total; j += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t row = j / width;
    const int off = static_cast<int>(j - row * width);
    const int32_t t = tids[row];
    dst[j] = t < 0 ? uint8_t(0) : src[static_cast<int64_t>(t) * width + off];
  }
}

constexpr int kMaxSegments = 64;
struct SegmentTable {
  int num;
  const void *ptr[kMaxSegments];
  int64_t first_row[kMaxSegments];
};

template <typename T>
__global__ __launch_bounds__(kBlock) void gather_segmented_kernel(SegmentTable seg, const int32_t *__restrict__ tids,
                                                                  int64_t n, T *__restrict__ dst) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int32_t t = tids[i];
    if (t < 0) {
      dst[i] = T();
      continue;
    }
    int lo = 0, hi = seg.num - 1;  // last segment whose first_row <= t
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (seg.first_row[mid] <= t) lo = mid; else hi = mid - 1;
    }
    dst[i] = static_cast<const T *>(seg.ptr[lo])[t - seg.first_row[lo]];
  }
}

__global__ __launch_bounds__(kBlock) void gather_segmented_bytes_kernel(SegmentTable seg, int width, const int32_t *__restrict__ tids,
                                                                        int64_t n, uint8_t *__restrict__ dst) {
  const int64_t total = n * width;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; j < total; j += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t row = j / width;
    const int off = static_cast<int>(j - row * width);
    const int32_t t = tids[row];
    if (t < 0) {
      dst[j] = 0;
      continue;
    }
    int lo = 0, hi = seg.num - 1;  // last segment whose first_row <= t
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (seg.first_row[mid] <= t) lo = mid; else hi = mid - 1;
    }
    dst[j] = static_cast<const uint8_t *>(seg.ptr[lo])[(t - seg.first_row[lo]) * width + off];
  }
}

// Segments of equal length (the blocks of a relation, all but the last full): segment = tid / rows_per_segment, no search.
template <typename T>
__global__ __launch_bounds__(kBlock) void gather_uniform_segments_kernel(const long long *__restrict__ ptrs, int num, int32_t rows_per_segment,
                                                                         int shift, const int32_t *__restrict__ tids, int64_t n,
                                                                         T *__restrict__ dst) {
  constexpr int R = 4;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i0 < n; i0 += stride * R) {
    int32_t t[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t i = i0 + r * stride;
      t[r] = i < n ? tids[i] : -1;
    }
    T v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int32_t tt = t[r] < 0 ? 0 : t[r];
      int seg = shift >= 0 ? tt >> shift : tt / rows_per_segment;
      seg = seg < num ? seg : num - 1;
      v[r] = as_global(reinterpret_cast<const T *>(ptrs[seg]))[tt - seg * rows_per_segment];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t i = i0 + r * stride;
      if (i < n) dst[i] = t[r] < 0 ? T() : v[r];
    }
  }
}

// More than kMaxSegments segments (a long run of blocks): the table lives in device memory — first rows as 32-bit words
// (tuple ids are int32), then the segment addresses — and every workgroup copies the first rows to LDS for the search.
constexpr int kMaxTableSegments = 16384;   // 64 KiB of LDS
__device__ __forceinline__ int segment_of(const int32_t *__restrict__ s_first, int num, int32_t t) {
  int lo = 0, hi = num - 1;                // last segment whose first row <= t
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (s_first[mid] <= t) lo = mid; else hi = mid - 1;
  }
  return lo;
}
template <typename T>
__global__ __launch_bounds__(kBlock) void gather_segmented_table_kernel(const int32_t *__restrict__ first_rows,
                                                                        const long long *__restrict__ ptrs, int num,
                                                                        const int32_t *__restrict__ tids, int64_t n,
                                                                        T *__restrict__ dst) {
  extern __shared__ int32_t s_first[];
  for (int i = threadIdx.x; i < num; i += kBlock) s_first[i] = first_rows[i];
  __syncthreads();
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int32_t t = tids[i];
    if (t < 0) {
      dst[i] = T();
      continue;
    }
    const int seg = segment_of(s_first, num, t);
    dst[i] = reinterpret_cast<const T *>(ptrs[seg])[t - s_first[seg]];
  }
}
__global__ __launch_bounds__(kBlock) void gather_segmented_table_bytes_kernel(const int32_t *__restrict__ first_rows,
                                                                              const long long *__restrict__ ptrs, int num, int width,
                                                                              const int32_t *__restrict__ tids, int64_t n,
                                                                              uint8_t *__restrict__ dst) {
  extern __shared__ int32_t s_first[];
  for (int i = threadIdx.x; i < num; i += kBlock) s_first[i] = first_rows[i];
  __syncthreads();
  const int64_t total = n * width;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; j < total; j += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t row = j / width;
    const int off = static_cast<int>(j - row * width);
    const int32_t t = tids[row];
    if (t < 0) {
      dst[j] = 0;
      continue;
    }
    const int seg = segment_of(s_first, num, t);
    dst[j] = reinterpret_cast<const uint8_t *>(ptrs[seg])[static_cast<int64_t>(t - s_first[seg]) * width + off];
  }
}
__global__ __launch_bounds__(kBlock) void bitmap_gather_segmented_table_kernel(const int32_t *__restrict__ first_rows,
                                                                               const long long *__restrict__ ptrs, int num,
                                                                               const int32_t *__restrict__ tids, int64_t n,
                                                                               uint64_t *__restrict__ out) {
  extern __shared__ int32_t s_first[];
  for (int i = threadIdx.x; i < num; i += kBlock) s_first[i] = first_rows[i];
  __syncthreads();
  const int64_t rounded = (n + kWave - 1) / kWave * kWave;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < rounded;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    bool is_null = false;
    if (i < n) {
      const int32_t t = tids[i];
      if (t < 0) {
        is_null = true;
      } else {
        const int seg = segment_of(s_first, num, t);
        const uint64_t *bits = reinterpret_cast<const uint64_t *>(ptrs[seg]);
        if (bits != nullptr) {
          const int64_t r = t - s_first[seg];
          is_null = msb_bit(bits[r >> 6], static_cast<int>(r & 63));
        }
      }
    }
    const uint64_t word = msb_first(__ballot(is_null));
    if (lane_id() == 0) out[i >> 6] = word;
  }
}

// Null bits travelling with gathered values: bit i of the output = the null bit of row tids[i] in the segment holding it
// (a segment without a bitmap has no NULLs), 1 for a negative tid (outer-join padding).  One row per lane, the wave's
// ballot is the output word.
__global__ __launch_bounds__(kBlock) void bitmap_gather_segmented_kernel(SegmentTable seg, const int32_t *__restrict__ tids,
                                                                        int64_t n, uint64_t *__restrict__ out) {
  const int64_t rounded = (n + kWave - 1) / kWave * kWave;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < rounded;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    bool is_null = false;
    if (i < n) {
      const int32_t t = tids[i];
      if (t < 0) {
        is_null = true;
      } else {
        int lo = 0, hi = seg.num - 1;
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (seg.first_row[mid] <= t) lo = mid; else hi = mid - 1;
        }
        const uint64_t *bits = static_cast<const uint64_t *>(seg.ptr[lo]);
        if (bits != nullptr) {
          const int64_t r = t - seg.first_row[lo];
          is_null = msb_bit(bits[r >> 6], static_cast<int>(r & 63));
        }
      }
    }
    const uint64_t word = msb_first(__ballot(is_null));
    if (lane_id() == 0) out[i >> 6] = word;
  }
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int run_compaction(const GatherArgs &args, const uint64_t *bitmap, int64_t n,
                          int32_t *out_tids, int32_t base_tid, int64_t *out_count, void *workspace,
                          size_t workspace_bytes, hipStream_t stream) {
  const int64_t num_words = (n + 63) >> 6;
  // A wave walks its tile's selected rows 256 at a time and column by column: ~2 us of memory latency per step that only other
  // waves can hide.  A 2-4 MB block is 30 tiles of 4096 rows — 30 waves on 256 CUs, 30 us per call — so inputs of few tiles
  // are cut into 1024-row tiles instead (four times the waves, a quarter of the steps each).
  const int tile_words = (num_words + kTileWords - 1) / kTileWords < 8 * kCUs ? kSmallTileWords : kTileWords;
  const int64_t num_tiles = (num_words + tile_words - 1) / tile_words;
  if (workspace_bytes < qsx_compact_workspace_bytes(n)) return QSX_ERR_CAPACITY;
  if (n == 0) {
    if (out_count != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(int64_t), stream));
    return QSX_OK;
  }
  int64_t *tile_offsets = static_cast<int64_t *>(workspace);
  int32_t *tile_counts = reinterpret_cast<int32_t *>(
      static_cast<char *>(workspace) + align_up(sizeof(int64_t) * (num_tiles + 1), 256));
  hipLaunchKernelGGL(tile_count_kernel, dim3(grid_for(num_tiles, kWavesPerBlock)), dim3(kBlock), 0,
                     stream, bitmap, num_words, num_tiles, tile_counts, tile_words);
  QSX_CHECK_LAUNCH();
  hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, stream, tile_counts, num_tiles,
                     tile_offsets, out_count);
  QSX_CHECK_LAUNCH();
  hipLaunchKernelGGL(compact_gather_kernel, dim3(grid_for(num_tiles, kWavesPerBlock)), dim3(kBlock),
                     0, stream, args, bitmap, num_words, num_tiles, tile_offsets, out_tids, base_tid, tile_words);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

// ---------------------------------------------------------------------------
// K1 on compressed attributes: comparisons on the code stripe (1/2/4-byte unsigned codes of a
// dictionary-coded or truncated attribute), storage/CompressedColumnStoreTupleStorageSubBlock.cpp:
// getEqualCodes / getNotEqualCodes / getLessCodes / getGreaterOrEqualCodes / getCodesInRange.
// Same wave layout as select_cmp_kernel; a 1-byte code column moves a quarter of the bytes of the INT
// column it stands for.
// ---------------------------------------------------------------------------
template <typename T, int R>
__global__ __launch_bounds__(kBlock) void select_codes_kernel(const T *__restrict__ codes, int64_t n, int op, uint32_t first,
                                                             uint32_t second, const uint64_t *__restrict__ filter,
                                                             uint64_t *__restrict__ out,
                                                             unsigned long long *__restrict__ out_count) {
  const int lane = lane_id();
  const int64_t num_words = (n + 63) >> 6;
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t num_waves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  // every operator is one range test lo <= code < hi on 64-bit bounds, NE its complement
  // (selects, not a switch: the switch form left `lo` undefined on the RANGE path in the generated code)
  const unsigned long long lo = op == QSX_CODE_LT ? 0ull : first;
  const unsigned long long hi = (op == QSX_CODE_EQ || op == QSX_CODE_NE) ? static_cast<unsigned long long>(first) + 1
                                : op == QSX_CODE_LT ? first
                                : op == QSX_CODE_GE ? (1ull << 32)
                                                    : second;   // QSX_CODE_RANGE: [first, second)
  const bool negate = op == QSX_CODE_NE;
  unsigned long long count = 0;
  for (int64_t w0 = wave * R; w0 < num_words; w0 += num_waves * R) {
    T v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = ((w0 + r) << 6) + lane;
      v[r] = codes[row < n ? row : n - 1];   // clamped, not guarded
    }
    uint64_t mine = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = ((w0 + r) << 6) + lane;
      const unsigned long long c = v[r];
      bool pred = row < n && ((c >= lo && c < hi) != negate);
      if (filter != nullptr && w0 + r < num_words) pred = pred && msb_bit(filter[w0 + r], lane);
      const uint64_t word = msb_first(__ballot(pred));
      count += __popcll(word);
      if (lane == r) mine = word;
    }
    if (lane < R && w0 + lane < num_words) out[w0 + lane] = mine;
  }
  if (out_count != nullptr) {
    __shared__ unsigned long long block_count;
    if (threadIdx.x == 0) block_count = 0;
    __syncthreads();
    if (lane == 0 && count != 0) atomicAdd(&block_count, count);
    __syncthreads();
    if (threadIdx.x == 0 && block_count != 0) atomicAdd(out_count, block_count);
  }
}

// ---------------------------------------------------------------------------
// K1, packed variant: every lane loads 16 bytes = K = 16 / sizeof(T) consecutive rows, so one wave load
// covers 64 * K rows (16 bitmap words for 1-byte codes) instead of 64 — a byte-wide column with one row
// per lane is bound by the number of memory instructions, not by HBM (0.138 ms / 100 M 1-byte codes against
// 0.10 ms for the 4x larger INT column).  A lane turns its K rows into a K-bit MSB-first mask; the 64 / K
// lanes that share a bitmap word merge their masks with log2(64 / K) xor-shuffles, the group's first lane
// ANDs the filter word and stores.  Needs a 16-byte aligned stripe (the callers fall back otherwise).
// ---------------------------------------------------------------------------
struct CodeRangePred {   // lo <= code < hi, optionally negated (the five comparisons of qsx_select_codes)
  unsigned long long lo, hi;
  bool negate;
  template <typename T>
  __device__ __forceinline__ bool operator()(T v) const {
    const unsigned long long c = v;
    return (c >= lo && c < hi) != negate;
  }
};
template <typename T, int OP>
struct LiteralPred {     // value OP literal
  T lit;
  __device__ __forceinline__ bool operator()(T v) const { return cmp_static<T, OP>(v, lit); }
};

// One group of R wave loads starting at load l0 of a stripe (col, n): the rows' bits go to out, the matches are added to count.
template <typename T, typename Pred, int R>
__device__ __forceinline__ void select_packed_group(const T *__restrict__ col, int64_t n, const Pred &pred,
                                                    const uint64_t *__restrict__ filter, uint64_t *__restrict__ out,
                                                    int64_t l0, int lane, unsigned long long &count) {
  constexpr int K = 16 / sizeof(T);        // rows per lane per load
  constexpr int G = kWave / K;             // lanes per bitmap word
  constexpr int kRowsPerLoad = kWave * K;  // rows per wave load = K bitmap words
  const int64_t num_words = (n + 63) >> 6;
  uint4 raw[R];
  if (sizeof(T) <= 2 && (l0 + R) * kRowsPerLoad <= n) {
    // code stripes (8 / 16 rows per 16-byte read): every row of the group exists (all groups but the last), so R unguarded
    // reads and no per-row bounds test — 0.059 -> 0.043 ms per 100 M one-byte codes; for 4 / 8-byte values the same split
    // measured slower (0.082 -> 0.094 ms), so they keep the single guarded form
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row0 = (l0 + r) * kRowsPerLoad + static_cast<int64_t>(lane) * K;
      raw[r] = stream_load16(col + row0);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      T v[K];
      *reinterpret_cast<uint4 *>(v) = raw[r];
      unsigned long long m = 0;            // K-bit mask, first row = most significant bit
#pragma unroll
      for (int i = 0; i < K; ++i) m = (m << 1) | (pred(v[i]) ? 1ull : 0ull);
#pragma unroll
      for (int d = 1; d < G; d <<= 1) {
        const unsigned long long other = __shfl_xor(m, d, kWave);
        m = (m << (d * K)) | other;
      }
      const int64_t word = (l0 + r) * K + lane / G;
      if ((lane % G) == 0) {
        if (filter != nullptr) m &= filter[word];
        out[word] = m;
        count += __popcll(m);
      }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t row0 = (l0 + r) * kRowsPerLoad + static_cast<int64_t>(lane) * K;
    raw[r] = make_uint4(0, 0, 0, 0);
    if (row0 + K <= n) {
      raw[r] = stream_load16(col + row0);
    } else if (row0 < n) {               // the last, partial 16 bytes of the stripe: element by element
      T tmp[K];
#pragma unroll
      for (int i = 0; i < K; ++i) tmp[i] = row0 + i < n ? col[row0 + i] : T();
      raw[r] = *reinterpret_cast<const uint4 *>(tmp);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t row0 = (l0 + r) * kRowsPerLoad + static_cast<int64_t>(lane) * K;
    T v[K];
    *reinterpret_cast<uint4 *>(v) = raw[r];
    unsigned long long m = 0;            // K-bit mask, first row = most significant bit
#pragma unroll
    for (int i = 0; i < K; ++i) m = (m << 1) | ((row0 + i < n && pred(v[i])) ? 1ull : 0ull);
    // merge the G lanes of a word: after step d the lower lane of every 2d-group holds 2d * K bits
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
      const unsigned long long other = __shfl_xor(m, d, kWave);
      m = (m << (d * K)) | other;        // only meaningful in lanes whose bit d is clear; those are the ones kept
    }
    const int64_t word = (l0 + r) * K + lane / G;
    if ((lane % G) == 0 && word < num_words) {
      if (filter != nullptr) m &= filter[word];
      out[word] = m;
      count += __popcll(m);
    }
  }
}

template <typename T, typename Pred, int R>
__global__ __launch_bounds__(kBlock) void select_packed_kernel(const T *__restrict__ col, int64_t n, Pred pred,
                                                              const uint64_t *__restrict__ filter,
                                                              uint64_t *__restrict__ out,
                                                              unsigned long long *__restrict__ out_count) {
  constexpr int kRowsPerLoad = kWave * (16 / static_cast<int>(sizeof(T)));
  const int lane = lane_id();
  const int64_t num_loads = (n + kRowsPerLoad - 1) / kRowsPerLoad;
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t num_waves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  unsigned long long count = 0;
  for (int64_t l0 = wave * R; l0 < num_loads; l0 += num_waves * R) {
    select_packed_group<T, Pred, R>(col, n, pred, filter, out, l0, lane, count);
  }
  if (out_count != nullptr) {
    __shared__ unsigned long long block_count;
    if (threadIdx.x == 0) block_count = 0;
    __syncthreads();
    count = wave_reduce_add(count);
    if (lane == 0 && count != 0) atomicAdd(&block_count, count);
    __syncthreads();
    if (threadIdx.x == 0 && block_count != 0) atomicAdd(out_count, block_count);
  }
}

// K1 over a run of blocks (qsx_select_cmp_blocks): a wave's unit of work is one group of R loads of ONE block — the
// table (block_runs.hpp) counts "tiles" of R * kRowsPerLoad rows — and every block keeps its own stripe, filter and output
// bitmap.  A workgroup takes a contiguous range of the run's tiles (its waves interleaved inside it), so it meets few
// blocks: a wave adds its matches to a block's counter when it moves on to another block, and the counts of the range's
// last block meet in LDS first — about one atomic per workgroup plus four per block.  (Counters of neighbouring blocks
// share a cache line and same-line atomics serialise in L2: one atomic per wave and tile visit made 12 blocks of 10 M rows
// take 0.89 ms instead of 0.25.)
// extra != 0 (qsx_select_codes_blocks, Pred = CodeRangePred): the predicate of block b — the comparison rewritten on that block's
// own codes — sits behind the run table at word extra + 3 b: lo, hi, negate.
template <typename T, typename Pred, int R>
__global__ __launch_bounds__(kBlock) void select_packed_runs_kernel(const long long *__restrict__ runs, Pred pred_all,
                                                                   unsigned long long *__restrict__ out_counts, long long extra = 0) {
  __shared__ unsigned long long s_last_count;
  const int lane = lane_id();
  const long long num_tiles = runs[2];
  const int first = static_cast<int>(num_tiles * blockIdx.x / gridDim.x);
  const int end = static_cast<int>(num_tiles * (blockIdx.x + 1) / gridDim.x);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  if (threadIdx.x == 0) s_last_count = 0;
  unsigned long long count = 0;
  int counted_block = -1;
  for (int tile = first + wave; tile < end; tile += kWavesPerBlock) {
    const RunTile at = run_locate(runs, tile);
    if (at.block != counted_block) {
      if (out_counts != nullptr && counted_block >= 0) {
        count = wave_reduce_add(count);
        if (lane == 0 && count != 0) atomicAdd(&out_counts[counted_block], count);
      }
      count = 0;
      counted_block = at.block;
    }
    Pred pred = pred_all;
    if constexpr (std::is_same<Pred, CodeRangePred>::value) {
      if (extra != 0) {
        pred.lo = static_cast<unsigned long long>(runs[extra + 3 * at.block]);
        pred.hi = static_cast<unsigned long long>(runs[extra + 3 * at.block + 1]);
        pred.negate = runs[extra + 3 * at.block + 2] != 0;
      }
    }
    select_packed_group<T, Pred, R>(run_in<T>(runs, at.block), run_rows(runs, at.block), pred, run_filter(runs, at.block),
                                    run_out<uint64_t>(runs, at.block), static_cast<int64_t>(at.tile_in_block) * R, lane, count);
  }
  if (out_counts == nullptr) return;
  const int last_block = end > first ? run_locate(runs, end - 1).block : -1;
  __syncthreads();                       // s_last_count is zero
  count = wave_reduce_add(count);
  if (lane == 0 && count != 0 && counted_block >= 0) {
    if (counted_block == last_block) atomicAdd(&s_last_count, count);
    else atomicAdd(&out_counts[counted_block], count);
  }
  __syncthreads();
  if (threadIdx.x == 0 && last_block >= 0 && s_last_count != 0) atomicAdd(&out_counts[last_block], s_last_count);
}

template <typename T, typename Pred>
static int launch_select_packed(const void *col, int64_t n, Pred pred, const uint64_t *filter, uint64_t *out, int64_t *out_count,
                                hipStream_t stream) {
  constexpr int R = 4;
  constexpr int kRowsPerLoad = kWave * (16 / static_cast<int>(sizeof(T)));
  const int64_t num_loads = (n + kRowsPerLoad - 1) / kRowsPerLoad;
  const int grid = grid_for(num_loads, kWavesPerBlock * R);
  hipLaunchKernelGGL((select_packed_kernel<T, Pred, R>), dim3(grid), dim3(kBlock), 0, stream, static_cast<const T *>(col), n, pred,
                     filter, out, reinterpret_cast<unsigned long long *>(out_count));
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

// May a stripe that starts at p take the kernels that read 16 bytes per lane?  Any stripe may: gfx950 under this stack serves
// unaligned 16-byte global loads (tools/unaligned_probe.py), and the stripes of a reference block image start at multiples of the
// block's tuple capacity, aligned to nothing — a run of 2726 adopted lineitem blocks took 2726 row-per-lane launches per term
// before (10 ms per 600 M rows instead of 0.3).  QSX_SELECT_ALIGNED_ONLY=1 restores the 16-byte requirement.
static bool aligned16(const void *p) {
  static const bool strict = [] { const char *e = getenv("QSX_SELECT_ALIGNED_ONLY"); return e != nullptr && e[0] == '1'; }();
  return !strict || (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}

template <typename T, int OP>
static int launch_select_runs(const long long *runs_dev, long long tiles, const void *literal, int64_t *out_counts, hipStream_t stream) {
  constexpr int R = 4;
  T lit = T();
  std::memcpy(&lit, literal, sizeof(T));
  const int grid = grid_for(tiles, kWavesPerBlock);
  hipLaunchKernelGGL((select_packed_runs_kernel<T, LiteralPred<T, OP>, R>), dim3(grid), dim3(kBlock), 0, stream, runs_dev,
                     LiteralPred<T, OP>{lit}, reinterpret_cast<unsigned long long *>(out_counts));
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}
template <typename T>
static int dispatch_select_runs(int op, const long long *runs_dev, long long tiles, const void *literal, int64_t *out_counts,
                                hipStream_t stream) {
  switch (op) {
    case QSX_EQ: return launch_select_runs<T, QSX_EQ>(runs_dev, tiles, literal, out_counts, stream);
    case QSX_NE: return launch_select_runs<T, QSX_NE>(runs_dev, tiles, literal, out_counts, stream);
    case QSX_LT: return launch_select_runs<T, QSX_LT>(runs_dev, tiles, literal, out_counts, stream);
    case QSX_LE: return launch_select_runs<T, QSX_LE>(runs_dev, tiles, literal, out_counts, stream);
    case QSX_GT: return launch_select_runs<T, QSX_GT>(runs_dev, tiles, literal, out_counts, stream);
    case QSX_GE: return launch_select_runs<T, QSX_GE>(runs_dev, tiles, literal, out_counts, stream);
    default: return QSX_ERR_INVALID_ARGUMENT;
  }
}

// ---------------------------------------------------------------------------
// K1 on a CHAR(width) stripe (qsx_select_cmp_char): a tile of tile_rows x width bytes is copied to LDS with 16-byte
// reads (a row per lane straight from HBM would read `width`-strided bytes), then every lane compares its row with the
// literal the way AsciiStringUncheckedComparator::strcmpHelper does (AsciiStringComparators.hpp:218-251): C strings
// ending at the first NUL or at their maximum length, unsigned bytes, a proper prefix is smaller.
// ---------------------------------------------------------------------------
constexpr int kCharUnroll = 4;
constexpr int kCharStripWords = 16 * 16 + 8;   // a wave's strip for one bitmap word of CHAR(16) rows, and the spare word
constexpr size_t kCharDirectLds = static_cast<size_t>(kWavesPerBlock) * kCharUnroll * kCharStripWords * 4;
struct CharLiteral {
  unsigned char bytes[QSX_MAX_CHAR_LITERAL];
  int length;   // bytes before the first NUL
};
// All ones up to (not including) the first zero byte of x (byte 0 = the lowest), all ones when there is none.  (The classic
// zero-byte test flags bytes ABOVE a zero byte by mistake now and then; its lowest flag is always a true one.)
__device__ __forceinline__ unsigned long long up_to_first_zero_byte(unsigned long long x) {
  const unsigned long long z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
  return z != 0ull ? ((z & (0ull - z)) - 1ull) : ~0ull;
}
// One tile (rows [row0, row0 + tile_rows) of a stripe) of the CHAR comparison; lane 0 of every wave adds its words' bits to count.
__device__ __forceinline__ void select_char_tile(const unsigned char *__restrict__ col, int width, int64_t n, int op, const CharLiteral &lit,
                                                 const uint64_t *__restrict__ filter, uint64_t *__restrict__ out, int64_t row0,
                                                 int tile_rows, unsigned char *s_tile, unsigned long long &count) {
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int rows = static_cast<int>(n - row0 < tile_rows ? n - row0 : tile_rows);
  if (width <= 16 && lit.length <= 16 && (reinterpret_cast<uintptr_t>(col) & 3) == 0) {
    // short fields (c_mktsegment CHAR(10), l_shipmode CHAR(10), flags): no workgroup-wide staging, no barrier — wave by wave
    // (below); a row is compared in registers, 16 bytes from its first byte on.  Through a tile in LDS a workgroup pays load ->
    // store -> barrier -> a byte-at-a-time walk per tile: 0.236 ms per 25 M CHAR(10) rows.
    const uint32_t *stripe_words = reinterpret_cast<const uint32_t *>(col);
    const long long last_word = ((n * width + 3) >> 2) - 1;
    constexpr int kUnroll = kCharUnroll;   // bitmap words per wave and step: the loads of four rows per lane are in flight together
    // The 64 rows of a bitmap word are 64 * width contiguous bytes from a 64-byte aligned offset of the stripe: the wave reads them
    // with coalesced 4-byte loads (16 * width words: every cache line once) into a strip of LDS of its own and every lane
    // picks its row's five words from there.  (Five 4-byte loads per lane straight from the stripe, at a lane stride of `width`
    // bytes, touched every line five times over: 0.153 ms per 25 M CHAR(10) rows, a fifth of the HBM peak.)
    uint32_t *const strips = reinterpret_cast<uint32_t *>(s_tile) + static_cast<size_t>(wave) * kUnroll * kCharStripWords;
    const int strip_words = 16 * width + 1;   // (+ 1: a row's fifth word may lie behind the last row)
    const unsigned long long width_mask_lo = width >= 8 ? ~0ull : ((1ull << (8 * width)) - 1ull);
    const unsigned long long width_mask_hi = width >= 16 ? ~0ull : (width > 8 ? ((1ull << (8 * (width - 8))) - 1ull) : 0ull);
    unsigned long long lit_first = 0, lit_second = 0;   // the literal's first / second eight bytes, big-endian, zero-padded
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      lit_first |= static_cast<unsigned long long>(i < lit.length ? lit.bytes[i] : 0) << (8 * (7 - i));
      lit_second |= static_cast<unsigned long long>(8 + i < lit.length ? lit.bytes[8 + i] : 0) << (8 * (7 - i));
    }
    // words a row can touch from its first byte's word on: 2 (width <= 4), 3 (<= 8), 5 (<= 16) — also the rounds of 64 lanes
    // that load a bitmap word's rows
    auto rows_of_wave = [&](auto kw_tag) {
      constexpr int KW = decltype(kw_tag)::value;
      for (int w0 = wave * kUnroll; w0 * 64 < rows; w0 += kWavesPerBlock * kUnroll) {
        uint32_t x[kUnroll][KW];
        int shift[kUnroll];
        {
          uint32_t g[kUnroll][KW];   // strip_words <= 64 (KW - 1) + 1: KW rounds of 64 lanes
  #pragma unroll
          for (int u = 0; u < kUnroll; ++u) {
            const long long w_base = ((row0 + static_cast<long long>(w0 + u) * 64) * width) >> 2;
  #pragma unroll
            for (int k = 0; k < KW; ++k) {
              const long long at = w_base + k * 64 + lane;
              g[u][k] = k * 64 + lane < strip_words ? stripe_words[at <= last_word ? at : last_word] : 0u;   // clamped: bytes past the stripe are masked
            }
          }
  #pragma unroll
          for (int u = 0; u < kUnroll; ++u) {
  #pragma unroll
            for (int k = 0; k < KW; ++k) {
              if (k * 64 + lane < strip_words) strips[u * kCharStripWords + k * 64 + lane] = g[u][k];
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  #pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const int first_byte = lane * width;   // inside the strip
          shift[u] = (first_byte & 3) * 8;
  #pragma unroll
          for (int k = 0; k < KW; ++k) x[u][k] = strips[u * kCharStripWords + (first_byte >> 2) + k];
        }
        __builtin_amdgcn_wave_barrier();   // the strips are rewritten by the next step
  #pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const int w = w0 + u;
          if (w * 64 >= rows) break;   // wave-uniform
          const int r = w * 64 + lane;
          const unsigned long long q0 = x[u][0] | (static_cast<unsigned long long>(x[u][1]) << 32);
          unsigned long long lo, hi = 0;
          if constexpr (KW == 5) {
            const unsigned long long q1 = x[u][2] | (static_cast<unsigned long long>(x[u][3]) << 32);
            lo = shift[u] != 0 ? (q0 >> shift[u]) | (q1 << (64 - shift[u])) : q0;
            hi = shift[u] != 0 ? (q1 >> shift[u]) | (static_cast<unsigned long long>(x[u][4]) << (64 - shift[u])) : q1;
          } else if constexpr (KW == 3) {   // width <= 8: the row ends inside the third word
            lo = shift[u] != 0 ? (q0 >> shift[u]) | (static_cast<unsigned long long>(x[u][2]) << (64 - shift[u])) : q0;
          } else {                          // width <= 4: inside the second
            lo = q0 >> shift[u];
          }
          // strcmp on zero-padded 16-byte strings = an unsigned compare of their bytes read as one big-endian number: bytes from
          // the field's width on and everything behind its first NUL are cleared (the literal's were, by the host), then two
          // 64-bit compares.  (A byte-at-a-time walk with a "decided" flag was ~130 instructions per row: the kernel took
          // 0.11 ms per 25 M rows whatever the width — CHAR(1) included.)
          const unsigned long long a_lo = lo & width_mask_lo, a_hi = hi & width_mask_hi;
          const unsigned long long keep_lo = up_to_first_zero_byte(a_lo);
          const unsigned long long s_lo = a_lo & keep_lo;
          unsigned long long s_hi = 0;
          if constexpr (KW == 5) s_hi = keep_lo == ~0ull ? (a_hi & up_to_first_zero_byte(a_hi)) : 0ull;
          const unsigned long long x_first = __builtin_bswap64(s_lo), x_second = __builtin_bswap64(s_hi);
          const int res = x_first != lit_first ? (x_first < lit_first ? -1 : 1) : (x_second != lit_second ? (x_second < lit_second ? -1 : 1) : 0);
          const bool pred = r < rows && compare_op<int>(res, op, 0);
          uint64_t word = msb_first(__ballot(pred));
          const int64_t word_index = (row0 >> 6) + w;
          if (filter != nullptr) word &= filter[word_index];
          if (lane == 0) {
            out[word_index] = word;
            count += __popcll(word);
          }
        }
      }
    };
    if (width <= 4) rows_of_wave(std::integral_constant<int, 2>{});
    else if (width <= 8) rows_of_wave(std::integral_constant<int, 3>{});
    else rows_of_wave(std::integral_constant<int, 5>{});
    return;
  }
  const unsigned char *src = col + row0 * width;
  const int bytes = rows * width;
  __syncthreads();   // every wave is done with the previous tile
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    const int full = bytes & ~15;
    for (int o = threadIdx.x * 16; o < full; o += kBlock * 16) {
      *reinterpret_cast<uint4 *>(s_tile + o) = stream_load16(src + o);
    }
    for (int o = full + threadIdx.x; o < bytes; o += kBlock) s_tile[o] = src[o];
  } else {
    for (int o = threadIdx.x; o < bytes; o += kBlock) s_tile[o] = src[o];
  }
  __syncthreads();
  for (int w = wave; w * 64 < rows; w += kWavesPerBlock) {
    const int r = w * 64 + lane;
    bool pred = false;
    if (r < rows) {
      const unsigned char *v = s_tile + r * width;
      int res = 0;
      {
        const int longest = width > lit.length ? width : lit.length;
        for (int i = 0; i < longest; ++i) {
          const unsigned char a = i < width ? v[i] : 0;
          const unsigned char b = i < lit.length ? lit.bytes[i] : 0;
          if (a != b) {
            res = a < b ? -1 : 1;
            break;
          }
          if (a == 0) break;
        }
      }
      pred = compare_op<int>(res, op, 0);
    }
    uint64_t word = msb_first(__ballot(pred));
    const int64_t word_index = (row0 >> 6) + w;
    if (filter != nullptr) word &= filter[word_index];
    if (lane == 0) {
      out[word_index] = word;
      count += __popcll(word);
    }
  }
}

__global__ __launch_bounds__(kBlock) void select_char_kernel(const unsigned char *__restrict__ col, int width, int64_t n, int op,
                                                            CharLiteral lit, const uint64_t *__restrict__ filter,
                                                            uint64_t *__restrict__ out, unsigned long long *__restrict__ out_count,
                                                            int tile_rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_tile[];
  const int lane = lane_id();
  const int64_t num_tiles = (n + tile_rows - 1) / tile_rows;
  unsigned long long count = 0;
  for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    select_char_tile(col, width, n, op, lit, filter, out, tile * tile_rows, tile_rows, s_tile, count);
  }
  if (out_count != nullptr) {
    __shared__ unsigned long long block_count;
    if (threadIdx.x == 0) block_count = 0;
    __syncthreads();
    if (lane == 0 && count != 0) atomicAdd(&block_count, count);
    __syncthreads();
    if (threadIdx.x == 0 && block_count != 0) atomicAdd(out_count, block_count);
  }
}

// The same over a run of blocks (qsx_select_cmp_char_blocks): a workgroup takes a contiguous range of the run's tiles, a wave
// adds its matches to a block's counter when the workgroup moves on to another block (as select_packed_runs_kernel).
__global__ __launch_bounds__(kBlock) void select_char_runs_kernel(const long long *__restrict__ runs, int width, int op, CharLiteral lit,
                                                                 unsigned long long *__restrict__ out_counts, int tile_rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_tile[];
  const int lane = lane_id();
  const long long num_tiles = runs[2];
  const int first = static_cast<int>(num_tiles * blockIdx.x / gridDim.x);
  const int end = static_cast<int>(num_tiles * (blockIdx.x + 1) / gridDim.x);
  unsigned long long count = 0;
  int counted_block = -1;
  for (int tile = first; tile < end; ++tile) {
    const RunTile at = run_locate(runs, tile);
    if (at.block != counted_block) {
      if (out_counts != nullptr && counted_block >= 0 && lane == 0 && count != 0) atomicAdd(&out_counts[counted_block], count);
      count = 0;
      counted_block = at.block;
    }
    select_char_tile(run_in<unsigned char>(runs, at.block), width, run_rows(runs, at.block), op, lit, run_filter(runs, at.block),
                     run_out<uint64_t>(runs, at.block), static_cast<int64_t>(at.tile_in_block) * tile_rows, tile_rows, s_tile, count);
  }
  if (out_counts != nullptr && counted_block >= 0 && lane == 0 && count != 0) atomicAdd(&out_counts[counted_block], count);
}

// Decode a code stripe: dictionary lookup (codes index a dictionary of `value_width`-byte values that
// stays in L2 / L1) or zero-extension of a truncated value.
template <typename C, typename V>
__global__ __launch_bounds__(kBlock) void decode_codes_kernel(const C *__restrict__ codes, int64_t n,
                                                             const V *__restrict__ dictionary, V *__restrict__ out) {
  // four coalesced code reads in flight per thread, values written past the caches (both stripes are touched once)
  constexpr int U = 4;
  for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * (kBlock * U) + threadIdx.x; i0 < n;
       i0 += static_cast<int64_t>(gridDim.x) * (kBlock * U)) {
    C c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * kBlock;
      c[u] = load_global_nt(&codes[i < n ? i : n - 1]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * kBlock;
      if (i < n) store_global_nt(dictionary != nullptr ? dictionary[c[u]] : static_cast<V>(c[u]), &out[i]);
    }
  }
}

// The same with 16 bytes of values per lane and store (K = 2 DOUBLE / LONG or 4 INT / FLOAT values): the K codes come with one
// read of K * sizeof(C) bytes.  A decode is nine tenths writing; 8-byte stores of a lane reach 3.5 TB/s here, 16-byte stores
// what a copy writes at.  Needs the code stripe aligned to its K-code groups and the value stripe to 16 bytes; the rows behind
// the last whole group (fewer than K) are the scalar kernel's.
template <typename C, typename V>
__global__ __launch_bounds__(kBlock) void decode_codes_packed_kernel(const C *__restrict__ codes, int64_t groups,
                                                                    const V *__restrict__ dictionary, V *__restrict__ out) {
  constexpr int K = 16 / static_cast<int>(sizeof(V));
  constexpr int kCodeBytes = K * static_cast<int>(sizeof(C));   // 2 .. 16
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr int U = 4;
  for (int64_t g0 = static_cast<int64_t>(blockIdx.x) * (kBlock * U) + threadIdx.x; g0 < groups;
       g0 += static_cast<int64_t>(gridDim.x) * (kBlock * U)) {
    C c[U][K];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t g = g0 + u * kBlock < groups ? g0 + u * kBlock : groups - 1;
      const unsigned char *at = reinterpret_cast<const unsigned char *>(codes) + g * kCodeBytes;
      if constexpr (kCodeBytes == 16) {
        const uint4 bits = stream_load16(at);
        __builtin_memcpy(&c[u][0], &bits, 16);
      } else {
        typedef typename BitsOfSize<kCodeBytes>::type Bits;
        const Bits bits = load_global_nt(reinterpret_cast<const Bits *>(at));
        __builtin_memcpy(&c[u][0], &bits, kCodeBytes);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t g = g0 + u * kBlock;
      if (g >= groups) continue;
      V v[K];
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = dictionary != nullptr ? dictionary[c[u][k]] : static_cast<V>(c[u][k]);
      u32x4 bits;
      __builtin_memcpy(&bits, &v[0], 16);
      __builtin_nontemporal_store(bits, (__attribute__((address_space(1))) u32x4 *)reinterpret_cast<uintptr_t>(out + g * K));
    }
  }
}

template <typename C, typename V>
static void launch_decode_t(const C *codes, int64_t n, const V *dictionary, V *out, hipStream_t s) {
  constexpr int K = 16 / static_cast<int>(sizeof(V));
  static const bool packed_off = getenv("QSX_DECODE_PACKED") != nullptr && atoi(getenv("QSX_DECODE_PACKED")) == 0;
  int64_t done = 0;
  if (!packed_off && n >= K && (reinterpret_cast<uintptr_t>(codes) % (K * sizeof(C))) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const int64_t groups = n / K;
    hipLaunchKernelGGL((decode_codes_packed_kernel<C, V>), dim3(grid_for(groups, kBlock * 4)), dim3(kBlock), 0, s, codes, groups, dictionary, out);
    done = groups * K;
  }
  if (done < n) {
    hipLaunchKernelGGL((decode_codes_kernel<C, V>), dim3(grid_for(n - done, kBlock * 4)), dim3(kBlock), 0, s, codes + done, n - done, dictionary,
                       out + done);
  }
}

template <typename C>
static int launch_decode(const void *codes, int64_t n, const void *dictionary, int value_width, void *out, hipStream_t s) {
  switch (value_width) {
    case 4:
      launch_decode_t<C, uint32_t>(static_cast<const C *>(codes), n, static_cast<const uint32_t *>(dictionary), static_cast<uint32_t *>(out), s);
      QSX_CHECK_LAUNCH();
      return QSX_OK;
    case 8:
      launch_decode_t<C, uint64_t>(static_cast<const C *>(codes), n, static_cast<const uint64_t *>(dictionary), static_cast<uint64_t *>(out), s);
      QSX_CHECK_LAUNCH();
      return QSX_OK;
    default: return QSX_ERR_UNSUPPORTED;
  }
}
// tuple-id list -> TupleIdSequence (bit = tid - base): one atomicOr per tid, skipped when the bit is already set
__global__ __launch_bounds__(kBlock) void tids_to_bitmap_kernel(const int32_t *__restrict__ tids, int64_t n, int32_t base_tid,
                                                               int64_t num_bits, unsigned long long *__restrict__ out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t bit = static_cast<int64_t>(tids[i]) - base_tid;
    if (bit < 0 || bit >= num_bits) continue;
    const unsigned long long mask = 1ull << (63 - (bit & 63));
    unsigned long long *word = &out[bit >> 6];
    if ((__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & mask) == 0) atomicOr(word, mask);
  }
}

}  // namespace qsx

using namespace qsx;

// Shared back end of the two sorted-column run entry points.  extra_words (codes): 3 words per block behind the run table.
template <typename T, bool kCodes>
static int sorted_runs(int64_t num_blocks, const int64_t *block_rows, const void *const *block_cols, T literal, int op,
                       const std::vector<long long> &extra_words, const uint64_t *const *block_filters, uint64_t *const *block_out_bitmaps,
                       int64_t *out_counts_dev, hipStream_t s) {
  for (int64_t b = 0; b < num_blocks; ++b) {
    if (block_rows[b] < 0 || (block_rows[b] > 0 && (block_cols[b] == nullptr || block_out_bitmaps[b] == nullptr))) return QSX_ERR_INVALID_ARGUMENT;
  }
  if (out_counts_dev != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_counts_dev, 0, sizeof(int64_t) * static_cast<size_t>(num_blocks), s));
  std::vector<long long> table;
  const long long tiles = build_run_table(kTileWords * 64, num_blocks, block_rows, block_cols, reinterpret_cast<const void *const *>(block_filters),
                                          reinterpret_cast<void *const *>(block_out_bitmaps), nullptr, &table);
  if (tiles < 0) return QSX_ERR_INVALID_ARGUMENT;
  if (tiles == 0) return QSX_OK;
  const long long extra = kCodes ? static_cast<long long>(table.size()) : 0;
  table.insert(table.end(), extra_words.begin(), extra_words.end());
  const size_t bytes = table.size() * sizeof(long long);
  const long long *runs_dev = static_cast<const long long *>(staged_device_buffer(s, bytes));
  if (runs_dev == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  int rc = staged_upload(s, table.data(), bytes);
  if (rc != QSX_OK) return rc;
  CallScratch scratch(s);
  rc = scratch.reserve(sizeof(SortedBounds) * static_cast<size_t>(num_blocks));
  if (rc != QSX_OK) return rc;
  SortedBounds *bounds = static_cast<SortedBounds *>(scratch.take(sizeof(SortedBounds) * static_cast<size_t>(num_blocks)));
  hipLaunchKernelGGL((sorted_bounds_runs_kernel<T, kCodes>), dim3(grid_for(num_blocks, kWavesPerBlock)), dim3(kBlock), 0, s, runs_dev, literal,
                     extra, bounds);
  QSX_CHECK_LAUNCH();
  hipLaunchKernelGGL(sorted_range_bitmap_runs_kernel, dim3(grid_for(tiles, kWavesPerBlock)), dim3(kBlock), 0, s, runs_dev, bounds, op, extra,
                     reinterpret_cast<unsigned long long *>(out_counts_dev));
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

extern "C" {

int qsx_select_cmp_char(const void *col_dev, int width, int64_t n, int op, const void *literal, int literal_length,
                        const uint64_t *filter_dev, uint64_t *out_bitmap_dev, int64_t *out_count_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || width < 1 || width > 255 || literal_length < 0 || (literal_length > 0 && literal == nullptr) || op < QSX_EQ ||
      op > QSX_GE || (n > 0 && (col_dev == nullptr || out_bitmap_dev == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (literal_length > QSX_MAX_CHAR_LITERAL) return QSX_ERR_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
  if (out_count_dev != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), s));
  if (n == 0) return QSX_OK;
  CharLiteral lit{};
  lit.length = 0;
  while (lit.length < literal_length && static_cast<const unsigned char *>(literal)[lit.length] != 0) {   // a NUL ends it
    lit.bytes[lit.length] = static_cast<const unsigned char *>(literal)[lit.length];
    ++lit.length;
  }
  // rows per tile: a multiple of 64 (whole bitmap words), at most 48 KiB of LDS
  int tile_rows = (48 * 1024 / width) / 64 * 64;
  if (tile_rows > 1024) tile_rows = 1024;
  if (tile_rows < 64) tile_rows = 64;
  // short fields in a 4-byte aligned stripe are read straight from it (select_char_tile): no LDS, 16 bitmap words per wave and tile
  const bool direct = width <= 16 && lit.length <= 16 && (reinterpret_cast<uintptr_t>(col_dev) & 3) == 0;
  if (direct) tile_rows = 4096;
  const int64_t tiles = (n + tile_rows - 1) / tile_rows;
  const int grid = static_cast<int>(tiles < 8 * kCUs ? tiles : 8 * kCUs);
  const size_t lds = direct ? kCharDirectLds : (static_cast<size_t>(tile_rows) * width + 15) / 16 * 16;
  hipLaunchKernelGGL(select_char_kernel, dim3(grid), dim3(kBlock), lds, s, static_cast<const unsigned char *>(col_dev), width, n, op,
                     lit, filter_dev, out_bitmap_dev, reinterpret_cast<unsigned long long *>(out_count_dev), tile_rows);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_select_cmp_char_blocks(int width, int64_t num_blocks, const int64_t *block_rows, const void *const *block_cols, int op,
                               const void *literal, int literal_length, const uint64_t *const *block_filters,
                               uint64_t *const *block_out_bitmaps, int64_t *out_counts_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (num_blocks < 0 || width < 1 || width > 255 || literal_length < 0 || (literal_length > 0 && literal == nullptr) || op < QSX_EQ || op > QSX_GE ||
      (num_blocks > 0 && (block_rows == nullptr || block_cols == nullptr || block_out_bitmaps == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (literal_length > QSX_MAX_CHAR_LITERAL) return QSX_ERR_UNSUPPORTED;
  if (num_blocks == 0) return QSX_OK;
  for (int64_t b = 0; b < num_blocks; ++b) {
    if (block_rows[b] < 0 || (block_rows[b] > 0 && (block_cols[b] == nullptr || block_out_bitmaps[b] == nullptr))) return QSX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = as_stream(stream);
  if (out_counts_dev != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_counts_dev, 0, sizeof(int64_t) * static_cast<size_t>(num_blocks), s));
  CharLiteral lit{};
  lit.length = 0;
  while (lit.length < literal_length && static_cast<const unsigned char *>(literal)[lit.length] != 0) {   // a NUL ends it
    lit.bytes[lit.length] = static_cast<const unsigned char *>(literal)[lit.length];
    ++lit.length;
  }
  int tile_rows = (48 * 1024 / width) / 64 * 64;   // as qsx_select_cmp_char
  if (tile_rows > 1024) tile_rows = 1024;
  if (tile_rows < 64) tile_rows = 64;
  bool direct = width <= 16 && lit.length <= 16;
  for (int64_t b = 0; b < num_blocks && direct; ++b) direct = (reinterpret_cast<uintptr_t>(block_cols[b]) & 3) == 0;
  if (direct) tile_rows = 4096;
  std::vector<long long> table;
  const long long tiles = build_run_table(tile_rows, num_blocks, block_rows, block_cols, reinterpret_cast<const void *const *>(block_filters),
                                          reinterpret_cast<void *const *>(block_out_bitmaps), nullptr, &table);
  if (tiles < 0) return QSX_ERR_INVALID_ARGUMENT;
  if (tiles == 0) return QSX_OK;
  const size_t bytes = table.size() * sizeof(long long);
  const long long *runs_dev = static_cast<const long long *>(staged_device_buffer(s, bytes));
  if (runs_dev == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  const int rc = staged_upload(s, table.data(), bytes);
  if (rc != QSX_OK) return rc;
  const int grid = static_cast<int>(tiles < 8 * kCUs ? tiles : 8 * kCUs);
  const size_t lds = direct ? kCharDirectLds : (static_cast<size_t>(tile_rows) * width + 15) / 16 * 16;
  hipLaunchKernelGGL(select_char_runs_kernel, dim3(grid), dim3(kBlock), lds, s, runs_dev, width, op, lit,
                     reinterpret_cast<unsigned long long *>(out_counts_dev), tile_rows);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_select_cmp(int type, const void *col_dev, int64_t n, int op, const void *literal,
                   const uint64_t *filter_dev, uint64_t *out_bitmap_dev, int64_t *out_count_dev,
                   qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || literal == nullptr || (n > 0 && (col_dev == nullptr || out_bitmap_dev == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = as_stream(stream);
  if (out_count_dev != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), s));
  if (n == 0) return QSX_OK;
  switch (type) {
    case QSX_INT: return dispatch_select_op<int32_t>(op, col_dev, nullptr, n, literal, filter_dev, out_bitmap_dev, out_count_dev, s);
    case QSX_LONG: return dispatch_select_op<int64_t>(op, col_dev, nullptr, n, literal, filter_dev, out_bitmap_dev, out_count_dev, s);
    case QSX_FLOAT: return dispatch_select_op<float>(op, col_dev, nullptr, n, literal, filter_dev, out_bitmap_dev, out_count_dev, s);
    case QSX_DOUBLE: return dispatch_select_op<double>(op, col_dev, nullptr, n, literal, filter_dev, out_bitmap_dev, out_count_dev, s);
    case QSX_DATE: return dispatch_select_op<DateValue>(op, col_dev, nullptr, n, literal, filter_dev, out_bitmap_dev, out_count_dev, s);
    default: return QSX_ERR_UNSUPPORTED;
  }
}

int qsx_select_cmp_blocks(int type, int64_t num_blocks, const int64_t *block_rows, const void *const *block_cols, int op,
                          const void *literal, const uint64_t *const *block_filters, uint64_t *const *block_out_bitmaps,
                          int64_t *out_counts_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  const int width = type_width(type);
  if (width == 0) return QSX_ERR_UNSUPPORTED;
  if (num_blocks < 0 || literal == nullptr || op < QSX_EQ || op > QSX_GE ||
      (num_blocks > 0 && (block_rows == nullptr || block_cols == nullptr || block_out_bitmaps == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  bool aligned = true;
  for (int64_t b = 0; b < num_blocks; ++b) {
    if (block_rows[b] < 0 || (block_rows[b] > 0 && (block_cols[b] == nullptr || block_out_bitmaps[b] == nullptr))) {
      return QSX_ERR_INVALID_ARGUMENT;
    }
    aligned = aligned && aligned16(block_cols[b]);
  }
  hipStream_t s = as_stream(stream);
  if (num_blocks == 0) return QSX_OK;
  if (!aligned) {
    // a stripe that does not start on a 16-byte boundary takes the row-per-lane kernel: block by block
    for (int64_t b = 0; b < num_blocks; ++b) {
      const int rc = qsx_select_cmp(type, block_cols[b], block_rows[b], op, literal, block_filters != nullptr ? block_filters[b] : nullptr,
                                    block_out_bitmaps[b], out_counts_dev != nullptr ? out_counts_dev + b : nullptr, stream);
      if (rc != QSX_OK) return rc;
    }
    return QSX_OK;
  }
  if (out_counts_dev != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_counts_dev, 0, sizeof(int64_t) * static_cast<size_t>(num_blocks), s));
  constexpr int kLoadsPerTile = 4;                                   // R of select_packed_runs_kernel
  const long long tile_rows = static_cast<long long>(kWave) * (16 / width) * kLoadsPerTile;
  std::vector<long long> table;
  const long long tiles = build_run_table(tile_rows, num_blocks, block_rows, block_cols,
                                          reinterpret_cast<const void *const *>(block_filters),
                                          reinterpret_cast<void *const *>(block_out_bitmaps), nullptr, &table);
  if (tiles < 0) return QSX_ERR_INVALID_ARGUMENT;
  if (tiles == 0) return QSX_OK;
  const size_t bytes = table.size() * sizeof(long long);
  const long long *runs_dev = static_cast<const long long *>(staged_device_buffer(s, bytes));
  if (runs_dev == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  const int rc = staged_upload(s, table.data(), bytes);
  if (rc != QSX_OK) return rc;
  switch (type) {
    case QSX_INT: return dispatch_select_runs<int32_t>(op, runs_dev, tiles, literal, out_counts_dev, s);
    case QSX_LONG: return dispatch_select_runs<int64_t>(op, runs_dev, tiles, literal, out_counts_dev, s);
    case QSX_FLOAT: return dispatch_select_runs<float>(op, runs_dev, tiles, literal, out_counts_dev, s);
    case QSX_DOUBLE: return dispatch_select_runs<double>(op, runs_dev, tiles, literal, out_counts_dev, s);
    case QSX_DATE: return dispatch_select_runs<DateValue>(op, runs_dev, tiles, literal, out_counts_dev, s);
    default: return QSX_ERR_UNSUPPORTED;
  }
}

int qsx_select_cmp_sorted(int type, const void *col_dev, int64_t n, int op, const void *literal, const uint64_t *filter_dev,
                          uint64_t *out_bitmap_dev, int64_t *out_count_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || literal == nullptr || op < QSX_EQ || op > QSX_GE || (n > 0 && (col_dev == nullptr || out_bitmap_dev == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = as_stream(stream);
  if (out_count_dev != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), s));
  if (n == 0) return QSX_OK;
  SortedBounds *bounds = device_slot<SortedBounds>(s);
  if (bounds == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  switch (type) {
    case QSX_INT:
      hipLaunchKernelGGL(sorted_bounds_kernel<int32_t>, dim3(1), dim3(kWave), 0, s, static_cast<const int32_t *>(col_dev), n,
                         *static_cast<const int32_t *>(literal), bounds);
      break;
    case QSX_LONG:
      hipLaunchKernelGGL(sorted_bounds_kernel<int64_t>, dim3(1), dim3(kWave), 0, s, static_cast<const int64_t *>(col_dev), n,
                         *static_cast<const int64_t *>(literal), bounds);
      break;
    case QSX_FLOAT:
      hipLaunchKernelGGL(sorted_bounds_kernel<float>, dim3(1), dim3(kWave), 0, s, static_cast<const float *>(col_dev), n,
                         *static_cast<const float *>(literal), bounds);
      break;
    case QSX_DOUBLE:
      hipLaunchKernelGGL(sorted_bounds_kernel<double>, dim3(1), dim3(kWave), 0, s, static_cast<const double *>(col_dev), n,
                         *static_cast<const double *>(literal), bounds);
      break;
    case QSX_DATE:
      hipLaunchKernelGGL(sorted_bounds_kernel<DateValue>, dim3(1), dim3(kWave), 0, s, static_cast<const DateValue *>(col_dev), n,
                         *static_cast<const DateValue *>(literal), bounds);
      break;
    default: return QSX_ERR_UNSUPPORTED;
  }
  QSX_CHECK_LAUNCH();
  hipLaunchKernelGGL(sorted_range_bitmap_kernel, dim3(grid_for((n + 63) >> 6, kBlock * 8) < 2 * kCUs ? grid_for((n + 63) >> 6, kBlock * 8) : 2 * kCUs), dim3(kBlock), 0, s, bounds, n, op, filter_dev,
                     out_bitmap_dev, reinterpret_cast<unsigned long long *>(out_count_dev));
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_select_cmp_columns(int type, const void *lhs_dev, const void *rhs_dev, int64_t n, int op,
                           const uint64_t *filter_dev, uint64_t *out_bitmap_dev, int64_t *out_count_dev,
                           qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || (n > 0 && (lhs_dev == nullptr || rhs_dev == nullptr || out_bitmap_dev == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = as_stream(stream);
  if (out_count_dev != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), s));
  if (n == 0) return QSX_OK;
  switch (type) {
    case QSX_INT: return dispatch_select_op<int32_t>(op, lhs_dev, rhs_dev, n, nullptr, filter_dev, out_bitmap_dev, out_count_dev, s);
    case QSX_LONG: return dispatch_select_op<int64_t>(op, lhs_dev, rhs_dev, n, nullptr, filter_dev, out_bitmap_dev, out_count_dev, s);
    case QSX_FLOAT: return dispatch_select_op<float>(op, lhs_dev, rhs_dev, n, nullptr, filter_dev, out_bitmap_dev, out_count_dev, s);
    case QSX_DOUBLE: return dispatch_select_op<double>(op, lhs_dev, rhs_dev, n, nullptr, filter_dev, out_bitmap_dev, out_count_dev, s);
    case QSX_DATE: return dispatch_select_op<DateValue>(op, lhs_dev, rhs_dev, n, nullptr, filter_dev, out_bitmap_dev, out_count_dev, s);
    default: return QSX_ERR_UNSUPPORTED;
  }
}

int qsx_select_codes(int code_width, const void *codes_dev, int64_t n, int op, uint32_t first, uint32_t second,
                     const uint64_t *filter_dev, uint64_t *out_bitmap_dev, int64_t *out_count_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || op < QSX_CODE_EQ || op > QSX_CODE_RANGE || (n > 0 && (codes_dev == nullptr || out_bitmap_dev == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = as_stream(stream);
  if (out_count_dev != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), s));
  if (n == 0) return QSX_OK;
  if (aligned16(codes_dev)) {
    CodeRangePred pred;
    pred.lo = op == QSX_CODE_LT ? 0ull : first;
    pred.hi = (op == QSX_CODE_EQ || op == QSX_CODE_NE) ? static_cast<unsigned long long>(first) + 1
              : op == QSX_CODE_LT ? first : op == QSX_CODE_GE ? (1ull << 32) : second;
    pred.negate = op == QSX_CODE_NE;
    switch (code_width) {
      case 1: return launch_select_packed<uint8_t>(codes_dev, n, pred, filter_dev, out_bitmap_dev, out_count_dev, s);
      case 2: return launch_select_packed<uint16_t>(codes_dev, n, pred, filter_dev, out_bitmap_dev, out_count_dev, s);
      case 4: return launch_select_packed<uint32_t>(codes_dev, n, pred, filter_dev, out_bitmap_dev, out_count_dev, s);
      default: return QSX_ERR_UNSUPPORTED;
    }
  }
  const int64_t num_words = (n + 63) >> 6;
  constexpr int R = 8;
  const int grid = grid_for(num_words, kWavesPerBlock * R);
  unsigned long long *count = reinterpret_cast<unsigned long long *>(out_count_dev);
  switch (code_width) {
    case 1:
      hipLaunchKernelGGL((select_codes_kernel<uint8_t, R>), dim3(grid), dim3(kBlock), 0, s, static_cast<const uint8_t *>(codes_dev),
                         n, op, first, second, filter_dev, out_bitmap_dev, count);
      break;
    case 2:
      hipLaunchKernelGGL((select_codes_kernel<uint16_t, R>), dim3(grid), dim3(kBlock), 0, s, static_cast<const uint16_t *>(codes_dev),
                         n, op, first, second, filter_dev, out_bitmap_dev, count);
      break;
    case 4:
      hipLaunchKernelGGL((select_codes_kernel<uint32_t, R>), dim3(grid), dim3(kBlock), 0, s, static_cast<const uint32_t *>(codes_dev),
                         n, op, first, second, filter_dev, out_bitmap_dev, count);
      break;
    default: return QSX_ERR_UNSUPPORTED;
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_select_codes_blocks(int code_width, int64_t num_blocks, const int64_t *block_rows, const void *const *block_codes,
                            const int32_t *block_ops, const uint32_t *block_first, const uint32_t *block_second,
                            const uint64_t *const *block_filters, uint64_t *const *block_out_bitmaps, int64_t *out_counts_dev,
                            qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (num_blocks < 0 || (num_blocks > 0 && (block_rows == nullptr || block_codes == nullptr || block_ops == nullptr || block_first == nullptr ||
                                            block_second == nullptr || block_out_bitmaps == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (code_width != 1 && code_width != 2 && code_width != 4) return QSX_ERR_UNSUPPORTED;
  if (num_blocks == 0) return QSX_OK;
  bool aligned = true;
  for (int64_t b = 0; b < num_blocks; ++b) {
    if (block_rows[b] < 0 || (block_rows[b] > 0 && (block_codes[b] == nullptr || block_out_bitmaps[b] == nullptr)) ||
        block_ops[b] < QSX_CODE_EQ || block_ops[b] > QSX_CODE_RANGE) {
      return QSX_ERR_INVALID_ARGUMENT;
    }
    aligned = aligned && aligned16(block_codes[b]);
  }
  if (!aligned) {   // a stripe that does not start on a 16-byte boundary takes the row-per-lane kernel: block by block
    for (int64_t b = 0; b < num_blocks; ++b) {
      const int rc = qsx_select_codes(code_width, block_codes[b], block_rows[b], block_ops[b], block_first[b], block_second[b],
                                      block_filters != nullptr ? block_filters[b] : nullptr, block_out_bitmaps[b],
                                      out_counts_dev != nullptr ? out_counts_dev + b : nullptr, stream);
      if (rc != QSX_OK) return rc;
    }
    return QSX_OK;
  }
  hipStream_t s = as_stream(stream);
  if (out_counts_dev != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_counts_dev, 0, sizeof(int64_t) * static_cast<size_t>(num_blocks), s));
  constexpr int kLoadsPerTile = 4;                                   // R of select_packed_runs_kernel
  const long long tile_rows = static_cast<long long>(kWave) * (16 / code_width) * kLoadsPerTile;
  std::vector<long long> table;
  const long long tiles = build_run_table(tile_rows, num_blocks, block_rows, block_codes, reinterpret_cast<const void *const *>(block_filters),
                                          reinterpret_cast<void *const *>(block_out_bitmaps), nullptr, &table);
  if (tiles < 0) return QSX_ERR_INVALID_ARGUMENT;
  if (tiles == 0) return QSX_OK;
  const long long extra = static_cast<long long>(table.size());
  for (int64_t b = 0; b < num_blocks; ++b) {   // the comparison as a code range [lo, hi), != as its complement (as qsx_select_codes)
    const int op = block_ops[b];
    const unsigned long long first = block_first[b], second = block_second[b];
    table.push_back(static_cast<long long>(op == QSX_CODE_LT ? 0ull : first));
    table.push_back(static_cast<long long>((op == QSX_CODE_EQ || op == QSX_CODE_NE) ? first + 1 : op == QSX_CODE_LT ? first
                                           : op == QSX_CODE_GE ? (1ull << 32) : second));
    table.push_back(op == QSX_CODE_NE ? 1 : 0);
  }
  const size_t bytes = table.size() * sizeof(long long);
  const long long *runs_dev = static_cast<const long long *>(staged_device_buffer(s, bytes));
  if (runs_dev == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  const int rc = staged_upload(s, table.data(), bytes);
  if (rc != QSX_OK) return rc;
  const int grid = grid_for(tiles, kWavesPerBlock);
  unsigned long long *counts = reinterpret_cast<unsigned long long *>(out_counts_dev);
  switch (code_width) {
    case 1: hipLaunchKernelGGL((select_packed_runs_kernel<uint8_t, CodeRangePred, kLoadsPerTile>), dim3(grid), dim3(kBlock), 0, s, runs_dev, CodeRangePred{}, counts, extra); break;
    case 2: hipLaunchKernelGGL((select_packed_runs_kernel<uint16_t, CodeRangePred, kLoadsPerTile>), dim3(grid), dim3(kBlock), 0, s, runs_dev, CodeRangePred{}, counts, extra); break;
    default: hipLaunchKernelGGL((select_packed_runs_kernel<uint32_t, CodeRangePred, kLoadsPerTile>), dim3(grid), dim3(kBlock), 0, s, runs_dev, CodeRangePred{}, counts, extra); break;
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_select_codes_sorted(int code_width, const void *codes_dev, int64_t n, int op, uint32_t first, uint32_t second,
                            const uint64_t *filter_dev, uint64_t *out_bitmap_dev, int64_t *out_count_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || op < QSX_CODE_EQ || op > QSX_CODE_RANGE || (n > 0 && (codes_dev == nullptr || out_bitmap_dev == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (code_width != 1 && code_width != 2 && code_width != 4) return QSX_ERR_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
  if (out_count_dev != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), s));
  if (n == 0) return QSX_OK;
  // the comparison as a code range [lo, hi), != as its complement (as qsx_select_codes)
  const unsigned long long lo = op == QSX_CODE_LT ? 0ull : first;
  const unsigned long long hi = (op == QSX_CODE_EQ || op == QSX_CODE_NE) ? static_cast<unsigned long long>(first) + 1
                                : op == QSX_CODE_LT ? first : op == QSX_CODE_GE ? (1ull << 32) : second;
  SortedBounds *bounds = device_slot<SortedBounds>(s);
  if (bounds == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  switch (code_width) {
    case 1:
      hipLaunchKernelGGL(sorted_code_bounds_kernel<uint8_t>, dim3(1), dim3(kWave), 0, s, static_cast<const uint8_t *>(codes_dev), n, lo, hi, bounds);
      break;
    case 2:
      hipLaunchKernelGGL(sorted_code_bounds_kernel<uint16_t>, dim3(1), dim3(kWave), 0, s, static_cast<const uint16_t *>(codes_dev), n, lo, hi, bounds);
      break;
    default:
      hipLaunchKernelGGL(sorted_code_bounds_kernel<uint32_t>, dim3(1), dim3(kWave), 0, s, static_cast<const uint32_t *>(codes_dev), n, lo, hi, bounds);
      break;
  }
  QSX_CHECK_LAUNCH();
  const int grid = grid_for((n + 63) >> 6, kBlock * 8) < 2 * kCUs ? grid_for((n + 63) >> 6, kBlock * 8) : 2 * kCUs;
  hipLaunchKernelGGL(sorted_range_bitmap_kernel, dim3(grid), dim3(kBlock), 0, s, bounds, n, op == QSX_CODE_NE ? QSX_NE : QSX_EQ, filter_dev,
                     out_bitmap_dev, reinterpret_cast<unsigned long long *>(out_count_dev));
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_select_cmp_sorted_blocks(int type, int64_t num_blocks, const int64_t *block_rows, const void *const *block_cols, int op,
                                 const void *literal, const uint64_t *const *block_filters, uint64_t *const *block_out_bitmaps,
                                 int64_t *out_counts_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (num_blocks < 0 || literal == nullptr || op < QSX_EQ || op > QSX_GE ||
      (num_blocks > 0 && (block_rows == nullptr || block_cols == nullptr || block_out_bitmaps == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (num_blocks == 0) return QSX_OK;
  hipStream_t s = as_stream(stream);
  const std::vector<long long> none;
  switch (type) {
    case QSX_INT: return sorted_runs<int32_t, false>(num_blocks, block_rows, block_cols, *static_cast<const int32_t *>(literal), op, none, block_filters, block_out_bitmaps, out_counts_dev, s);
    case QSX_LONG: return sorted_runs<int64_t, false>(num_blocks, block_rows, block_cols, *static_cast<const int64_t *>(literal), op, none, block_filters, block_out_bitmaps, out_counts_dev, s);
    case QSX_FLOAT: return sorted_runs<float, false>(num_blocks, block_rows, block_cols, *static_cast<const float *>(literal), op, none, block_filters, block_out_bitmaps, out_counts_dev, s);
    case QSX_DOUBLE: return sorted_runs<double, false>(num_blocks, block_rows, block_cols, *static_cast<const double *>(literal), op, none, block_filters, block_out_bitmaps, out_counts_dev, s);
    case QSX_DATE: {
      DateValue d;
      std::memcpy(&d, literal, sizeof(d));
      return sorted_runs<DateValue, false>(num_blocks, block_rows, block_cols, d, op, none, block_filters, block_out_bitmaps, out_counts_dev, s);
    }
    default: return QSX_ERR_UNSUPPORTED;
  }
}

int qsx_select_codes_sorted_blocks(int code_width, int64_t num_blocks, const int64_t *block_rows, const void *const *block_codes,
                                   const int32_t *block_ops, const uint32_t *block_first, const uint32_t *block_second,
                                   const uint64_t *const *block_filters, uint64_t *const *block_out_bitmaps, int64_t *out_counts_dev,
                                   qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (num_blocks < 0 || (num_blocks > 0 && (block_rows == nullptr || block_codes == nullptr || block_ops == nullptr || block_first == nullptr ||
                                            block_second == nullptr || block_out_bitmaps == nullptr))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (code_width != 1 && code_width != 2 && code_width != 4) return QSX_ERR_UNSUPPORTED;
  if (num_blocks == 0) return QSX_OK;
  std::vector<long long> extra(static_cast<size_t>(num_blocks) * 3);
  for (int64_t b = 0; b < num_blocks; ++b) {
    const int op = block_ops[b];
    if (op < QSX_CODE_EQ || op > QSX_CODE_RANGE) return QSX_ERR_INVALID_ARGUMENT;
    // the comparison as a code range [lo, hi), != as its complement (as qsx_select_codes_sorted)
    const unsigned long long first = block_first[b], second = block_second[b];
    extra[3 * b] = static_cast<long long>(op == QSX_CODE_LT ? 0ull : first);
    extra[3 * b + 1] = static_cast<long long>((op == QSX_CODE_EQ || op == QSX_CODE_NE) ? first + 1 : op == QSX_CODE_LT ? first
                                              : op == QSX_CODE_GE ? (1ull << 32) : second);
    extra[3 * b + 2] = op;
  }
  hipStream_t s = as_stream(stream);
  switch (code_width) {
    case 1: return sorted_runs<uint8_t, true>(num_blocks, block_rows, block_codes, 0, QSX_EQ, extra, block_filters, block_out_bitmaps, out_counts_dev, s);
    case 2: return sorted_runs<uint16_t, true>(num_blocks, block_rows, block_codes, 0, QSX_EQ, extra, block_filters, block_out_bitmaps, out_counts_dev, s);
    default: return sorted_runs<uint32_t, true>(num_blocks, block_rows, block_codes, 0, QSX_EQ, extra, block_filters, block_out_bitmaps, out_counts_dev, s);
  }
}

int qsx_decode_codes(int code_width, const void *codes_dev, int64_t n, const void *dictionary_dev, int value_width,
                     void *out_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || (n > 0 && (codes_dev == nullptr || out_dev == nullptr))) return QSX_ERR_INVALID_ARGUMENT;
  if (n == 0) return QSX_OK;
  hipStream_t s = as_stream(stream);
  switch (code_width) {
    case 1: return launch_decode<uint8_t>(codes_dev, n, dictionary_dev, value_width, out_dev, s);
    case 2: return launch_decode<uint16_t>(codes_dev, n, dictionary_dev, value_width, out_dev, s);
    case 4: return launch_decode<uint32_t>(codes_dev, n, dictionary_dev, value_width, out_dev, s);
    default: return QSX_ERR_UNSUPPORTED;
  }
}

int qsx_tids_to_bitmap(const int32_t *tids_dev, int64_t n, int32_t base_tid, int64_t num_bits,
                       uint64_t *out_bitmap_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || num_bits < 0 || (num_bits > 0 && out_bitmap_dev == nullptr) || (n > 0 && tids_dev == nullptr)) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (num_bits == 0) return QSX_OK;
  hipStream_t s = as_stream(stream);
  QSX_HIP_TRY(hipMemsetAsync(out_bitmap_dev, 0, static_cast<size_t>((num_bits + 63) >> 6) * 8, s));
  if (n == 0) return QSX_OK;
  hipLaunchKernelGGL(tids_to_bitmap_kernel, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, s, tids_dev, n, base_tid,
                     num_bits, reinterpret_cast<unsigned long long *>(out_bitmap_dev));
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_bitmap_combine(int op, const uint64_t *a_dev, const uint64_t *b_dev, int64_t n,
                       uint64_t *out_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || op < 0 || op > 3) return QSX_ERR_INVALID_ARGUMENT;
  if (n == 0) return QSX_OK;
  if (a_dev == nullptr || out_dev == nullptr || (op != 3 && b_dev == nullptr)) return QSX_ERR_INVALID_ARGUMENT;
  const int64_t num_words = (n + 63) >> 6;
  hipLaunchKernelGGL(bitmap_combine_kernel, dim3(grid_for(num_words, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), op, a_dev, b_dev, n, out_dev);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_bitmap_count(const uint64_t *bitmap_dev, int64_t n, int64_t *out_count_dev,
                     qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || out_count_dev == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  hipStream_t s = as_stream(stream);
  QSX_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), s));
  if (n == 0) return QSX_OK;
  const int64_t num_words = (n + 63) >> 6;
  hipLaunchKernelGGL(bitmap_count_kernel, dim3(grid_for(num_words, kBlock * 4)), dim3(kBlock), 0, s,
                     bitmap_dev, num_words, reinterpret_cast<unsigned long long *>(out_count_dev));
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

size_t qsx_compact_workspace_bytes(int64_t n) {
  const int64_t num_words = (n + 63) >> 6;
  const int64_t num_tiles = (num_words + kSmallTileWords - 1) / kSmallTileWords;   // (room for the small tiles of run_compaction)
  return align_up(sizeof(int64_t) * (num_tiles + 1), 256) + align_up(sizeof(int32_t) * (num_tiles + 1), 256);
}

int qsx_compact_gather(int ncols, const void *const *cols, const int32_t *widths,
                       const uint64_t *bitmap_dev, int64_t n, void *const *out_cols,
                       int64_t *out_count_dev, void *workspace_dev, size_t workspace_bytes,
                       qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || ncols < 0 || ncols > QSX_MAX_COLUMNS) return QSX_ERR_INVALID_ARGUMENT;
  GatherArgs args;
  args.ncols = ncols;
  for (int c = 0; c < ncols; ++c) {
    const int w = widths[c];
    if (w < 1 || w > 4096) return QSX_ERR_UNSUPPORTED;   // 1 / 2 / 4 / 8: one load and store per value; other widths (CHAR(n)) byte by byte
    args.width[c] = w;
    args.src[c] = cols[c];
    args.dst[c] = out_cols[c];
  }
  return run_compaction(args, bitmap_dev, n, nullptr, 0, out_count_dev, workspace_dev,
                        workspace_bytes, as_stream(stream));
}

size_t qsx_compact_blocks_workspace_bytes(int64_t num_blocks, const int64_t *block_rows) {
  int64_t tiles = 0;
  for (int64_t b = 0; b < num_blocks; ++b) tiles += (block_rows[b] + kTileWords * 64 - 1) / (kTileWords * 64);
  return align_up(sizeof(int64_t) * (tiles + 1), 256) + align_up(sizeof(int32_t) * (tiles + 1), 256);
}

int qsx_compact_gather_blocks(int ncols, const int32_t *widths, int64_t num_blocks, const int64_t *block_rows,
                              const void *const *block_cols, const uint64_t *const *block_bitmaps,
                              const int32_t *block_base_tids, void *const *out_cols, int32_t *out_tids_dev,
                              int64_t *out_count_dev, void *workspace_dev, size_t workspace_bytes, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (ncols < 0 || ncols > QSX_MAX_COLUMNS || num_blocks < 0 || (ncols > 0 && (widths == nullptr || out_cols == nullptr)) ||
      (num_blocks > 0 && (block_rows == nullptr || block_bitmaps == nullptr || (ncols > 0 && block_cols == nullptr)))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  GatherArgs args;
  args.ncols = ncols;
  for (int c = 0; c < ncols; ++c) {
    const int w = widths[c];
    if (w < 1 || w > 4096) return QSX_ERR_UNSUPPORTED;   // 1 / 2 / 4 / 8: one load and store per value; other widths (CHAR(n)) byte by byte
    args.width[c] = w;
    args.src[c] = nullptr;
    args.dst[c] = out_cols[c];
  }
  hipStream_t s = as_stream(stream);
  std::vector<int64_t> base(static_cast<size_t>(num_blocks));
  int64_t total = 0;
  for (int64_t b = 0; b < num_blocks; ++b) {
    if (block_rows[b] < 0 || (block_rows[b] > 0 && block_bitmaps[b] == nullptr)) return QSX_ERR_INVALID_ARGUMENT;
    base[b] = block_base_tids != nullptr ? block_base_tids[b] : total;
    if (out_tids_dev != nullptr && (base[b] < 0 || base[b] + block_rows[b] > INT32_MAX)) return QSX_ERR_INVALID_ARGUMENT;
    total += block_rows[b];
  }
  if (workspace_bytes < qsx_compact_blocks_workspace_bytes(num_blocks, block_rows)) return QSX_ERR_CAPACITY;
  std::vector<long long> table;
  std::vector<const void *> no_input(static_cast<size_t>(num_blocks), nullptr);
  const long long tiles = build_run_table(kTileWords * 64, num_blocks, block_rows, no_input.data(),
                                          reinterpret_cast<const void *const *>(block_bitmaps), nullptr, base.data(), &table);
  if (tiles < 0) return QSX_ERR_INVALID_ARGUMENT;
  if (tiles == 0) {
    if (out_count_dev != nullptr) QSX_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), s));
    return QSX_OK;
  }
  const long long cols_offset = static_cast<long long>(table.size());
  for (int64_t b = 0; b < num_blocks; ++b) {
    for (int c = 0; c < ncols; ++c) {
      const void *col = block_cols[b * ncols + c];
      if (block_rows[b] > 0 && col == nullptr) return QSX_ERR_INVALID_ARGUMENT;
      table.push_back(static_cast<long long>(reinterpret_cast<uintptr_t>(col)));
    }
  }
  const size_t bytes = table.size() * sizeof(long long);
  const long long *runs_dev = static_cast<const long long *>(staged_device_buffer(s, bytes));
  if (runs_dev == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  const int rc = staged_upload(s, table.data(), bytes);
  if (rc != QSX_OK) return rc;
  int64_t *tile_offsets = static_cast<int64_t *>(workspace_dev);
  int32_t *tile_counts = reinterpret_cast<int32_t *>(static_cast<char *>(workspace_dev) + align_up(sizeof(int64_t) * (tiles + 1), 256));
  hipLaunchKernelGGL(tile_count_runs_kernel, dim3(grid_for(tiles, kWavesPerBlock)), dim3(kBlock), 0, s, runs_dev, tile_counts);
  QSX_CHECK_LAUNCH();
  hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, tile_counts, static_cast<int64_t>(tiles), tile_offsets, out_count_dev);
  QSX_CHECK_LAUNCH();
  hipLaunchKernelGGL(compact_gather_runs_kernel, dim3(grid_for(tiles, kWavesPerBlock)), dim3(kBlock), 0, s, args, runs_dev,
                     cols_offset, tile_offsets, out_tids_dev);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_bitmap_to_tids(const uint64_t *bitmap_dev, int64_t n, int32_t base_tid,
                       int32_t *out_tids_dev, int64_t *out_count_dev, void *workspace_dev,
                       size_t workspace_bytes, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || (n > 0 && out_tids_dev == nullptr)) return QSX_ERR_INVALID_ARGUMENT;
  GatherArgs args;
  args.ncols = 0;
  return run_compaction(args, bitmap_dev, n, out_tids_dev, base_tid, out_count_dev, workspace_dev,
                        workspace_bytes, as_stream(stream));
}

int qsx_gather(int width, const void *src_dev, const int32_t *tids_dev, int64_t n, void *dst_dev,
               qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0) return QSX_ERR_INVALID_ARGUMENT;
  if (n == 0) return QSX_OK;
  const int grid = grid_for(n, kBlock * 8);
  hipStream_t s = as_stream(stream);
  switch (width) {
    case 1: hipLaunchKernelGGL(gather_kernel<uint8_t>, dim3(grid), dim3(kBlock), 0, s, static_cast<const uint8_t *>(src_dev), tids_dev, n, static_cast<uint8_t *>(dst_dev)); break;
    case 2: hipLaunchKernelGGL(gather_kernel<uint16_t>, dim3(grid), dim3(kBlock), 0, s, static_cast<const uint16_t *>(src_dev), tids_dev, n, static_cast<uint16_t *>(dst_dev)); break;
    case 4: hipLaunchKernelGGL(gather_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, s, static_cast<const uint32_t *>(src_dev), tids_dev, n, static_cast<uint32_t *>(dst_dev)); break;
    case 8: hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(grid), dim3(kBlock), 0, s, static_cast<const uint64_t *>(src_dev), tids_dev, n, static_cast<uint64_t *>(dst_dev)); break;
    default:
      if (width < 1 || width > 4096) return QSX_ERR_UNSUPPORTED;
      hipLaunchKernelGGL(gather_bytes_kernel, dim3(grid_for(n * width, kBlock * 8)), dim3(kBlock), 0, s, static_cast<const uint8_t *>(src_dev), width,
                         tids_dev, n, static_cast<uint8_t *>(dst_dev));
      break;
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

// The segment table of a long run in device memory: [first rows as int32, padded to 8 bytes | addresses].
static int upload_segment_table(int num_segments, const void *const *ptrs, const int64_t *first_row, hipStream_t stream,
                                const int32_t **first_dev, const long long **ptrs_dev) {
  const size_t first_words = (static_cast<size_t>(num_segments) + 1) / 2;     // 64-bit words holding the int32 first rows
  std::vector<long long> table(first_words + static_cast<size_t>(num_segments), 0);
  int32_t *first = reinterpret_cast<int32_t *>(table.data());
  for (int i = 0; i < num_segments; ++i) {
    if (first_row[i] < 0 || first_row[i] > INT32_MAX) return QSX_ERR_INVALID_ARGUMENT;
    first[i] = static_cast<int32_t>(first_row[i]);
    table[first_words + i] = static_cast<long long>(reinterpret_cast<uintptr_t>(ptrs[i]));
  }
  const size_t bytes = table.size() * sizeof(long long);
  const long long *dev = static_cast<const long long *>(staged_device_buffer(stream, bytes));
  if (dev == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  const int rc = staged_upload(stream, table.data(), bytes);
  if (rc != QSX_OK) return rc;
  *first_dev = reinterpret_cast<const int32_t *>(dev);
  *ptrs_dev = dev + first_words;
  return QSX_OK;
}

int qsx_gather_segmented(int width, int num_segments, const void *const *segment_ptrs,
                         const int64_t *segment_first_row, const int32_t *tids_dev, int64_t n, void *dst_dev,
                         qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || num_segments < 1 || segment_ptrs == nullptr || segment_first_row == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  if (num_segments > kMaxTableSegments) return QSX_ERR_UNSUPPORTED;
  if (n == 0) return QSX_OK;
  // equal-length segments (every block of a relation but the last is full): the segment of a tuple id is a division
  bool uniform = num_segments > 1 && segment_first_row[0] == 0 && segment_first_row[1] > 0 && segment_first_row[1] <= INT32_MAX &&
                 (width == 1 || width == 2 || width == 4 || width == 8);
  for (int i = 2; uniform && i < num_segments; ++i) uniform = segment_first_row[i] == segment_first_row[1] * i;
  if (uniform) {
    hipStream_t s = as_stream(stream);
    const int32_t *first_dev = nullptr;
    const long long *ptrs_dev = nullptr;
    const int rc = upload_segment_table(num_segments, segment_ptrs, segment_first_row, s, &first_dev, &ptrs_dev);
    if (rc != QSX_OK) return rc;
    const int32_t rows = static_cast<int32_t>(segment_first_row[1]);
    int shift = -1;
    if ((rows & (rows - 1)) == 0) {
      shift = 0;
      while ((1 << shift) < rows) ++shift;
    }
    const int grid = grid_for(n, kBlock * 8);
    switch (width) {
      case 1: hipLaunchKernelGGL(gather_uniform_segments_kernel<uint8_t>, dim3(grid), dim3(kBlock), 0, s, ptrs_dev, num_segments, rows, shift, tids_dev, n, static_cast<uint8_t *>(dst_dev)); break;
      case 2: hipLaunchKernelGGL(gather_uniform_segments_kernel<uint16_t>, dim3(grid), dim3(kBlock), 0, s, ptrs_dev, num_segments, rows, shift, tids_dev, n, static_cast<uint16_t *>(dst_dev)); break;
      case 4: hipLaunchKernelGGL(gather_uniform_segments_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, s, ptrs_dev, num_segments, rows, shift, tids_dev, n, static_cast<uint32_t *>(dst_dev)); break;
      default: hipLaunchKernelGGL(gather_uniform_segments_kernel<uint64_t>, dim3(grid), dim3(kBlock), 0, s, ptrs_dev, num_segments, rows, shift, tids_dev, n, static_cast<uint64_t *>(dst_dev)); break;
    }
    QSX_CHECK_LAUNCH();
    return QSX_OK;
  }
  if (num_segments > kMaxSegments) {
    hipStream_t s = as_stream(stream);
    const int32_t *first_dev = nullptr;
    const long long *ptrs_dev = nullptr;
    const int rc = upload_segment_table(num_segments, segment_ptrs, segment_first_row, s, &first_dev, &ptrs_dev);
    if (rc != QSX_OK) return rc;
    const int grid = grid_for(n, kBlock * 16);      // every workgroup pays the copy of the first rows to LDS
    const size_t lds = static_cast<size_t>(num_segments) * sizeof(int32_t);
    switch (width) {
      case 1: hipLaunchKernelGGL(gather_segmented_table_kernel<uint8_t>, dim3(grid), dim3(kBlock), lds, s, first_dev, ptrs_dev, num_segments, tids_dev, n, static_cast<uint8_t *>(dst_dev)); break;
      case 2: hipLaunchKernelGGL(gather_segmented_table_kernel<uint16_t>, dim3(grid), dim3(kBlock), lds, s, first_dev, ptrs_dev, num_segments, tids_dev, n, static_cast<uint16_t *>(dst_dev)); break;
      case 4: hipLaunchKernelGGL(gather_segmented_table_kernel<uint32_t>, dim3(grid), dim3(kBlock), lds, s, first_dev, ptrs_dev, num_segments, tids_dev, n, static_cast<uint32_t *>(dst_dev)); break;
      case 8: hipLaunchKernelGGL(gather_segmented_table_kernel<uint64_t>, dim3(grid), dim3(kBlock), lds, s, first_dev, ptrs_dev, num_segments, tids_dev, n, static_cast<uint64_t *>(dst_dev)); break;
      default:
        if (width < 1 || width > 4096) return QSX_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(gather_segmented_table_bytes_kernel, dim3(grid_for(n * width, kBlock * 16)), dim3(kBlock), lds, s, first_dev, ptrs_dev,
                           num_segments, width, tids_dev, n, static_cast<uint8_t *>(dst_dev));
        break;
    }
    QSX_CHECK_LAUNCH();
    return QSX_OK;
  }
  SegmentTable seg;
  seg.num = num_segments;
  for (int i = 0; i < num_segments; ++i) {
    seg.ptr[i] = segment_ptrs[i];
    seg.first_row[i] = segment_first_row[i];
  }
  const int grid = grid_for(n, kBlock * 4);
  hipStream_t s = as_stream(stream);
  switch (width) {
    case 1: hipLaunchKernelGGL(gather_segmented_kernel<uint8_t>, dim3(grid), dim3(kBlock), 0, s, seg, tids_dev, n, static_cast<uint8_t *>(dst_dev)); break;
    case 2: hipLaunchKernelGGL(gather_segmented_kernel<uint16_t>, dim3(grid), dim3(kBlock), 0, s, seg, tids_dev, n, static_cast<uint16_t *>(dst_dev)); break;
    case 4: hipLaunchKernelGGL(gather_segmented_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, s, seg, tids_dev, n, static_cast<uint32_t *>(dst_dev)); break;
    case 8: hipLaunchKernelGGL(gather_segmented_kernel<uint64_t>, dim3(grid), dim3(kBlock), 0, s, seg, tids_dev, n, static_cast<uint64_t *>(dst_dev)); break;
    default:
      if (width < 1 || width > 4096) return QSX_ERR_UNSUPPORTED;
      hipLaunchKernelGGL(gather_segmented_bytes_kernel, dim3(grid_for(n * width, kBlock * 4)), dim3(kBlock), 0, s, seg, width, tids_dev, n,
                         static_cast<uint8_t *>(dst_dev));
      break;
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

int qsx_bitmap_gather_segmented(int num_segments, const uint64_t *const *segment_bitmaps, const int64_t *segment_first_row,
                                 const int32_t *tids_dev, int64_t n, uint64_t *out_bitmap_dev, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || num_segments < 1 || segment_bitmaps == nullptr || segment_first_row == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  if (num_segments > kMaxTableSegments) return QSX_ERR_UNSUPPORTED;
  if (n == 0) return QSX_OK;
  if (tids_dev == nullptr || out_bitmap_dev == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  if (num_segments > kMaxSegments) {
    hipStream_t s = as_stream(stream);
    const int32_t *first_dev = nullptr;
    const long long *ptrs_dev = nullptr;
    const int rc = upload_segment_table(num_segments, reinterpret_cast<const void *const *>(segment_bitmaps), segment_first_row, s,
                                        &first_dev, &ptrs_dev);
    if (rc != QSX_OK) return rc;
    hipLaunchKernelGGL(bitmap_gather_segmented_table_kernel, dim3(grid_for(n, kBlock * 16)), dim3(kBlock),
                       static_cast<size_t>(num_segments) * sizeof(int32_t), s, first_dev, ptrs_dev, num_segments, tids_dev, n,
                       out_bitmap_dev);
    QSX_CHECK_LAUNCH();
    return QSX_OK;
  }
  SegmentTable seg;
  seg.num = num_segments;
  for (int i = 0; i < num_segments; ++i) {
    seg.ptr[i] = segment_bitmaps[i];
    seg.first_row[i] = segment_first_row[i];
  }
  hipLaunchKernelGGL(bitmap_gather_segmented_kernel, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, as_stream(stream), seg,
                     tids_dev, n, out_bitmap_dev);
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

}  // extern "C"
