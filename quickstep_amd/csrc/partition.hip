// partition.hip — K9: stable hash-partition scatter (first half of the
// multi-GPU join-key shuffle; also the partitioned-aggregation router).
//
// Partition id = HashPartitionSchemeHeader::getPartitionId
// (catalog/PartitionSchemeHeader.hpp:200-214) over the identity hash of the
// key's zero-extended bit pattern (types/TypedValue.hpp:575-592); the routing
// itself is what PartitionAwareInsertDestination does per output row
// (storage/InsertDestination.hpp:490-660).
//
// Three passes; every workgroup owns one contiguous row range and walks it in 2048-row tiles, so
// the scatter is stable (rows keep their relative order inside a partition):
//   1. per-workgroup histograms, written partition-major  hist[p * G + b]
//   2. exclusive scan of that array  ->  absolute start of (partition p, workgroup b)
//   3. scatter: a tile's rows are ranked per partition (ballot + mbcnt inside a wave, a wave
//      scan per partition over the tile's (step, wave) cells), staged in LDS in partition
//      order and copied out as contiguous runs — one run per partition per tile, so the writes
//      are coalesced whatever P is (the first version wrote one 64-row batch at a time: 32-byte
//      fragments at P = 8, 1.24 ms / 100 M (key, tid) rows).
// Limits: at most 64 partitions (lane p owns partition p).

#include "common.hpp"
#include "block_runs.hpp"
#include "partition.hpp"
#include "scan.hpp"

#include <type_traits>
#include <vector>

namespace qsx {

constexpr int kPBlock = 256;
constexpr int kPWaves = kPBlock / kWave;
constexpr int kPSteps = 8;                        // rows per thread per tile
constexpr int kPTile = kPBlock * kPSteps;         // 2048 rows
constexpr int kPCells = kPSteps * kPWaves;        // 32 (step, wave) cells per tile: scanned by one wave

// MODE 0: the reference's partition function (identity hash); MODE 1: the top bits of a mixing hash
// (internal re-partitioning: aggregation groups, join table slices) — P must be a power of two there;
// MODE 2: a 6-bit radix digit of a 64-bit sort key (stable: one LSD radix sort pass, sort.hip).
// Where a row's key comes from: one INT / LONG column (zero-extended bit pattern = the identity hash), or several
// key columns packed into one 64-bit code on the fly (the aggregation's key code; saves materialising it).
template <typename KeyT>
struct ColumnKey {
  const KeyT *keys;
  __device__ __forceinline__ void rebase(const void *stripe) { keys = static_cast<const KeyT *>(stripe); }
  __device__ __forceinline__ unsigned long long operator()(int64_t row) const {
    if (sizeof(KeyT) == 4) return static_cast<uint32_t>(keys[row]);
    return static_cast<unsigned long long>(keys[row]);
  }
};
struct PackedKey {
  const void *col[QSX_MAX_KEYS];
  int width[QSX_MAX_KEYS];
  int shift[QSX_MAX_KEYS];
  int num;
  __device__ __forceinline__ unsigned long long operator()(int64_t row) const {
    unsigned long long code = 0;
#pragma unroll
    for (int k = 0; k < QSX_MAX_KEYS; ++k) {
      if (k < num) {
        unsigned long long v;
        switch (width[k]) {
          case 1: v = static_cast<const uint8_t *>(col[k])[row]; break;
          case 2: v = static_cast<const uint16_t *>(col[k])[row]; break;
          case 4: v = static_cast<const uint32_t *>(col[k])[row]; break;
          default: v = static_cast<const unsigned long long *>(col[k])[row]; break;
        }
        code |= v << shift[k];
      }
    }
    return code;
  }
};

template <int MODE>
__device__ __forceinline__ int partition_of(unsigned long long h, int P, int pow2) {
  if (MODE == 2) return static_cast<int>((h >> pow2) & static_cast<unsigned long long>(P - 1));   // radix digit, pow2 = bit shift
  // MODE 3: a digit of the MIXING hash (the hash the aggregation's global table is addressed by, agg_common.hpp code_slot), keeping
  // the order of the digit below it: two passes — bits 52-57, then bits 58-63 — order the rows by the hash's top 12 bits
  // (partition_scatter_packed_digit)
  // (MODE 4: the same digit without the row order — the FIRST pass of an LSD ordering has no earlier order to keep)
  if (MODE == 3 || MODE == 4) return static_cast<int>(((mix64(h) * 0x9E3779B97F4A7C15ull) >> pow2) & static_cast<unsigned long long>(P - 1));
  if (MODE == 1) return static_cast<int>((mix64(h) * 0x9E3779B97F4A7C15ull) >> (64 - pow2));   // pow2 = log2(P) here
  if (pow2) return static_cast<int>(h & static_cast<unsigned long long>(P - 1));
  return static_cast<int>(h >= static_cast<unsigned long long>(P) ? h % static_cast<unsigned long long>(P) : h);
}

// One 64-row step of a wave: rank of every row among the rows of its partition (row order) and, in
// lane p, the number of rows of partition p.  kSmallP (P <= 8): P independent ballots, no dependent
// chain (the general loop walks the partitions present one at a time through a shuffle).
// MODE 3: the digit the previous pass ordered the rows by (the one right below this pass's).
__device__ __forceinline__ int digit_below(unsigned long long h, int P, int pow2) {
  return static_cast<int>(((mix64(h) * 0x9E3779B97F4A7C15ull) >> (pow2 - 6)) & static_cast<unsigned long long>(P - 1));
}

template <bool kSmallP>
__device__ __forceinline__ void step_ranks(int pid, int P, int &rank, int &count_in_lane) {
  // Bit-sliced match: one ballot per bit of the partition number.  A lane's peers are the lanes that agree with it on
  // every bit; lane p counts partition p by agreeing with its own lane number instead.  log2(P) ballots whatever the
  // number of partitions present (the previous general path walked them one by one: 3.15 ms per 100 M rows at P = 64).
  const int lane = lane_id();
  const uint64_t live = __ballot(pid >= 0);
  uint64_t peers = live, mine = live;
  const int bits = kSmallP ? 3 : (P <= 16 ? 4 : (P <= 32 ? 5 : 6));
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    if (b < bits) {
      const uint64_t set = __ballot(pid >= 0 && ((pid >> b) & 1));
      peers &= ((pid >> b) & 1) ? set : ~set;
      mine &= ((lane >> b) & 1) ? set : ~set;
    }
  }
  rank = pid >= 0 ? rank_below(peers) : 0;
  count_in_lane = lane < P ? __popcll(mine) : 0;
}

// ---- a run of storage blocks as K9's input (qsx_partition_scatter_blocks) ---------------------------------------------
// The repartitioning Select of a partitioned join reads a stored relation: tens to hundreds of blocks, every one with its own
// stripes.  Laid end to end first (one qsx_copy_segments launch) the rows cross HBM twice more than the scatter needs; here a
// workgroup's chunk of rows lies inside ONE block — the run table of block_runs.hpp with a chunk as its "tile", in[b] = the key
// stripe of block b, and behind it one more array of nb words per column: the column's stripe in every block — so a workgroup
// looks its block up once (scalar loads) and walks block-local rows.  Chunks follow each other in row order: the (partition,
// workgroup) cells scan to the same places as over the concatenation and the scatter stays stable.
template <bool kRuns>
struct ChunkSource {
  int block;
  long long cols_at;   // word of the table where column 0's array starts
  long long nb;
};
template <bool kRuns>
__device__ __forceinline__ const void *chunk_column(const long long *__restrict__ runs, const ChunkSource<kRuns> &at, const void *plain, int c) {
  if constexpr (kRuns) return as_global(reinterpret_cast<const unsigned char *>(runs[at.cols_at + static_cast<long long>(c) * at.nb + at.block]));
  return plain;
}

struct ScatterArgs {
  int ncols;
  int width[QSX_MAX_COLUMNS];
  const void *src[QSX_MAX_COLUMNS];
  void *dst[QSX_MAX_COLUMNS];
};

// PT > 0: the partition count as a constant (8 = one partition per GPU of a node, the shuffle's case); 0 = run-time P.
template <typename Loader, int MODE, bool kSmallP, int PT, bool kRuns>
__device__ __forceinline__ void partition_hist_body(Loader load_key, int64_t n, int P_arg, int pow2, int64_t rows_per_block, int64_t G,
                                                    int32_t *__restrict__ hist, const long long *__restrict__ runs) {
  const int P = PT > 0 ? PT : P_arg;
  __shared__ int s_total[kWave];
  const int lane = lane_id();
  if (threadIdx.x < kWave) s_total[threadIdx.x] = 0;
  __syncthreads();
  int64_t begin = static_cast<int64_t>(blockIdx.x) * rows_per_block;
  if constexpr (kRuns) {   // this workgroup's chunk of its block (rows are block-local from here on)
    const RunTile at = run_locate(runs, static_cast<int>(blockIdx.x));
    begin = static_cast<int64_t>(at.tile_in_block) * rows_per_block;
    n = run_rows(runs, at.block);
    load_key.rebase(run_in<unsigned char>(runs, at.block));
  }
  const int64_t end = begin + rows_per_block < n ? begin + rows_per_block : n;
  constexpr bool kIntKeys = std::is_same<Loader, ColumnKey<int32_t>>::value;
  if constexpr (kSmallP && (kIntKeys || std::is_same<Loader, ColumnKey<int64_t>>::value)) {
    // Counts need no row order: four (INT) or two (LONG) keys per 16-byte read, a thread's counts of the (at most 8)
    // partitions packed into the bytes of one register and unpacked every 63 reads (252 < 256).  The ranking form below
    // reads one key per lane and spends three ballots per 64 rows on ranks nobody asks for here: 0.145 ms per 100 M INT keys
    // against 0.09.
    constexpr int kShift = kIntKeys ? 2 : 1;   // log2 of the keys per 16 bytes
    const auto *keys = load_key.keys;
    if ((reinterpret_cast<uintptr_t>(keys) & 15) == 0) {   // (begin is a multiple of the tile: 16-byte aligned with the stripe)
      unsigned long long packed = 0;
      unsigned int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      auto unpack = [&]() {
#pragma unroll
        for (int p = 0; p < 8; ++p) cnt[p] += static_cast<unsigned int>(packed >> (8 * p)) & 255u;
        packed = 0;
      };
      // whole 16-byte groups of the chunk (a trailing workgroup's chunk may lie behind the stripe: nothing to count)
      const int64_t first = begin >> kShift, last = begin < end ? end >> kShift : first;
      int since = 0;
      constexpr int U = 4;
      for (int64_t v0 = first + threadIdx.x; v0 < last; v0 += static_cast<int64_t>(kPBlock) * U) {
        uint4 k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t v = v0 + static_cast<int64_t>(u) * kPBlock;
          k[u] = reinterpret_cast<const uint4 *>(keys)[v < last ? v : last - 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (v0 + static_cast<int64_t>(u) * kPBlock < last) {
            if constexpr (kIntKeys) {
              packed += 1ull << (8 * partition_of<MODE>(k[u].x, P, pow2));
              packed += 1ull << (8 * partition_of<MODE>(k[u].y, P, pow2));
              packed += 1ull << (8 * partition_of<MODE>(k[u].z, P, pow2));
              packed += 1ull << (8 * partition_of<MODE>(k[u].w, P, pow2));
            } else {
              packed += 1ull << (8 * partition_of<MODE>(k[u].x | (static_cast<unsigned long long>(k[u].y) << 32), P, pow2));
              packed += 1ull << (8 * partition_of<MODE>(k[u].z | (static_cast<unsigned long long>(k[u].w) << 32), P, pow2));
            }
          }
        }
        since += U;
        if (since + U > 63) {
          unpack();
          since = 0;
        }
      }
      for (int64_t row = (last << kShift) + threadIdx.x; begin < end && row < end; row += kPBlock) {   // (the stripe's last, partial group)
        packed += 1ull << (8 * partition_of<MODE>(load_key(row), P, pow2));
      }
      unpack();
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const unsigned int total = wave_reduce_add(cnt[p]);
        if (lane == 0 && total != 0 && p < P) atomicAdd(&s_total[p], static_cast<int>(total));
      }
      __syncthreads();
      if (threadIdx.x < P) hist[static_cast<int64_t>(threadIdx.x) * G + blockIdx.x] = s_total[threadIdx.x];
      return;
    }
  }
  int my_count = 0;  // lane p counts partition p
  for (int64_t tile = begin; tile < end; tile += kPTile) {
    int pid[kPSteps];
#pragma unroll
    for (int j = 0; j < kPSteps; ++j) {
      const int64_t row = tile + j * kPBlock + threadIdx.x;
      pid[j] = row < end ? partition_of<MODE>(load_key(row), P, pow2) : -1;
    }
    if ((MODE == 1 || MODE == 3 || MODE == 4) && !kSmallP) {
      // internal re-partitioning needs no row order (and a count never does): one LDS atomic per row instead of the ranking loop
#pragma unroll
      for (int j = 0; j < kPSteps; ++j) {
        // (clustered keys: a wave's 64 rows are one partition's — 64 adds to one LDS word go one after the other; one add then)
        const uint64_t live = __ballot(pid[j] >= 0);
        if (live == 0) continue;
        const int first = __builtin_amdgcn_readlane(pid[j], __ffsll(static_cast<long long>(live)) - 1);
        if (__ballot(pid[j] == first) == live) {
          if (lane == 0) atomicAdd(&s_total[first], static_cast<int>(__popcll(live)));
        } else if (pid[j] >= 0) {
          atomicAdd(&s_total[pid[j]], 1);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < kPSteps; ++j) {
        int rank, count;
        step_ranks<kSmallP>(pid[j], P, rank, count);
        my_count += count;
      }
    }
  }
  if (lane < P && my_count != 0) atomicAdd(&s_total[lane], my_count);
  __syncthreads();
  if (threadIdx.x < P) hist[static_cast<int64_t>(threadIdx.x) * G + blockIdx.x] = s_total[threadIdx.x];
}
template <typename Loader, int MODE, bool kSmallP, int PT = 0>
__global__ __launch_bounds__(kPBlock) void partition_hist_kernel(Loader load_key, int64_t n, int P_arg,
                                                                int pow2, int64_t rows_per_block, int64_t G,
                                                                int32_t *__restrict__ hist) {
  partition_hist_body<Loader, MODE, kSmallP, PT, false>(load_key, n, P_arg, pow2, rows_per_block, G, hist, nullptr);
}
template <typename Loader, int MODE, bool kSmallP, int PT = 0>
__global__ __launch_bounds__(kPBlock) void partition_hist_runs_kernel(Loader load_key, const long long *__restrict__ runs, int P_arg,
                                                                     int pow2, int64_t rows_per_block, int64_t G,
                                                                     int32_t *__restrict__ hist) {
  partition_hist_body<Loader, MODE, kSmallP, PT, true>(load_key, 0, P_arg, pow2, rows_per_block, G, hist, runs);
}

template <int W>
__device__ __forceinline__ void stage_and_copy(const void *src, void *dst, unsigned char *stage, const int64_t (&row)[kPSteps],
                                               const int (&pos)[kPSteps], const int (&pid)[kPSteps],
                                               const unsigned char *s_pid, const int *s_part_start,
                                               const long long *s_glob, int tile_rows) {
  using V = typename std::conditional<W == 1, uint8_t, typename std::conditional<W == 2, uint16_t,
            typename std::conditional<W == 4, uint32_t, uint64_t>::type>::type>::type;
  V v[kPSteps];
#pragma unroll
  for (int j = 0; j < kPSteps; ++j) v[j] = pid[j] >= 0 ? static_cast<const V *>(src)[row[j]] : V();
#pragma unroll
  for (int j = 0; j < kPSteps; ++j) {
    if (pid[j] >= 0) reinterpret_cast<V *>(stage)[pos[j]] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kPSteps; ++j) {   // fixed trip count: the 16 slot reads / stores of a thread are independent
    const int i = j * kPBlock + threadIdx.x;
    if (i < tile_rows) {
      const int p = s_pid[i];
      static_cast<V *>(dst)[s_glob[p] + (i - s_part_start[p])] = reinterpret_cast<const V *>(stage)[i];
    }
  }
  __syncthreads();
}

template <typename Loader, int MODE, bool kSmallP, int PT, bool kRuns>
__device__ __forceinline__ void partition_scatter_body(Loader load_key, int64_t n, int P_arg, int pow2, int64_t rows_per_block, int64_t G,
                                                       const int64_t *__restrict__ starts, const ScatterArgs &args, int stage_width,
                                                       int64_t *__restrict__ out_offsets, const long long *__restrict__ runs) {
  // dynamic LDS: stage[kPTile * widest column] | cnt[kPCells * P] — sized by the launch so that small
  // P / narrow columns leave room for more workgroups per CU (the tile loop is a chain of
  // load -> LDS -> barrier -> store phases: occupancy is what hides their latencies)
  const int P = PT > 0 ? PT : P_arg;
  extern __shared__ __attribute__((aligned(8))) unsigned char s_dyn[];
  unsigned char *s_stage = s_dyn;
  int *s_cnt = reinterpret_cast<int *>(s_dyn + static_cast<size_t>(kPTile) * stage_width);  // [cell][partition]
  __shared__ int s_part_start[kWave + 1];         // tile-local start of every partition's run
  __shared__ long long s_glob[kWave];             // next global row of (partition, this workgroup)
  __shared__ unsigned char s_pid[kPTile];         // partition of every staged slot
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && out_offsets != nullptr) {
    if (threadIdx.x < P) out_offsets[threadIdx.x] = starts[static_cast<int64_t>(threadIdx.x) * G];
    if (threadIdx.x == 0) out_offsets[P] = n;
  }
  if (threadIdx.x < P) s_glob[threadIdx.x] = starts[static_cast<int64_t>(threadIdx.x) * G + blockIdx.x];
  int64_t begin = static_cast<int64_t>(blockIdx.x) * rows_per_block;
  ChunkSource<kRuns> chunk{0, 0, 0};
  if constexpr (kRuns) {   // (n, the rows of the whole run, is out_offsets[P] above; from here on: of this workgroup's block)
    const RunTile at = run_locate(runs, static_cast<int>(blockIdx.x));
    chunk.block = at.block;
    chunk.nb = runs[0];
    chunk.cols_at = kRunHeaderWords + (chunk.nb + 1) + 5 * chunk.nb;
    begin = static_cast<int64_t>(at.tile_in_block) * rows_per_block;
    n = run_rows(runs, at.block);
    load_key.rebase(run_in<unsigned char>(runs, at.block));
  }
  const int64_t end = begin + rows_per_block < n ? begin + rows_per_block : n;
  // the keys of the NEXT tile are requested before the current tile goes through its rank / stage / copy phases (each
  // of them ends in a barrier): one exposed round trip less per tile
  using KeyValue = decltype(load_key(static_cast<int64_t>(0)));
  KeyValue key[kPSteps], next_key[kPSteps];
#pragma unroll
  for (int j = 0; j < kPSteps; ++j) {
    const int64_t r = begin + j * kPBlock + threadIdx.x;
    key[j] = r < end ? load_key(r) : KeyValue();
  }
  // (MODE 3: the keys at both ends of a tile, requested one tile ahead like the tile's own keys — see `unordered` below)
  constexpr bool kBorders = MODE == 3 && !kSmallP;
  KeyValue edge_first = KeyValue(), edge_last = KeyValue();
  if (kBorders && begin < end) {
    edge_first = load_key(begin);
    edge_last = load_key((begin + kPTile < end ? begin + kPTile : end) - 1);
  }
  for (int64_t tile = begin; tile < end; tile += kPTile) {
    const int tile_rows = static_cast<int>(end - tile < kPTile ? end - tile : kPTile);
    int64_t row[kPSteps];
    int pid[kPSteps], pos[kPSteps];
#pragma unroll
    for (int j = 0; j < kPSteps; ++j) {
      const int64_t r = tile + kPTile + j * kPBlock + threadIdx.x;
      next_key[j] = r < end ? load_key(r) : KeyValue();
    }
    KeyValue next_first = KeyValue(), next_last = KeyValue();
    if (kBorders && tile + kPTile < end) {
      next_first = load_key(tile + kPTile);
      next_last = load_key((tile + 2 * kPTile < end ? tile + 2 * kPTile : end) - 1);
    }
#pragma unroll
    for (int j = 0; j < kPSteps; ++j) {
      row[j] = tile + j * kPBlock + threadIdx.x;
      pid[j] = row[j] < end ? partition_of<MODE>(key[j], P, pow2) : -1;
      key[j] = next_key[j];
    }
    constexpr bool kUnordered = (MODE == 1 || MODE == 4) && !kSmallP;
    bool unordered = kUnordered;
    if constexpr (MODE == 3 && !kSmallP) {
      // The order this pass has to keep is the previous digit's, and its output is what this pass reads: 64 runs of ~n / 64 rows.
      // A tile whose ends carry the same previous digit lies inside one run — nothing to keep — and only the tiles on the 63
      // borders pay for the ranks in row order (1.57 -> 0.9 ms per 100 M rows of a 4-byte key and an 8-byte value).
      if (pow2 >= 6) unordered = digit_below(edge_first, P, pow2) == digit_below(edge_last, P, pow2);
      edge_first = next_first;
      edge_last = next_last;
    }
    if (unordered) {
      // no row order to keep: the rank inside the tile is an LDS fetch-add on the partition's counter
      if (threadIdx.x < P) s_cnt[threadIdx.x] = 0;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < kPSteps; ++j) {
        // (a wave whose 64 rows are one partition's — clustered keys — takes its 64 places with one add: 64 adds to one LDS word
        // go one after the other)
        const uint64_t live = __ballot(pid[j] >= 0);
        pos[j] = 0;
        if (live == 0) continue;
        const int leader = __ffsll(static_cast<long long>(live)) - 1;
        const int first = __builtin_amdgcn_readlane(pid[j], leader);
        if (__ballot(pid[j] == first) == live) {
          int base = 0;
          if (lane == leader) base = atomicAdd(&s_cnt[first], static_cast<int>(__popcll(live)));
          pos[j] = __builtin_amdgcn_readlane(base, leader) + rank_below(live);
        } else if (pid[j] >= 0) {
          pos[j] = atomicAdd(&s_cnt[pid[j]], 1);
        }
      }
      __syncthreads();
      if (threadIdx.x < P) s_part_start[threadIdx.x + 1] = s_cnt[threadIdx.x];
    } else {
    // rank inside the wave + the cell's count per partition
#pragma unroll
    for (int j = 0; j < kPSteps; ++j) {
      int count;
      step_ranks<kSmallP>(pid[j], P, pos[j], count);
      if (lane < P) s_cnt[(j * kPWaves + wave) * P + lane] = count;   // every cell is written: no zeroing pass
    }
    __syncthreads();
    // per partition: exclusive scan over the 64 cells (lane = cell), total into s_part_start[p + 1]
    for (int p = wave; p < P; p += kPWaves) {
      const int c = lane < kPCells ? s_cnt[lane * P + p] : 0;
      int incl = c;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const int up = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += up;
      }
      if (lane < kPCells) s_cnt[lane * P + p] = incl - c;
      if (lane == kWave - 1) s_part_start[p + 1] = incl;
    }
    }
    __syncthreads();
    if (wave == 0) {  // exclusive scan of the partition totals (lane = partition)
      const int c = lane < P ? s_part_start[lane + 1] : 0;
      int incl = c;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const int up = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += up;
      }
      if (lane < P) s_part_start[lane + 1] = incl;
      if (lane == 0) s_part_start[0] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPSteps; ++j) {
      if (pid[j] >= 0) {
        pos[j] += s_part_start[pid[j]] + (unordered ? 0 : s_cnt[(j * kPWaves + wave) * P + pid[j]]);
        s_pid[pos[j]] = static_cast<unsigned char>(pid[j]);
      }
    }
    __syncthreads();
    for (int c = 0; c < args.ncols; ++c) {
      const void *src = chunk_column<kRuns>(runs, chunk, args.src[c], c);
      switch (args.width[c]) {
        case 1: stage_and_copy<1>(src, args.dst[c], s_stage, row, pos, pid, s_pid, s_part_start, s_glob, tile_rows); break;
        case 2: stage_and_copy<2>(src, args.dst[c], s_stage, row, pos, pid, s_pid, s_part_start, s_glob, tile_rows); break;
        case 4: stage_and_copy<4>(src, args.dst[c], s_stage, row, pos, pid, s_pid, s_part_start, s_glob, tile_rows); break;
        default: stage_and_copy<8>(src, args.dst[c], s_stage, row, pos, pid, s_pid, s_part_start, s_glob, tile_rows); break;
      }
    }
    if (threadIdx.x < P) s_glob[threadIdx.x] += s_part_start[threadIdx.x + 1] - s_part_start[threadIdx.x];
    __syncthreads();
  }
}
template <typename Loader, int MODE, bool kSmallP, int PT = 0>
__global__ __launch_bounds__(kPBlock) __attribute__((amdgpu_waves_per_eu(MODE == 3 && !kSmallP ? 4 : 1))) void partition_scatter_kernel(Loader load_key, int64_t n, int P_arg,
                                                                   int pow2, int64_t rows_per_block, int64_t G,
                                                                   const int64_t *__restrict__ starts,
                                                                   ScatterArgs args, int stage_width,
                                                                   int64_t *__restrict__ out_offsets) {
  partition_scatter_body<Loader, MODE, kSmallP, PT, false>(load_key, n, P_arg, pow2, rows_per_block, G, starts, args, stage_width, out_offsets, nullptr);
}
template <typename Loader, int MODE, bool kSmallP, int PT = 0>
__global__ __launch_bounds__(kPBlock) void partition_scatter_runs_kernel(Loader load_key, int64_t n, const long long *__restrict__ runs, int P_arg,
                                                                        int pow2, int64_t rows_per_block, int64_t G,
                                                                        const int64_t *__restrict__ starts,
                                                                        ScatterArgs args, int stage_width,
                                                                        int64_t *__restrict__ out_offsets) {
  partition_scatter_body<Loader, MODE, kSmallP, PT, true>(load_key, n, P_arg, pow2, rows_per_block, G, starts, args, stage_width, out_offsets, runs);
}

// The scatter pass for at most 8 partitions whose columns fit the staging area side by side (the shuffle of a join side
// across the GPUs of a node: key + a few payload columns into 8).  Same result as partition_scatter_kernel — stable, one
// contiguous run per partition per tile — with three workgroup barriers per tile instead of 4 + 2 per column: after the
// cell counts are in LDS every wave derives the offsets of its own rows by itself (lane p walks partition p's 32 cells and
// the partitions' totals are scanned across 8 lanes), and all columns are staged before the one barrier that precedes the
// copy-out.  The general kernel spends its time in those barriers, not on bytes (open items).
struct SmallScatterLayout {
  int stage_off[QSX_MAX_COLUMNS];   // byte offset of column c's slots in the staging area
};
// (PT == 8, the shuffle's form, fits 80 registers without scratch: six waves per SIMD instead of five)
template <typename Loader, int MODE, int PT, bool kRuns>
__device__ __forceinline__ void partition_scatter_small_body(Loader load_key, int64_t n, int P_arg, int pow2, int64_t rows_per_block, int64_t G,
                                                             const int64_t *__restrict__ starts, const ScatterArgs &args,
                                                             const SmallScatterLayout &layout, int stage_bytes,
                                                             int64_t *__restrict__ out_offsets, const long long *__restrict__ runs) {
  const int P = PT > 0 ? PT : P_arg;
  extern __shared__ __attribute__((aligned(8))) unsigned char s_dyn[];
  unsigned char *s_stage = s_dyn;
  int *s_cnt = reinterpret_cast<int *>(s_dyn + stage_bytes);   // [cell][partition], cell = step * kPWaves + wave: row order
  __shared__ int s_part_start[kWave + 1];
  __shared__ long long s_glob[kWave];
  __shared__ unsigned char s_pid[kPTile];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && out_offsets != nullptr) {
    if (threadIdx.x < P) out_offsets[threadIdx.x] = starts[static_cast<int64_t>(threadIdx.x) * G];
    if (threadIdx.x == 0) out_offsets[P] = n;
  }
  if (threadIdx.x < P) s_glob[threadIdx.x] = starts[static_cast<int64_t>(threadIdx.x) * G + blockIdx.x];
  int64_t begin = static_cast<int64_t>(blockIdx.x) * rows_per_block;
  ChunkSource<kRuns> chunk{0, 0, 0};
  if constexpr (kRuns) {
    const RunTile at = run_locate(runs, static_cast<int>(blockIdx.x));
    chunk.block = at.block;
    chunk.nb = runs[0];
    chunk.cols_at = kRunHeaderWords + (chunk.nb + 1) + 5 * chunk.nb;
    begin = static_cast<int64_t>(at.tile_in_block) * rows_per_block;
    n = run_rows(runs, at.block);
    load_key.rebase(run_in<unsigned char>(runs, at.block));
  }
  const int64_t end = begin + rows_per_block < n ? begin + rows_per_block : n;
  using KeyValue = decltype(load_key(static_cast<int64_t>(0)));
  KeyValue key[kPSteps], next_key[kPSteps];
#pragma unroll
  for (int j = 0; j < kPSteps; ++j) {
    const int64_t r = begin + j * kPBlock + threadIdx.x;
    key[j] = r < end ? load_key(r) : KeyValue();
  }
  for (int64_t tile = begin; tile < end; tile += kPTile) {
    const int tile_rows = static_cast<int>(end - tile < kPTile ? end - tile : kPTile);
    int64_t row[kPSteps];
    int pid[kPSteps], pos[kPSteps];
#pragma unroll
    for (int j = 0; j < kPSteps; ++j) {
      const int64_t r = tile + kPTile + j * kPBlock + threadIdx.x;
      next_key[j] = r < end ? load_key(r) : KeyValue();
    }
#pragma unroll
    for (int j = 0; j < kPSteps; ++j) {
      row[j] = tile + j * kPBlock + threadIdx.x;
      pid[j] = row[j] < end ? partition_of<MODE>(key[j], P, pow2) : -1;
      key[j] = next_key[j];
    }
#pragma unroll
    for (int j = 0; j < kPSteps; ++j) {
      int count;
      step_ranks<true>(pid[j], P, pos[j], count);
      if (lane < P) s_cnt[(j * kPWaves + wave) * P + lane] = count;
    }
    __syncthreads();   // (1) the cell counts
    {
      // lane 8 j + p holds (step j, partition p): the four waves' cells of that step, then an exclusive scan over the steps
      // (lanes 8 apart) — the rows of partition p in front of this wave's cell of step j — and, in the last step's lanes, the
      // partition's total, scanned over the 8 partitions for their starts.  (Walking a partition's 32 cells in one lane
      // kept 32 registers in flight: 116 VGPRs, four waves per SIMD.)
      static_assert(kPSteps == 8 && kPWaves == 4, "the (step, partition) lane layout below");
      const int cell_step = lane >> 3, cell_part = lane & 7;
      int in_step_before = 0, step_total = 0;
#pragma unroll
      for (int w = 0; w < kPWaves; ++w) {
        const int c = cell_part < P ? s_cnt[(cell_step * kPWaves + w) * P + cell_part] : 0;
        if (w == wave) in_step_before = step_total;
        step_total += c;
      }
      int incl = step_total;
#pragma unroll
      for (int off = 8; off < kWave; off <<= 1) {
        const int up = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += up;
      }
      const int before_cell = incl - step_total + in_step_before;   // rows of (partition) in front of (step, this wave)
      // totals: lanes 56 + p
      int total = __shfl(incl, 56 + cell_part, kWave);
      int scan = cell_step == 0 ? total : 0;                         // lanes 0..7 scan the 8 totals
#pragma unroll
      for (int off = 1; off < 8; off <<= 1) {
        const int up = __shfl_up(scan, off, kWave);
        if (lane >= off && lane < 8) scan += up;
      }
      const int start_of_part = __shfl(scan - total, cell_part, kWave);   // exclusive: lane p of the first row of lanes
      if (wave == 0 && lane < 8) {
        if (lane < P) s_part_start[lane] = scan - total;
        if (lane == P - 1) s_part_start[P] = scan;
      }
      const int cell_base = start_of_part + before_cell;
#pragma unroll
      for (int j = 0; j < kPSteps; ++j) {
        const int base = __shfl(cell_base, (j << 3) + (pid[j] >= 0 ? pid[j] : 0), kWave);
        pos[j] += base;
        if (pid[j] >= 0) s_pid[pos[j]] = static_cast<unsigned char>(pid[j]);
      }
    }
    for (int c = 0; c < args.ncols; ++c) {   // every column into its own part of the staging area: no barrier in between
      unsigned char *stage = s_stage + layout.stage_off[c];
      const void *src = chunk_column<kRuns>(runs, chunk, args.src[c], c);
      switch (args.width[c]) {
        case 1: {
          uint8_t v[kPSteps];
#pragma unroll
          for (int j = 0; j < kPSteps; ++j) v[j] = pid[j] >= 0 ? load_global(&static_cast<const uint8_t *>(src)[row[j]]) : uint8_t(0);
#pragma unroll
          for (int j = 0; j < kPSteps; ++j) if (pid[j] >= 0) reinterpret_cast<uint8_t *>(stage)[pos[j]] = v[j];
          break;
        }
        case 2: {
          uint16_t v[kPSteps];
#pragma unroll
          for (int j = 0; j < kPSteps; ++j) v[j] = pid[j] >= 0 ? load_global(&static_cast<const uint16_t *>(src)[row[j]]) : uint16_t(0);
#pragma unroll
          for (int j = 0; j < kPSteps; ++j) if (pid[j] >= 0) reinterpret_cast<uint16_t *>(stage)[pos[j]] = v[j];
          break;
        }
        case 4: {
          uint32_t v[kPSteps];
#pragma unroll
          for (int j = 0; j < kPSteps; ++j) v[j] = pid[j] >= 0 ? load_global(&static_cast<const uint32_t *>(src)[row[j]]) : 0u;
#pragma unroll
          for (int j = 0; j < kPSteps; ++j) if (pid[j] >= 0) reinterpret_cast<uint32_t *>(stage)[pos[j]] = v[j];
          break;
        }
        default: {
          uint64_t v[kPSteps];
#pragma unroll
          for (int j = 0; j < kPSteps; ++j) v[j] = pid[j] >= 0 ? load_global(&static_cast<const uint64_t *>(src)[row[j]]) : 0ull;
#pragma unroll
          for (int j = 0; j < kPSteps; ++j) if (pid[j] >= 0) reinterpret_cast<uint64_t *>(stage)[pos[j]] = v[j];
          break;
        }
      }
    }
    __syncthreads();   // (2) slots, their partitions, the partitions' starts
#pragma unroll
    for (int j = 0; j < kPSteps; ++j) {
      const int i = j * kPBlock + threadIdx.x;
      if (i < tile_rows) {
        const int p = s_pid[i];
        const long long at = s_glob[p] + (i - s_part_start[p]);
        for (int c = 0; c < args.ncols; ++c) {
          const unsigned char *stage = s_stage + layout.stage_off[c];
          switch (args.width[c]) {
            case 1: store_global(reinterpret_cast<const uint8_t *>(stage)[i], &static_cast<uint8_t *>(args.dst[c])[at]); break;
            case 2: store_global(reinterpret_cast<const uint16_t *>(stage)[i], &static_cast<uint16_t *>(args.dst[c])[at]); break;
            case 4: store_global(reinterpret_cast<const uint32_t *>(stage)[i], &static_cast<uint32_t *>(args.dst[c])[at]); break;
            default: store_global(reinterpret_cast<const uint64_t *>(stage)[i], &static_cast<uint64_t *>(args.dst[c])[at]); break;
          }
        }
      }
    }
    __syncthreads();   // (3) the staging area and the starts are free again
    if (threadIdx.x < P) s_glob[threadIdx.x] += s_part_start[threadIdx.x + 1] - s_part_start[threadIdx.x];
  }
}
template <typename Loader, int MODE, int PT = 0>
__global__ __launch_bounds__(kPBlock) __attribute__((amdgpu_waves_per_eu(PT == 8 ? 6 : 4))) void partition_scatter_small_kernel(Loader load_key, int64_t n, int P_arg, int pow2,
                                                                         int64_t rows_per_block, int64_t G,
                                                                         const int64_t *__restrict__ starts, ScatterArgs args,
                                                                         SmallScatterLayout layout, int stage_bytes,
                                                                         int64_t *__restrict__ out_offsets) {
  partition_scatter_small_body<Loader, MODE, PT, false>(load_key, n, P_arg, pow2, rows_per_block, G, starts, args, layout, stage_bytes, out_offsets, nullptr);
}
template <typename Loader, int MODE, int PT = 0>
__global__ __launch_bounds__(kPBlock) __attribute__((amdgpu_waves_per_eu(PT == 8 ? 6 : 4))) void partition_scatter_small_runs_kernel(Loader load_key, int64_t n, const long long *__restrict__ runs,
                                                                         int P_arg, int pow2, int64_t rows_per_block, int64_t G,
                                                                         const int64_t *__restrict__ starts, ScatterArgs args,
                                                                         SmallScatterLayout layout, int stage_bytes,
                                                                         int64_t *__restrict__ out_offsets) {
  partition_scatter_small_body<Loader, MODE, PT, true>(load_key, n, P_arg, pow2, rows_per_block, G, starts, args, layout, stage_bytes, out_offsets, runs);
}

// Internal re-partitioning with aligned pieces: every partition's run starts at a multiple of `align` rows of the
// output (so that a consumer can DMA 16-byte chunks of any column straight out of a piece); the gaps are never
// read.  Rewrites the (partition, workgroup) start cells in place and publishes pieces[p] = start row,
// pieces[P + p] = row count.
__global__ __launch_bounds__(1024) void align_starts_kernel(int64_t *__restrict__ starts, int P, int64_t G, int64_t n, int align,
                                                              int64_t *__restrict__ pieces) {
  __shared__ long long s_shift[kWave];
  if (threadIdx.x < kWave) {   // wave 0: lane p owns partition p; aligned starts = exclusive scan of the padded sizes
    const int p = threadIdx.x;
    long long begin = 0, size = 0;
    if (p < P) {
      begin = starts[static_cast<int64_t>(p) * G];
      const long long end = p + 1 < P ? starts[static_cast<int64_t>(p + 1) * G] : n;
      size = end - begin;
    }
    const long long padded = (size + align - 1) / align * align;
    long long incl = padded;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const long long up = __shfl_up(incl, off, kWave);
      if (p >= off) incl += up;
    }
    const long long next = incl - padded;
    if (p < P) {
      s_shift[p] = next - begin;
      pieces[p] = next;
      pieces[P + p] = size;
    }
  }
  __syncthreads();
  // (one workgroup: every shift depends on the unshifted first cell of two partitions)
  for (int p = 0; p < P; ++p) {
    const long long shift = s_shift[p];
    for (int64_t g = threadIdx.x; g < G; g += blockDim.x) starts[static_cast<int64_t>(p) * G + g] += shift;
  }
}

static size_t p_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int64_t blocks_for(int64_t n) {
  // whole tiles per workgroup, at most 4096 workgroups
  int64_t G = (n + kPTile - 1) / kPTile;
  if (G < 1) G = 1;
  // 4096 chunks: twice what is resident at most — the second half fills in behind workgroups that finish early (2048, one
  // chunk per resident workgroup of the lightest kernel, left the tail to the slowest chunk: 0.70 against 0.67 ms per 100 M
  // rows of (key, tid)); QSX_K9_GRID: sweeps
  static const int64_t cap = getenv("QSX_K9_GRID") != nullptr && atoll(getenv("QSX_K9_GRID")) > 0 ? atoll(getenv("QSX_K9_GRID")) : 2 * kMaxGridBlocks;
  if (G > cap) G = cap;
  return G;
}

size_t partition_workspace_bytes(int64_t n, int num_partitions) {
  const int64_t cells = blocks_for(n) * num_partitions;
  return p_align_up(sizeof(int64_t) * (cells + 1), 256) + p_align_up(sizeof(int32_t) * cells, 256) +
         scan_workspace_words(cells) * sizeof(int64_t);
}

// The rows a workgroup takes: whole tiles, blocks_for(n) chunks over one stripe.
static int64_t chunk_rows_for(int64_t n) {
  const int64_t G = blocks_for(n);
  const int64_t rows_per_block = (n + G - 1) / G;
  return (rows_per_block + kPTile - 1) / kPTile * kPTile;
}

// A run of blocks (ChunkSource above): chunks of chunk_rows_for(all rows) rows that never straddle two blocks — at most one
// chunk per block more than over the concatenation.
size_t partition_blocks_workspace_bytes(int64_t n, int64_t num_blocks, int num_partitions) {
  const int64_t cells = (blocks_for(n) + num_blocks) * num_partitions;
  return p_align_up(sizeof(int64_t) * (cells + 1), 256) + p_align_up(sizeof(int32_t) * cells, 256) +
         scan_workspace_words(cells) * sizeof(int64_t);
}

// runs_dev != nullptr: the run form — G_runs chunks of rows_runs rows, the keys and columns found through the table.
template <typename Loader, int MODE>
static int launch_partition_t(Loader keys, int64_t n, int P, int pow2, const ScatterArgs &args, int64_t *out_offsets,
                              void *workspace, int align_rows, hipStream_t s, const long long *runs_dev = nullptr,
                              int64_t G_runs = 0, int64_t rows_runs = 0) {
  constexpr bool kRunForms = MODE == 0 && !std::is_same<Loader, PackedKey>::value;   // (the C ABI's partition function only)
  if (runs_dev != nullptr && !kRunForms) return QSX_ERR_UNSUPPORTED;
  const int64_t G = runs_dev != nullptr ? G_runs : blocks_for(n);
  const int64_t rows_per_block = runs_dev != nullptr ? rows_runs : chunk_rows_for(n);
  const int64_t cells = G * P;
  int64_t *starts = static_cast<int64_t *>(workspace);
  char *after = static_cast<char *>(workspace) + p_align_up(sizeof(int64_t) * (cells + 1), 256);
  int32_t *hist = reinterpret_cast<int32_t *>(after);
  int64_t *scan_ws = reinterpret_cast<int64_t *>(after + p_align_up(sizeof(int32_t) * cells, 256));
  if constexpr (kRunForms) {
    if (runs_dev != nullptr) {
      if (P == 8) {
        hipLaunchKernelGGL((partition_hist_runs_kernel<Loader, MODE, true, 8>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), 0, s, keys,
                           runs_dev, P, pow2, rows_per_block, G, hist);
      } else if (P <= 8) {
        hipLaunchKernelGGL((partition_hist_runs_kernel<Loader, MODE, true>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), 0, s, keys,
                           runs_dev, P, pow2, rows_per_block, G, hist);
      } else {
        hipLaunchKernelGGL((partition_hist_runs_kernel<Loader, MODE, false>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), 0, s, keys,
                           runs_dev, P, pow2, rows_per_block, G, hist);
      }
    }
  }
  if (runs_dev != nullptr) {
    // (launched above)
  } else if (P == 8) {
    hipLaunchKernelGGL((partition_hist_kernel<Loader, MODE, true, 8>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), 0, s, keys, n, P,
                       pow2, rows_per_block, G, hist);
  } else if (P <= 8) {
    hipLaunchKernelGGL((partition_hist_kernel<Loader, MODE, true>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), 0, s, keys, n, P,
                       pow2, rows_per_block, G, hist);
  } else {
    hipLaunchKernelGGL((partition_hist_kernel<Loader, MODE, false>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), 0, s, keys, n, P,
                       pow2, rows_per_block, G, hist);
  }
  QSX_CHECK_LAUNCH();
  QSX_HIP_TRY(launch_scan(hist, cells, starts, nullptr, scan_ws, s));
  int64_t *block0_offsets = out_offsets;
  if (align_rows > 0) {
    hipLaunchKernelGGL(align_starts_kernel, dim3(1), dim3(1024), 0, s, starts, P, G, n, align_rows, out_offsets);
    QSX_CHECK_LAUNCH();
    block0_offsets = nullptr;   // out_offsets holds the pieces (start, count) instead of P + 1 boundaries
  }
  if (P <= 8 && args.ncols > 0) {
    // all columns staged side by side (8-byte aligned parts): up to 48 KiB of staging keeps three workgroups on a CU
    SmallScatterLayout layout{};
    int stage_bytes = 0;
    for (int c = 0; c < args.ncols; ++c) {
      layout.stage_off[c] = stage_bytes;
      stage_bytes += (kPTile * args.width[c] + 7) & ~7;
    }
    static const bool small_off = getenv("QSX_K9_SMALL") != nullptr && atoi(getenv("QSX_K9_SMALL")) == 0;
    if (stage_bytes <= 48 * 1024 && !small_off) {
      const size_t small_lds = static_cast<size_t>(stage_bytes) + sizeof(int) * kPCells * P;
      if constexpr (kRunForms) {
        if (runs_dev != nullptr) {
          if (P == 8) {
            hipLaunchKernelGGL((partition_scatter_small_runs_kernel<Loader, MODE, 8>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), small_lds,
                               s, keys, n, runs_dev, P, pow2, rows_per_block, G, starts, args, layout, stage_bytes, block0_offsets);
          } else {
            hipLaunchKernelGGL((partition_scatter_small_runs_kernel<Loader, MODE>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), small_lds,
                               s, keys, n, runs_dev, P, pow2, rows_per_block, G, starts, args, layout, stage_bytes, block0_offsets);
          }
          QSX_CHECK_LAUNCH();
          return QSX_OK;
        }
      }
      if (P == 8) {
        hipLaunchKernelGGL((partition_scatter_small_kernel<Loader, MODE, 8>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), small_lds, s,
                           keys, n, P, pow2, rows_per_block, G, starts, args, layout, stage_bytes, block0_offsets);
      } else {
        hipLaunchKernelGGL((partition_scatter_small_kernel<Loader, MODE>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), small_lds, s,
                           keys, n, P, pow2, rows_per_block, G, starts, args, layout, stage_bytes, block0_offsets);
      }
      QSX_CHECK_LAUNCH();
      return QSX_OK;
    }
  }
  int stage_width = 1;
  for (int c = 0; c < args.ncols; ++c) stage_width = args.width[c] > stage_width ? args.width[c] : stage_width;
  const size_t lds = static_cast<size_t>(kPTile) * stage_width + sizeof(int) * kPCells * P;
  if constexpr (kRunForms) {
    if (runs_dev != nullptr) {
      if (P == 8) {
        hipLaunchKernelGGL((partition_scatter_runs_kernel<Loader, MODE, true, 8>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), lds, s, keys,
                           n, runs_dev, P, pow2, rows_per_block, G, starts, args, stage_width, block0_offsets);
      } else if (P <= 8) {
        hipLaunchKernelGGL((partition_scatter_runs_kernel<Loader, MODE, true>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), lds, s, keys,
                           n, runs_dev, P, pow2, rows_per_block, G, starts, args, stage_width, block0_offsets);
      } else {
        hipLaunchKernelGGL((partition_scatter_runs_kernel<Loader, MODE, false>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), lds, s, keys,
                           n, runs_dev, P, pow2, rows_per_block, G, starts, args, stage_width, block0_offsets);
      }
      QSX_CHECK_LAUNCH();
      return QSX_OK;
    }
  }
  if (P == 8) {
    hipLaunchKernelGGL((partition_scatter_kernel<Loader, MODE, true, 8>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), lds, s, keys,
                       n, P, pow2, rows_per_block, G, starts, args, stage_width, block0_offsets);
  } else if (P <= 8) {
    hipLaunchKernelGGL((partition_scatter_kernel<Loader, MODE, true>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), lds, s, keys,
                       n, P, pow2, rows_per_block, G, starts, args, stage_width, block0_offsets);
  } else {
    hipLaunchKernelGGL((partition_scatter_kernel<Loader, MODE, false>), dim3(static_cast<unsigned>(G)), dim3(kPBlock), lds, s, keys,
                       n, P, pow2, rows_per_block, G, starts, args, stage_width, block0_offsets);
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

// mode 0: reference partition function (qsx_partition_scatter); mode 1: mixing-hash partitions for
// internal use (P a power of two).  Shared by the C entry point, the partitioned aggregation and
// the partitioned probe.
int partition_scatter_impl(int mode, int key_type, const void *keys_dev, int64_t n, int num_partitions, int ncols,
                           const void *const *cols, const int32_t *widths, void *const *out_cols, int64_t *out_offsets_dev,
                           void *workspace_dev, size_t workspace_bytes, hipStream_t s, int align_rows) {
  if (n < 0 || num_partitions < 1 || ncols < 0 || ncols > QSX_MAX_COLUMNS || out_offsets_dev == nullptr) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (num_partitions > kWave) return QSX_ERR_UNSUPPORTED;
  if (key_type != QSX_INT && key_type != QSX_LONG) return QSX_ERR_UNSUPPORTED;
  const bool is_pow2 = (num_partitions & (num_partitions - 1)) == 0;
  if (mode == 1 && !is_pow2) return QSX_ERR_UNSUPPORTED;
  if (n == 0) {
    QSX_HIP_TRY(hipMemsetAsync(out_offsets_dev, 0, sizeof(int64_t) * (num_partitions + 1), s));
    return QSX_OK;
  }
  if (workspace_bytes < partition_workspace_bytes(n, num_partitions) || workspace_dev == nullptr) return QSX_ERR_CAPACITY;
  ScatterArgs args;
  args.ncols = ncols;
  for (int c = 0; c < ncols; ++c) {
    const int w = widths[c];
    if (w != 1 && w != 2 && w != 4 && w != 8) return QSX_ERR_UNSUPPORTED;
    args.width[c] = w;
    args.src[c] = cols[c];
    args.dst[c] = out_cols[c];
  }
  int pow2 = is_pow2 ? 1 : 0;
  if (mode == 1) {
    pow2 = 0;
    while ((1 << pow2) < num_partitions) ++pow2;   // log2(P): the hash shift
    if (pow2 == 0) pow2 = 0;
  }
  if (mode == 1 && num_partitions == 1) {
    // one partition: a plain copy keeps the contract (shift by 64 would be undefined)
    mode = 0;
    pow2 = 1;
  }
  if (key_type == QSX_INT) {
    const ColumnKey<int32_t> k{static_cast<const int32_t *>(keys_dev)};
    return mode == 0 ? launch_partition_t<ColumnKey<int32_t>, 0>(k, n, num_partitions, pow2, args, out_offsets_dev, workspace_dev, align_rows, s)
                     : launch_partition_t<ColumnKey<int32_t>, 1>(k, n, num_partitions, pow2, args, out_offsets_dev, workspace_dev, align_rows, s);
  }
  const ColumnKey<int64_t> k{static_cast<const int64_t *>(keys_dev)};
  return mode == 0 ? launch_partition_t<ColumnKey<int64_t>, 0>(k, n, num_partitions, pow2, args, out_offsets_dev, workspace_dev, align_rows, s)
                   : launch_partition_t<ColumnKey<int64_t>, 1>(k, n, num_partitions, pow2, args, out_offsets_dev, workspace_dev, align_rows, s);
}

// One stable LSD radix pass: rows ordered by the 6-bit digit (keys64 >> shift) & 63, ties in input order.
int partition_scatter_digit(const unsigned long long *keys64_dev, int64_t n, int shift, int ncols, const void *const *cols,
                            const int32_t *widths, void *const *out_cols, int64_t *out_offsets_dev, void *workspace_dev,
                            size_t workspace_bytes, hipStream_t s) {
  if (n <= 0 || ncols < 0 || ncols > QSX_MAX_COLUMNS || out_offsets_dev == nullptr || shift < 0 || shift > 63) return QSX_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < partition_workspace_bytes(n, kWave) || workspace_dev == nullptr) return QSX_ERR_CAPACITY;
  ScatterArgs args;
  args.ncols = ncols;
  for (int c = 0; c < ncols; ++c) {
    args.width[c] = widths[c];
    args.src[c] = cols[c];
    args.dst[c] = out_cols[c];
  }
  const ColumnKey<int64_t> k{reinterpret_cast<const int64_t *>(keys64_dev)};
  return launch_partition_t<ColumnKey<int64_t>, 2>(k, n, kWave, shift, args, out_offsets_dev, workspace_dev, 0, s);
}

// Mixing-hash partitioning on a key code packed on the fly from up to QSX_MAX_KEYS key columns (mode 1 only).
int partition_scatter_packed_keys(int num_keys, const void *const *key_cols, const int *key_widths, const int *key_shifts,
                                  int64_t n, int num_partitions, int ncols, const void *const *cols, const int32_t *widths,
                                  void *const *out_cols, int64_t *out_offsets_dev, void *workspace_dev, size_t workspace_bytes,
                                  hipStream_t s, int align_rows) {
  if (n <= 0 || num_keys < 1 || num_keys > QSX_MAX_KEYS || num_partitions < 2 || num_partitions > kWave ||
      (num_partitions & (num_partitions - 1)) != 0 || ncols < 0 || ncols > QSX_MAX_COLUMNS || out_offsets_dev == nullptr) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (workspace_bytes < partition_workspace_bytes(n, num_partitions) || workspace_dev == nullptr) return QSX_ERR_CAPACITY;
  ScatterArgs args;
  args.ncols = ncols;
  for (int c = 0; c < ncols; ++c) {
    args.width[c] = widths[c];
    args.src[c] = cols[c];
    args.dst[c] = out_cols[c];
  }
  PackedKey k;
  k.num = num_keys;
  for (int i = 0; i < QSX_MAX_KEYS; ++i) {
    k.col[i] = i < num_keys ? key_cols[i] : nullptr;
    k.width[i] = i < num_keys ? key_widths[i] : 0;
    k.shift[i] = i < num_keys ? key_shifts[i] : 0;
  }
  int log2p = 0;
  while ((1 << log2p) < num_partitions) ++log2p;
  return launch_partition_t<PackedKey, 1>(k, n, num_partitions, log2p, args, out_offsets_dev, workspace_dev, align_rows, s);
}

// One pass of an LSD ordering by the mixing hash of a key code packed on the fly: 64 buckets by the digit (hash >> shift) & 63;
// stable (every pass but the first has to be) or not.
int partition_scatter_packed_digit(int num_keys, const void *const *key_cols, const int *key_widths, const int *key_shifts, int64_t n,
                                   int shift, bool stable, int ncols, const void *const *cols, const int32_t *widths, void *const *out_cols,
                                   int64_t *out_offsets_dev, void *workspace_dev, size_t workspace_bytes, hipStream_t s) {
  if (n <= 0 || num_keys < 1 || num_keys > QSX_MAX_KEYS || ncols < 0 || ncols > QSX_MAX_COLUMNS || out_offsets_dev == nullptr || shift < 0 || shift > 58) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (workspace_bytes < partition_workspace_bytes(n, kWave) || workspace_dev == nullptr) return QSX_ERR_CAPACITY;
  ScatterArgs args;
  args.ncols = ncols;
  for (int c = 0; c < ncols; ++c) {
    args.width[c] = widths[c];
    args.src[c] = cols[c];
    args.dst[c] = out_cols[c];
  }
  PackedKey k;
  k.num = num_keys;
  for (int i = 0; i < QSX_MAX_KEYS; ++i) {
    k.col[i] = i < num_keys ? key_cols[i] : nullptr;
    k.width[i] = i < num_keys ? key_widths[i] : 0;
    k.shift[i] = i < num_keys ? key_shifts[i] : 0;
  }
  if (!stable) return launch_partition_t<PackedKey, 4>(k, n, kWave, shift, args, out_offsets_dev, workspace_dev, 0, s);
  return launch_partition_t<PackedKey, 3>(k, n, kWave, shift, args, out_offsets_dev, workspace_dev, 0, s);
}

}  // namespace qsx

using namespace qsx;

extern "C" {

size_t qsx_partition_workspace_bytes(int64_t n, int num_partitions) { return partition_workspace_bytes(n, num_partitions); }

int qsx_partition_scatter(int key_type, const void *keys_dev, int64_t n, int num_partitions, int ncols,
                          const void *const *cols, const int32_t *widths, void *const *out_cols,
                          int64_t *out_offsets_dev, void *workspace_dev, size_t workspace_bytes,
                          qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  return partition_scatter_impl(0, key_type, keys_dev, n, num_partitions, ncols, cols, widths, out_cols, out_offsets_dev,
                                workspace_dev, workspace_bytes, as_stream(stream), 0);
}

size_t qsx_partition_blocks_workspace_bytes(int64_t n, int64_t num_blocks, int num_partitions) {
  return partition_blocks_workspace_bytes(n, num_blocks < 0 ? 0 : num_blocks, num_partitions);
}

// K9 over a run of blocks: the result of qsx_partition_scatter over the blocks' rows laid end to end, without laying them so.
int qsx_partition_scatter_blocks(int key_type, int64_t num_blocks, const int64_t *block_rows, const void *const *block_keys,
                                 int num_partitions, int ncols, const void *const *block_cols, const int32_t *widths,
                                 void *const *out_cols, int64_t *out_offsets_dev, void *workspace_dev, size_t workspace_bytes,
                                 qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (num_blocks < 0 || num_partitions < 1 || ncols < 0 || ncols > QSX_MAX_COLUMNS || out_offsets_dev == nullptr ||
      (num_blocks > 0 && (block_rows == nullptr || block_keys == nullptr)) ||
      (ncols > 0 && (widths == nullptr || out_cols == nullptr || (num_blocks > 0 && block_cols == nullptr)))) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (num_partitions > kWave) return QSX_ERR_UNSUPPORTED;
  if (key_type != QSX_INT && key_type != QSX_LONG) return QSX_ERR_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
  ScatterArgs args;
  args.ncols = ncols;
  for (int c = 0; c < ncols; ++c) {
    if (widths[c] != 1 && widths[c] != 2 && widths[c] != 4 && widths[c] != 8) return QSX_ERR_UNSUPPORTED;
    args.width[c] = widths[c];
    args.src[c] = nullptr;
    args.dst[c] = out_cols[c];
  }
  // the blocks that hold rows
  std::vector<int64_t> rows;
  std::vector<const void *> keys;
  std::vector<int64_t> which;
  int64_t n = 0;
  for (int64_t b = 0; b < num_blocks; ++b) {
    if (block_rows[b] < 0) return QSX_ERR_INVALID_ARGUMENT;
    if (block_rows[b] == 0) continue;
    if (block_keys[b] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
    for (int c = 0; c < ncols; ++c) {
      if (block_cols[b * ncols + c] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
    }
    rows.push_back(block_rows[b]);
    keys.push_back(block_keys[b]);
    which.push_back(b);
    n += block_rows[b];
  }
  if (n == 0) {
    QSX_HIP_TRY(hipMemsetAsync(out_offsets_dev, 0, sizeof(int64_t) * (num_partitions + 1), s));
    return QSX_OK;
  }
  const int64_t nb = static_cast<int64_t>(rows.size());
  if (workspace_dev == nullptr || workspace_bytes < partition_blocks_workspace_bytes(n, num_blocks, num_partitions)) return QSX_ERR_CAPACITY;
  for (int c = 0; c < ncols; ++c) {
    if (out_cols[c] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  }
  const int64_t chunk_rows = chunk_rows_for(n);
  std::vector<long long> table;
  const long long chunks = build_run_table(chunk_rows, nb, rows.data(), keys.data(), nullptr, nullptr, nullptr, &table);
  if (chunks <= 0) return QSX_ERR_INVALID_ARGUMENT;
  for (int c = 0; c < ncols; ++c) {   // (ChunkSource::cols_at: right behind the table's five arrays)
    for (int64_t i = 0; i < nb; ++i) table.push_back(static_cast<long long>(reinterpret_cast<uintptr_t>(block_cols[which[i] * ncols + c])));
  }
  const size_t bytes = table.size() * sizeof(long long);
  const long long *runs_dev = static_cast<const long long *>(staged_device_buffer(s, bytes));
  if (runs_dev == nullptr) return QSX_ERR_OUT_OF_MEMORY;
  const int rc = staged_upload(s, table.data(), bytes);
  if (rc != QSX_OK) return rc;
  const bool is_pow2 = (num_partitions & (num_partitions - 1)) == 0;
  if (key_type == QSX_INT) {
    return launch_partition_t<ColumnKey<int32_t>, 0>(ColumnKey<int32_t>{nullptr}, n, num_partitions, is_pow2 ? 1 : 0, args, out_offsets_dev,
                                                     workspace_dev, 0, s, runs_dev, chunks, chunk_rows);
  }
  return launch_partition_t<ColumnKey<int64_t>, 0>(ColumnKey<int64_t>{nullptr}, n, num_partitions, is_pow2 ? 1 : 0, args, out_offsets_dev,
                                                   workspace_dev, 0, s, runs_dev, chunks, chunk_rows);
}

}  // extern "C"
