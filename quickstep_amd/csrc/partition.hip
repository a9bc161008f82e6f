// partition.hip — K9: stable hash-partition scatter (first half of the
// multi-GPU join-key shuffle; also the partitioned-aggregation router).
//
// Partition id = HashPartitionSchemeHeader::getPartitionId
// (catalog/PartitionSchemeHeader.hpp:200-214) over the identity hash of the
// key's zero-extended bit pattern (types/TypedValue.hpp:575-592); the routing
// itself is what PartitionAwareInsertDestination does per output row
// (storage/InsertDestination.hpp:490-660).
//
// Three passes, every wave owns one contiguous row range so that the scatter
// is stable (rows keep their relative order inside a partition):
//   1. per-wave histograms, written partition-major  hist[p * W + w]
//   2. exclusive scan of that array  ->  absolute start of (partition p, wave w)
//   3. scatter: per 64-row batch the wave walks the partition ids present
//      (readfirstlane loop), ranks the rows of one partition with ballot +
//      mbcnt and bumps that partition's running offset, held by lane p.
// Limits: at most 64 partitions (lane p owns partition p).

#include "common.hpp"
#include "scan.hpp"

namespace qsx {

constexpr int kPBlock = 256;
constexpr int kPWaves = kPBlock / kWave;

template <typename KeyT>
__device__ __forceinline__ int partition_of(KeyT key, int P, int pow2) {
  unsigned long long h;
  if (sizeof(KeyT) == 4) h = static_cast<uint32_t>(key); else h = static_cast<unsigned long long>(key);
  if (pow2) return static_cast<int>(h & static_cast<unsigned long long>(P - 1));
  return static_cast<int>(h >= static_cast<unsigned long long>(P) ? h % static_cast<unsigned long long>(P) : h);
}

struct ScatterArgs {
  int ncols;
  int width[QSX_MAX_COLUMNS];
  const void *src[QSX_MAX_COLUMNS];
  void *dst[QSX_MAX_COLUMNS];
};

template <typename KeyT>
__global__ __launch_bounds__(kPBlock) void partition_hist_kernel(const KeyT *__restrict__ keys, int64_t n, int P,
                                                                int pow2, int64_t rows_per_wave, int64_t W,
                                                                int32_t *__restrict__ hist) {
  const int lane = lane_id();
  const int64_t w = static_cast<int64_t>(blockIdx.x) * kPWaves + (threadIdx.x >> 6);
  if (w >= W) return;
  const int64_t begin = w * rows_per_wave;
  const int64_t end = begin + rows_per_wave < n ? begin + rows_per_wave : n;
  int my_count = 0;  // lane p counts partition p
  for (int64_t base = begin; base < end; base += kWave) {
    const int64_t row = base + lane;
    const int pid = row < end ? partition_of<KeyT>(keys[row], P, pow2) : -1;
    uint64_t remaining = __ballot(pid >= 0);
    while (remaining != 0) {
      const int leader = __ffsll(static_cast<long long>(remaining)) - 1;
      const int cur = __shfl(pid, leader, kWave);
      const uint64_t m = __ballot(pid == cur);
      if (lane == cur) my_count += __popcll(m);
      remaining &= ~m;
    }
  }
  if (lane < P) hist[static_cast<int64_t>(lane) * W + w] = my_count;
}

__device__ __forceinline__ void move_value(const void *src, int64_t si, void *dst, int64_t di, int width) {
  switch (width) {
    case 1: static_cast<uint8_t *>(dst)[di] = static_cast<const uint8_t *>(src)[si]; break;
    case 2: static_cast<uint16_t *>(dst)[di] = static_cast<const uint16_t *>(src)[si]; break;
    case 4: static_cast<uint32_t *>(dst)[di] = static_cast<const uint32_t *>(src)[si]; break;
    default: static_cast<uint64_t *>(dst)[di] = static_cast<const uint64_t *>(src)[si]; break;
  }
}

template <typename KeyT>
__global__ __launch_bounds__(kPBlock) void partition_scatter_kernel(const KeyT *__restrict__ keys, int64_t n, int P,
                                                                   int pow2, int64_t rows_per_wave, int64_t W,
                                                                   const int64_t *__restrict__ starts,
                                                                   ScatterArgs args,
                                                                   int64_t *__restrict__ out_offsets) {
  const int lane = lane_id();
  const int64_t w = static_cast<int64_t>(blockIdx.x) * kPWaves + (threadIdx.x >> 6);
  if (w >= W) return;
  if (w == 0) {
    if (lane < P) out_offsets[lane] = starts[static_cast<int64_t>(lane) * W];
    if (lane == 0) out_offsets[P] = n;
  }
  const int64_t begin = w * rows_per_wave;
  const int64_t end = begin + rows_per_wave < n ? begin + rows_per_wave : n;
  int64_t my_offset = lane < P ? starts[static_cast<int64_t>(lane) * W + w] : 0;  // lane p: next slot of partition p
  for (int64_t base = begin; base < end; base += kWave) {
    const int64_t row = base + lane;
    const int pid = row < end ? partition_of<KeyT>(keys[row], P, pow2) : -1;
    uint64_t remaining = __ballot(pid >= 0);
    int64_t dst_row = -1;
    while (remaining != 0) {
      const int leader = __ffsll(static_cast<long long>(remaining)) - 1;
      const int cur = __shfl(pid, leader, kWave);
      const uint64_t m = __ballot(pid == cur);
      const int64_t part_base = __shfl(my_offset, cur, kWave);
      if (pid == cur) dst_row = part_base + rank_below(m);
      if (lane == cur) my_offset += __popcll(m);
      remaining &= ~m;
    }
    if (dst_row >= 0) {
      for (int c = 0; c < args.ncols; ++c) move_value(args.src[c], row, args.dst[c], dst_row, args.width[c]);
    }
  }
}

static size_t p_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int64_t waves_for(int64_t n) {
  // at least 512 rows per wave, at most 8192 waves (2048 workgroups)
  int64_t W = (n + 511) / 512;
  if (W < 1) W = 1;
  if (W > static_cast<int64_t>(kMaxGridBlocks) * kPWaves) W = static_cast<int64_t>(kMaxGridBlocks) * kPWaves;
  return W;
}

}  // namespace qsx

using namespace qsx;

extern "C" {

size_t qsx_partition_workspace_bytes(int64_t n, int num_partitions) {
  const int64_t W = waves_for(n);
  const int64_t cells = W * num_partitions;
  return p_align_up(sizeof(int64_t) * (cells + 1), 256) + p_align_up(sizeof(int32_t) * cells, 256);
}

int qsx_partition_scatter(int key_type, const void *keys_dev, int64_t n, int num_partitions, int ncols,
                          const void *const *cols, const int32_t *widths, void *const *out_cols,
                          int64_t *out_offsets_dev, void *workspace_dev, size_t workspace_bytes,
                          qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (n < 0 || num_partitions < 1 || ncols < 0 || ncols > QSX_MAX_COLUMNS || out_offsets_dev == nullptr) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  if (num_partitions > kWave) return QSX_ERR_UNSUPPORTED;
  if (key_type != QSX_INT && key_type != QSX_LONG) return QSX_ERR_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
  if (n == 0) {
    QSX_HIP_TRY(hipMemsetAsync(out_offsets_dev, 0, sizeof(int64_t) * (num_partitions + 1), s));
    return QSX_OK;
  }
  if (workspace_bytes < qsx_partition_workspace_bytes(n, num_partitions) || workspace_dev == nullptr) return QSX_ERR_CAPACITY;
  ScatterArgs args;
  args.ncols = ncols;
  for (int c = 0; c < ncols; ++c) {
    const int w = widths[c];
    if (w != 1 && w != 2 && w != 4 && w != 8) return QSX_ERR_UNSUPPORTED;
    args.width[c] = w;
    args.src[c] = cols[c];
    args.dst[c] = out_cols[c];
  }
  const int64_t W = waves_for(n);
  int64_t rows_per_wave = (n + W - 1) / W;
  rows_per_wave = (rows_per_wave + kWave - 1) / kWave * kWave;
  const int64_t cells = W * num_partitions;
  int64_t *starts = static_cast<int64_t *>(workspace_dev);
  int32_t *hist = reinterpret_cast<int32_t *>(static_cast<char *>(workspace_dev) +
                                              p_align_up(sizeof(int64_t) * (cells + 1), 256));
  const int pow2 = (num_partitions & (num_partitions - 1)) == 0 ? 1 : 0;
  const int grid = static_cast<int>((W + kPWaves - 1) / kPWaves);
  if (key_type == QSX_INT) {
    hipLaunchKernelGGL(partition_hist_kernel<int32_t>, dim3(grid), dim3(kPBlock), 0, s,
                       static_cast<const int32_t *>(keys_dev), n, num_partitions, pow2, rows_per_wave, W, hist);
  } else {
    hipLaunchKernelGGL(partition_hist_kernel<int64_t>, dim3(grid), dim3(kPBlock), 0, s,
                       static_cast<const int64_t *>(keys_dev), n, num_partitions, pow2, rows_per_wave, W, hist);
  }
  QSX_CHECK_LAUNCH();
  hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, hist, cells, starts, static_cast<int64_t *>(nullptr));
  QSX_CHECK_LAUNCH();
  if (key_type == QSX_INT) {
    hipLaunchKernelGGL(partition_scatter_kernel<int32_t>, dim3(grid), dim3(kPBlock), 0, s,
                       static_cast<const int32_t *>(keys_dev), n, num_partitions, pow2, rows_per_wave, W, starts, args,
                       out_offsets_dev);
  } else {
    hipLaunchKernelGGL(partition_scatter_kernel<int64_t>, dim3(grid), dim3(kPBlock), 0, s,
                       static_cast<const int64_t *>(keys_dev), n, num_partitions, pow2, rows_per_wave, W, starts, args,
                       out_offsets_dev);
  }
  QSX_CHECK_LAUNCH();
  return QSX_OK;
}

}  // extern "C"
