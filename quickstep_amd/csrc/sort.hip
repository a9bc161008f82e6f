// sort.hip — ORDER BY: stable sort permutation over up to QSX_MAX_KEYS key columns (+ top-k by truncation).
//
// Reference (paths in the Quickstep tree): SortRunGenerationWorkOrder::execute
// (relational_operators/SortRunGenerationOperator.cpp:88-105) -> StorageBlock::sort (storage/StorageBlock.cpp:561-640)
// with the comparator chain of utility/SortConfiguration.hpp:51-130, and the run merge with top_k of
// relational_operators/SortMergeRunOperatorHelpers.cpp.  The device does not compare tuples: every key column is
// turned into an order-preserving unsigned 64-bit image (sign bit flipped for integers, IEEE sign-magnitude folded
// for FLOAT/DOUBLE, all bits inverted for DESC) and the rows are LSD-radix-sorted — one stable pass of the K9
// scatter per 6-bit digit — key by key from the least significant ORDER BY column to the most significant one.
// A merge of sorted runs is the same sort over their concatenation (a radix pass costs what a merge pass costs).

#include "common.hpp"
#include "partition.hpp"

namespace qsx {

constexpr int kSBlock = 256;

__global__ __launch_bounds__(kSBlock) void iota_kernel(int32_t *__restrict__ out, int64_t n) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kSBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kSBlock) {
    out[i] = static_cast<int32_t>(i);
  }
}

// keys64[i] = ordered image of col[tids[i]]
template <typename T>
__global__ __launch_bounds__(kSBlock) void sort_keys_kernel(const T *__restrict__ col, const int32_t *__restrict__ tids, int64_t n,
                                                           int type, int descending, unsigned long long *__restrict__ keys64) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kSBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kSBlock) {
    const T b = col[tids[i]];      // raw bits of the value (T = uint32_t for INT / FLOAT, uint64 for LONG / DOUBLE)
    constexpr T kSign = static_cast<T>(1) << (sizeof(T) * 8 - 1);
    unsigned long long k;
    if (type == QSX_INT || type == QSX_LONG) {
      k = b ^ kSign;                                   // two's complement: flip the sign bit
    } else {
      const T z = (b & static_cast<T>(~kSign)) == 0 ? static_cast<T>(0) : b;   // -0.0 compares equal to +0.0: same image
      k = (z & kSign) ? static_cast<T>(~z) : (z | kSign);   // IEEE sign-magnitude: negatives reversed below the positives
    }
    if (descending) k = sizeof(T) == 4 ? (static_cast<uint32_t>(~k)) : ~k;
    keys64[i] = k;
  }
}

static size_t s_align(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace qsx

using namespace qsx;

extern "C" {

size_t qsx_sort_workspace_bytes(int64_t n) {
  if (n < 0) n = 0;
  const size_t rows = static_cast<size_t>(n) + 16;
  return 2 * s_align(rows * 8) + s_align(rows * 4) + s_align(65 * 8) + partition_workspace_bytes(n, kWave) + 256;
}

int qsx_sort_permutation(int nkeys, const void *const *key_cols, const int32_t *key_types, const int32_t *descending, int64_t n,
                         int32_t *out_tids_dev, void *workspace_dev, size_t workspace_bytes, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  if (nkeys < 1 || nkeys > QSX_MAX_KEYS || key_cols == nullptr || key_types == nullptr || n < 0 || n > INT32_MAX ||
      (n > 0 && out_tids_dev == nullptr)) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  for (int k = 0; k < nkeys; ++k) {
    if (key_types[k] < QSX_INT || key_types[k] > QSX_DOUBLE) return QSX_ERR_UNSUPPORTED;
    if (n > 0 && key_cols[k] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  }
  if (n == 0) return QSX_OK;
  if (workspace_dev == nullptr || workspace_bytes < qsx_sort_workspace_bytes(n)) return QSX_ERR_CAPACITY;
  hipStream_t s = as_stream(stream);
  const size_t rows = static_cast<size_t>(n) + 16;
  char *w = static_cast<char *>(workspace_dev);
  unsigned long long *keys_a = reinterpret_cast<unsigned long long *>(w);
  unsigned long long *keys_b = reinterpret_cast<unsigned long long *>(w + s_align(rows * 8));
  int32_t *tids_b = reinterpret_cast<int32_t *>(w + 2 * s_align(rows * 8));
  int64_t *offsets = reinterpret_cast<int64_t *>(w + 2 * s_align(rows * 8) + s_align(rows * 4));
  void *part_ws = w + 2 * s_align(rows * 8) + s_align(rows * 4) + s_align(65 * 8);
  const size_t part_ws_bytes = partition_workspace_bytes(n, kWave);
  int32_t *tids_a = out_tids_dev;
  const int grid = grid_for(n, kSBlock * 4);
  hipLaunchKernelGGL(iota_kernel, dim3(grid), dim3(kSBlock), 0, s, tids_a, n);
  QSX_CHECK_LAUNCH();
  for (int k = nkeys - 1; k >= 0; --k) {
    const int type = key_types[k];
    const int desc = descending != nullptr && descending[k] != 0 ? 1 : 0;
    if (type == QSX_INT || type == QSX_FLOAT) {
      hipLaunchKernelGGL(sort_keys_kernel<uint32_t>, dim3(grid), dim3(kSBlock), 0, s, static_cast<const uint32_t *>(key_cols[k]), tids_a, n,
                         type, desc, keys_a);
    } else {
      hipLaunchKernelGGL(sort_keys_kernel<unsigned long long>, dim3(grid), dim3(kSBlock), 0, s,
                         static_cast<const unsigned long long *>(key_cols[k]), tids_a, n, type, desc, keys_a);
    }
    QSX_CHECK_LAUNCH();
    const int bits = (type == QSX_INT || type == QSX_FLOAT) ? 32 : 64;
    for (int shift = 0; shift < bits; shift += 6) {
      const void *src[2] = {keys_a, tids_a};
      void *dst[2] = {keys_b, tids_b};
      const int32_t widths[2] = {8, 4};
      int rc = partition_scatter_digit(keys_a, n, shift, 2, src, widths, dst, offsets, part_ws, part_ws_bytes, s);
      if (rc != QSX_OK) return rc;
      unsigned long long *tk = keys_a; keys_a = keys_b; keys_b = tk;
      int32_t *tt = tids_a; tids_a = tids_b; tids_b = tt;
    }
  }
  if (tids_a != out_tids_dev) {
    QSX_HIP_TRY(hipMemcpyAsync(out_tids_dev, tids_a, static_cast<size_t>(n) * 4, hipMemcpyDeviceToDevice, s));
  }
  return QSX_OK;
}

}  // extern "C"
