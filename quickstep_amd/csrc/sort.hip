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

#include <algorithm>
#include "partition.hpp"

namespace qsx {

constexpr int kSBlock = 256;

__global__ __launch_bounds__(kSBlock) void iota_kernel(int32_t *__restrict__ out, int64_t n) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kSBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kSBlock) {
    out[i] = static_cast<int32_t>(i);
  }
}

// Order-preserving unsigned image of a value given as raw bits (T = uint32_t for INT / FLOAT, uint64 for LONG / DOUBLE).
template <typename T>
__device__ __forceinline__ unsigned long long ordered_image(T b, int type, int descending) {
  constexpr T kSign = static_cast<T>(1) << (sizeof(T) * 8 - 1);
  unsigned long long k;
  if (type == QSX_CHAR) {
    k = b;                                           // unsigned byte
  } else if (type == QSX_DATE) {
    k = static_cast<unsigned long long>(date_ordered(static_cast<unsigned long long>(b))) ^ (1ull << 63);   // year, month, day
  } else if (type == QSX_INT || type == QSX_LONG) {
    k = b ^ kSign;                                   // two's complement: flip the sign bit
  } else {
    const T z = (b & static_cast<T>(~kSign)) == 0 ? static_cast<T>(0) : b;   // -0.0 compares equal to +0.0: same image
    k = (z & kSign) ? static_cast<T>(~z) : (z | kSign);   // IEEE sign-magnitude: negatives reversed below the positives
  }
  if (descending) k = sizeof(T) == 8 ? ~k : (~k & ((1ull << (sizeof(T) * 8)) - 1));
  return k;
}

// keys64[i] = ordered image of col[tids[i]]
template <typename T>
__global__ __launch_bounds__(kSBlock) void sort_keys_kernel(const T *__restrict__ col, const int32_t *__restrict__ tids, int64_t n,
                                                           int type, int descending, unsigned long long *__restrict__ keys64) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kSBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kSBlock) {
    keys64[i] = ordered_image<T>(col[tids[i]], type, descending);
  }
}

// Up to kSmallSort (key image, row number) pairs sorted by ONE workgroup in LDS — a bitonic network on (key, position), so
// equal keys keep their order — instead of 6-11 radix passes of three launches each: the candidates of an ORDER BY ... LIMIT k
// (qsx_sort_top_k: k rows and the population of one histogram bin), the runs of a few hundred groups of a finalize.
// out[i] = tids[position of the i-th smallest key].
constexpr int kSmallSort = 2048;
__global__ __launch_bounds__(1024) void small_sort_kernel(const unsigned long long *__restrict__ keys, const int32_t *__restrict__ tids, int n,
                                                         int32_t *__restrict__ out) {
  __shared__ unsigned long long s_key[kSmallSort];
  __shared__ unsigned short s_pos[kSmallSort];
  for (int i = threadIdx.x; i < kSmallSort; i += 1024) {
    s_key[i] = i < n ? keys[i] : ~0ull;     // (padding sorts behind every real pair: an all-ones key of a real row has the smaller position)
    s_pos[i] = static_cast<unsigned short>(i);
  }
  __syncthreads();
  const int t = threadIdx.x;
  for (int k = 2; k <= kSmallSort; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo + j;
      const bool ascending = (lo & k) == 0;
      const unsigned long long ka = s_key[lo], kb = s_key[hi];
      const unsigned short pa = s_pos[lo], pb = s_pos[hi];
      const bool a_behind_b = ka > kb || (ka == kb && pa > pb);
      if (a_behind_b == ascending) {
        s_key[lo] = kb; s_key[hi] = ka;
        s_pos[lo] = pb; s_pos[hi] = pa;
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += 1024) out[i] = tids[s_pos[i]];
}

// ---- top-k: threshold selection on the most significant key -------------------------------------------------------
// LIMIT k after ORDER BY needs the k first rows only.  A 4096-bin histogram of the leading 12 bits of the first key's
// image locates the bin in which the k-th row falls; the rows up to that bin (k + one bin's population, in input
// order) are the only candidates, and only they are sorted.  One pass over the key column + a pass writing a bitmap,
// instead of 6-11 scatter passes over all rows per key.
constexpr int kTopBins = 4096;

// Histogram of the next `bits` bits (below the `prefix_bits` leading bits already fixed) of key 0's image, over the
// rows whose leading bits equal `prefix`.
template <typename T>
__global__ __launch_bounds__(kSBlock) void topk_hist_kernel(const T *__restrict__ col, int64_t n, int type, int descending,
                                                           int prefix_bits, unsigned long long prefix, int bits,
                                                           unsigned long long *__restrict__ hist) {
  __shared__ unsigned int s_hist[kTopBins];
  for (int i = threadIdx.x; i < kTopBins; i += kSBlock) s_hist[i] = 0;
  __syncthreads();
  constexpr int kWidth = static_cast<int>(sizeof(T)) * 8;
  const int shift = kWidth - prefix_bits - bits;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kSBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kSBlock) {
    const unsigned long long k = ordered_image<T>(col[i], type, descending);
    const bool in_prefix = prefix_bits == 0 || (k >> (kWidth - prefix_bits)) == prefix;
    if (in_prefix) atomicAdd(&s_hist[(k >> shift) & ((1u << bits) - 1u)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kTopBins; i += kSBlock) {
    if (s_hist[i] != 0) atomicAdd(&hist[i], static_cast<unsigned long long>(s_hist[i]));
  }
}

// `below` rows sort before every row of the prefix.  control[0] = threshold bin t (smallest with below + count(bins <= t)
// >= k), control[1] = below + count(bins <= t) (the candidates when the selection stops here), control[2] = below +
// count(bins < t) (the rows in front of bin t: `below` of the next level).  One wave: lane L sums bins [64 L, 64 L + 64),
// a wave scan finds the lane whose range holds the k-th row, that lane walks its 64 bins.
__global__ __launch_bounds__(kWave) void topk_threshold_kernel(const unsigned long long *__restrict__ hist, long long k,
                                                              long long below, long long *__restrict__ control) {
  constexpr int kPerLane = kTopBins / kWave;
  const int lane = lane_id();
  unsigned long long mine = 0;
#pragma unroll 8
  for (int i = 0; i < kPerLane; ++i) mine += hist[lane * kPerLane + i];
  unsigned long long incl = mine;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned long long up = __shfl_up(incl, off, kWave);
    if (lane >= off) incl += up;
  }
  const unsigned long long want = static_cast<unsigned long long>(k - below);   // >= 1: the k-th row lies in this prefix
  const unsigned long long before = incl - mine;
  const uint64_t reached = __ballot(incl >= want);
  if (reached == 0) {   // cannot happen for k <= n; keep everything
    if (lane == kWave - 1) { control[0] = kTopBins - 1; control[1] = below + static_cast<long long>(incl); control[2] = below; }
    return;
  }
  if (lane == __ffsll(static_cast<long long>(reached)) - 1) {
    unsigned long long cum = before;
    int t = lane * kPerLane;
    for (; t < (lane + 1) * kPerLane; ++t) {
      if (cum + hist[t] >= want) break;
      cum += hist[t];
    }
    control[0] = t;
    control[1] = below + static_cast<long long>(cum + hist[t]);
    control[2] = below + static_cast<long long>(cum);
  }
}

// keep = the leading `bits` bits of the image are <= limit
template <typename T>
__global__ __launch_bounds__(kSBlock) void topk_mark_kernel(const T *__restrict__ col, int64_t n, int type, int descending,
                                                           int bits, unsigned long long limit, uint64_t *__restrict__ bitmap) {
  const int64_t num_words = (n + 63) >> 6;
  const int lane = lane_id();
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * (kSBlock / kWave) + (threadIdx.x >> 6); w < num_words;
       w += static_cast<int64_t>(gridDim.x) * (kSBlock / kWave)) {
    const int64_t row = (w << 6) + lane;
    bool keep = false;
    if (row < n) keep = (ordered_image<T>(col[row], type, descending) >> (sizeof(T) * 8 - bits)) <= limit;
    const uint64_t word = msb_first(__ballot(keep));
    if (lane == 0) bitmap[w] = word;
  }
}

static size_t s_align(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace qsx

using namespace qsx;

// ---- DISTINCT: first row of every run of equal tuples in the sorted order ---------------------------------------
struct TupleColumns {
  int ncols;
  const void *col[QSX_MAX_KEYS];
  int type[QSX_MAX_KEYS];
};

__device__ __forceinline__ unsigned long long tuple_image(const TupleColumns &t, int c, int32_t row) {
  const int type = t.type[c];
  if (type == QSX_CHAR) return static_cast<const uint8_t *>(t.col[c])[row];
  if (type == QSX_INT || type == QSX_FLOAT) return ordered_image<uint32_t>(static_cast<const uint32_t *>(t.col[c])[row], type, 0);
  return ordered_image<unsigned long long>(static_cast<const unsigned long long *>(t.col[c])[row], type, 0);
}

// bit i = tuple of row tids[i] differs from the tuple of row tids[i-1] (bit 0 always set); equality is the sort's
// (the ordered images: -0.0 == +0.0)
__global__ __launch_bounds__(kSBlock) void run_heads_kernel(TupleColumns t, const int32_t *__restrict__ tids, int64_t m,
                                                           uint64_t *__restrict__ bitmap) {
  const int64_t num_words = (m + 63) >> 6;
  const int lane = lane_id();
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * (kSBlock / kWave) + (threadIdx.x >> 6); w < num_words;
       w += static_cast<int64_t>(gridDim.x) * (kSBlock / kWave)) {
    const int64_t i = (w << 6) + lane;
    bool head = false;
    if (i < m) {
      head = i == 0;
      if (!head) {
        const int32_t row = tids[i], prev = tids[i - 1];
        for (int c = 0; c < t.ncols; ++c) head = head || tuple_image(t, c, row) != tuple_image(t, c, prev);
      }
    }
    const uint64_t word = msb_first(__ballot(head));
    if (lane == 0) bitmap[w] = word;
  }
}

// Sizes of the pieces of the sort workspace for n rows.
struct SortWorkspace {
  size_t keys, tids, offsets, part, hist, bitmap, compact, total;
  explicit SortWorkspace(int64_t n) {
    if (n < 0) n = 0;
    const size_t rows = static_cast<size_t>(n) + 16;
    keys = s_align(rows * 8);
    tids = s_align(rows * 4);
    offsets = s_align(65 * 8);
    part = s_align(partition_workspace_bytes(n, kWave));
    hist = s_align(kTopBins * 8 + 64);
    bitmap = s_align(((n + 63) / 64 + 2) * 8);
    compact = s_align(qsx_compact_workspace_bytes(n) + 64);
    total = 2 * keys + 2 * tids + offsets + part + hist + bitmap + compact + 256;
  }
};

static int validate_sort_args(int nkeys, const void *const *key_cols, const int32_t *key_types, int64_t n, const void *out) {
  if (nkeys < 1 || nkeys > QSX_MAX_KEYS || key_cols == nullptr || key_types == nullptr || n < 0 || n > INT32_MAX ||
      (n > 0 && out == nullptr)) {
    return QSX_ERR_INVALID_ARGUMENT;
  }
  for (int k = 0; k < nkeys; ++k) {
    if ((key_types[k] < QSX_INT || key_types[k] > QSX_CHAR) && key_types[k] != QSX_DATE) return QSX_ERR_UNSUPPORTED;
    if (n > 0 && key_cols[k] == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  }
  return QSX_OK;
}

// Stable sort of the m row numbers in tids_a (in place semantically: the result lands in *result, one of the two tid
// buffers) by the keys; tids_a must hold the rows in input order.
static int sort_tids(int nkeys, const void *const *key_cols, const int32_t *key_types, const int32_t *descending, int64_t m,
                     int32_t *tids_a, int32_t *tids_b, char *w, const SortWorkspace &ws, int32_t **result, hipStream_t s) {
  unsigned long long *keys_a = reinterpret_cast<unsigned long long *>(w);
  unsigned long long *keys_b = reinterpret_cast<unsigned long long *>(w + ws.keys);
  int64_t *offsets = reinterpret_cast<int64_t *>(w + 2 * ws.keys + 2 * ws.tids);
  void *part_ws = w + 2 * ws.keys + 2 * ws.tids + ws.offsets;
  const size_t part_ws_bytes = ws.part;
  const int64_t n = m;
  const int grid = grid_for(n, kSBlock * 4);
  for (int k = nkeys - 1; k >= 0; --k) {
    const int type = key_types[k];
    const int desc = descending != nullptr && descending[k] != 0 ? 1 : 0;
    if (type == QSX_CHAR) {
      hipLaunchKernelGGL(sort_keys_kernel<uint8_t>, dim3(grid), dim3(kSBlock), 0, s, static_cast<const uint8_t *>(key_cols[k]), tids_a, n,
                         type, desc, keys_a);
    } else if (type == QSX_INT || type == QSX_FLOAT) {
      hipLaunchKernelGGL(sort_keys_kernel<uint32_t>, dim3(grid), dim3(kSBlock), 0, s, static_cast<const uint32_t *>(key_cols[k]), tids_a, n,
                         type, desc, keys_a);
    } else {
      hipLaunchKernelGGL(sort_keys_kernel<unsigned long long>, dim3(grid), dim3(kSBlock), 0, s,
                         static_cast<const unsigned long long *>(key_cols[k]), tids_a, n, type, desc, keys_a);
    }
    QSX_CHECK_LAUNCH();
    if (n <= kSmallSort) {   // one workgroup, one launch
      hipLaunchKernelGGL(small_sort_kernel, dim3(1), dim3(1024), 0, s, keys_a, tids_a, static_cast<int>(n), tids_b);
      QSX_CHECK_LAUNCH();
      int32_t *tt = tids_a; tids_a = tids_b; tids_b = tt;
      continue;
    }
    const int bits = type == QSX_CHAR ? 8 : ((type == QSX_INT || type == QSX_FLOAT) ? 32 : 64);
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
  *result = tids_a;
  return QSX_OK;
}

extern "C" {

size_t qsx_sort_workspace_bytes(int64_t n) { return SortWorkspace(n).total; }

int qsx_sort_permutation(int nkeys, const void *const *key_cols, const int32_t *key_types, const int32_t *descending, int64_t n,
                         int32_t *out_tids_dev, void *workspace_dev, size_t workspace_bytes, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  int rc = validate_sort_args(nkeys, key_cols, key_types, n, out_tids_dev);
  if (rc != QSX_OK) return rc;
  if (n == 0) return QSX_OK;
  const SortWorkspace ws(n);
  if (workspace_dev == nullptr || workspace_bytes < ws.total) return QSX_ERR_CAPACITY;
  hipStream_t s = as_stream(stream);
  char *w = static_cast<char *>(workspace_dev);
  int32_t *tids_b = reinterpret_cast<int32_t *>(w + 2 * ws.keys);
  hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n, kSBlock * 4)), dim3(kSBlock), 0, s, out_tids_dev, n);
  QSX_CHECK_LAUNCH();
  int32_t *result = nullptr;
  rc = sort_tids(nkeys, key_cols, key_types, descending, n, out_tids_dev, tids_b, w, ws, &result, s);
  if (rc != QSX_OK) return rc;
  if (result != out_tids_dev) QSX_HIP_TRY(hipMemcpyAsync(out_tids_dev, result, static_cast<size_t>(n) * 4, hipMemcpyDeviceToDevice, s));
  return QSX_OK;
}

int qsx_sort_top_k(int nkeys, const void *const *key_cols, const int32_t *key_types, const int32_t *descending, int64_t n,
                   int64_t k, int32_t *out_tids_dev, void *workspace_dev, size_t workspace_bytes, qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  int rc = validate_sort_args(nkeys, key_cols, key_types, n, out_tids_dev);
  if (rc != QSX_OK) return rc;
  if (k < 0) return QSX_ERR_INVALID_ARGUMENT;
  if (k > n) k = n;
  if (n == 0 || k == 0) return QSX_OK;
  const SortWorkspace ws(n);
  if (workspace_dev == nullptr || workspace_bytes < ws.total) return QSX_ERR_CAPACITY;
  hipStream_t s = as_stream(stream);
  char *w = static_cast<char *>(workspace_dev);
  int32_t *tids_a = reinterpret_cast<int32_t *>(w + 2 * ws.keys + ws.tids);   // second tid buffer: candidates / iota
  int32_t *tids_b = reinterpret_cast<int32_t *>(w + 2 * ws.keys);
  char *extra = w + 2 * ws.keys + 2 * ws.tids + ws.offsets + ws.part;
  unsigned long long *hist = reinterpret_cast<unsigned long long *>(extra);
  long long *control = reinterpret_cast<long long *>(extra + kTopBins * 8);   // inside the hist piece (64 spare bytes)
  uint64_t *bitmap = reinterpret_cast<uint64_t *>(extra + ws.hist);
  void *compact_ws = extra + ws.hist + ws.bitmap;
  int64_t m = n;
  bool selected = false;
  if (n >= 65536 && k <= n / 64 && key_types[0] != QSX_CHAR) {
    // threshold selection on key 0, refined 12 bits at a time while the threshold bin still holds too many rows
    // (keys that share their leading bits: doubles in [0, 1), small integers)
    const int type = key_types[0];
    const int desc = descending != nullptr && descending[0] != 0 ? 1 : 0;
    const bool narrow = type == QSX_INT || type == QSX_FLOAT;
    const int width = narrow ? 32 : 64;
    const int grid = grid_for(n, kSBlock * 8);
    // (candidates that fit one workgroup's LDS sort are worth another histogram level — 60 us — against the radix passes'
    // 11 x 3 launches; a bin of very many equal keys ends the refinement at 60 bits either way)
    const int64_t good_enough = 4 * k <= kSmallSort ? kSmallSort : std::max<int64_t>(std::max<int64_t>(4 * k, 65536), n / 1024);
    int prefix_bits = 0;
    unsigned long long prefix = 0;
    long long below = 0;
    long long host_control[3] = {0, 0, 0};
    for (int level = 0; level < 5 && prefix_bits < width; ++level) {
      const int bits = width - prefix_bits < 12 ? width - prefix_bits : 12;
      QSX_HIP_TRY(hipMemsetAsync(hist, 0, kTopBins * 8 + 64, s));
      if (narrow) {
        hipLaunchKernelGGL(topk_hist_kernel<uint32_t>, dim3(grid), dim3(kSBlock), 0, s, static_cast<const uint32_t *>(key_cols[0]), n,
                           type, desc, prefix_bits, prefix, bits, hist);
      } else {
        hipLaunchKernelGGL(topk_hist_kernel<unsigned long long>, dim3(grid), dim3(kSBlock), 0, s,
                           static_cast<const unsigned long long *>(key_cols[0]), n, type, desc, prefix_bits, prefix, bits, hist);
      }
      hipLaunchKernelGGL(topk_threshold_kernel, dim3(1), dim3(64), 0, s, hist, static_cast<long long>(k), below, control);
      QSX_CHECK_LAUNCH();
      QSX_HIP_TRY(hipMemcpyAsync(host_control, control, sizeof(host_control), hipMemcpyDeviceToHost, s));
      QSX_HIP_TRY(hipStreamSynchronize(s));
      prefix = (prefix << bits) | static_cast<unsigned long long>(host_control[0]);
      prefix_bits += bits;
      below = host_control[2];
      if (host_control[1] <= good_enough) break;
    }
    if (host_control[1] >= k && host_control[1] <= n / 2) {
      if (narrow) {
        hipLaunchKernelGGL(topk_mark_kernel<uint32_t>, dim3(grid), dim3(kSBlock), 0, s, static_cast<const uint32_t *>(key_cols[0]), n,
                           type, desc, prefix_bits, prefix, bitmap);
      } else {
        hipLaunchKernelGGL(topk_mark_kernel<unsigned long long>, dim3(grid), dim3(kSBlock), 0, s,
                           static_cast<const unsigned long long *>(key_cols[0]), n, type, desc, prefix_bits, prefix, bitmap);
      }
      QSX_CHECK_LAUNCH();
    }
    if (host_control[1] >= k && host_control[1] <= n / 2) {
      // candidates in input order (ties of the final sort keep the input order, like a stable sort of everything)
      int64_t *count_dev = reinterpret_cast<int64_t *>(control + 4);
      rc = qsx_bitmap_to_tids(bitmap, n, 0, tids_a, count_dev, compact_ws, ws.compact, stream);
      if (rc != QSX_OK) return rc;
      m = host_control[1];
      selected = true;
    }
  }
  if (!selected) {
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n, kSBlock * 4)), dim3(kSBlock), 0, s, tids_a, n);
    QSX_CHECK_LAUNCH();
  }
  int32_t *result = nullptr;
  rc = sort_tids(nkeys, key_cols, key_types, descending, m, tids_a, tids_b, w, SortWorkspace(n), &result, s);
  if (rc != QSX_OK) return rc;
  QSX_HIP_TRY(hipMemcpyAsync(out_tids_dev, result, static_cast<size_t>(k) * 4, hipMemcpyDeviceToDevice, s));
  return QSX_OK;
}

int qsx_distinct_rows(int ncols, const void *const *cols, const int32_t *types, int64_t n, const uint64_t *filter_dev,
                      int32_t *out_tids_dev, int64_t *out_count_dev, void *workspace_dev, size_t workspace_bytes,
                      qsx_stream_t stream) {
  QSX_REQUIRE_DEVICE();
  int rc = validate_sort_args(ncols, cols, types, n, out_tids_dev);
  if (rc != QSX_OK) return rc;
  if (out_count_dev == nullptr) return QSX_ERR_INVALID_ARGUMENT;
  hipStream_t s = as_stream(stream);
  QSX_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), s));
  if (n == 0) return QSX_OK;
  const SortWorkspace ws(n);
  if (workspace_dev == nullptr || workspace_bytes < ws.total) return QSX_ERR_CAPACITY;
  char *w = static_cast<char *>(workspace_dev);
  int32_t *tids_a = reinterpret_cast<int32_t *>(w + 2 * ws.keys + ws.tids);
  int32_t *tids_b = reinterpret_cast<int32_t *>(w + 2 * ws.keys);
  char *extra = w + 2 * ws.keys + 2 * ws.tids + ws.offsets + ws.part;
  int64_t *count_dev = reinterpret_cast<int64_t *>(extra);                  // inside the hist piece
  uint64_t *bitmap = reinterpret_cast<uint64_t *>(extra + ws.hist);
  void *compact_ws = extra + ws.hist + ws.bitmap;
  int64_t m = n;
  if (filter_dev != nullptr) {   // the selected rows, in input order
    rc = qsx_bitmap_to_tids(filter_dev, n, 0, tids_a, count_dev, compact_ws, ws.compact, stream);
    if (rc != QSX_OK) return rc;
    QSX_HIP_TRY(hipMemcpyAsync(&m, count_dev, sizeof(m), hipMemcpyDeviceToHost, s));
    QSX_HIP_TRY(hipStreamSynchronize(s));
    if (m == 0) return QSX_OK;
  } else {
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n, kSBlock * 4)), dim3(kSBlock), 0, s, tids_a, n);
    QSX_CHECK_LAUNCH();
  }
  int32_t *sorted = nullptr;
  rc = sort_tids(ncols, cols, types, nullptr, m, tids_a, tids_b, w, ws, &sorted, s);
  if (rc != QSX_OK) return rc;
  TupleColumns t{};
  t.ncols = ncols;
  for (int c = 0; c < ncols; ++c) { t.col[c] = cols[c]; t.type[c] = types[c]; }
  hipLaunchKernelGGL(run_heads_kernel, dim3(grid_for(m, kSBlock * 4)), dim3(kSBlock), 0, s, t, sorted, m, bitmap);
  QSX_CHECK_LAUNCH();
  // positions of the run heads -> their row numbers
  int32_t *positions = sorted == tids_a ? tids_b : tids_a;
  rc = qsx_bitmap_to_tids(bitmap, m, 0, positions, out_count_dev, compact_ws, ws.compact, stream);
  if (rc != QSX_OK) return rc;
  int64_t distinct = 0;
  QSX_HIP_TRY(hipMemcpyAsync(&distinct, out_count_dev, sizeof(distinct), hipMemcpyDeviceToHost, s));
  QSX_HIP_TRY(hipStreamSynchronize(s));
  return qsx_gather(4, sorted, positions, distinct, out_tids_dev, stream);
}

}  // extern "C"
