// scan.hpp — single-workgroup exclusive scan over per-tile counts.  Tile counts
// are tiny next to the data they summarise (one int32 per 4096 rows), so one
// 1024-thread workgroup walking them in chunks is far below the cost of the
// passes around it.
#ifndef QSX_CSRC_SCAN_HPP_
#define QSX_CSRC_SCAN_HPP_

#include "common.hpp"

namespace qsx {

// offsets has num_tiles + 1 entries; the total also goes to *total_out when non-null.
// Each of the 1024 threads owns 16 consecutive counts per round (16 Ki counts per round).
constexpr int kScanItems = 16;
template <typename CountT>
static __device__ __forceinline__ void tile_scan_body(const CountT *__restrict__ counts, int64_t num_tiles,
                                                      int64_t *__restrict__ offsets, int64_t *__restrict__ total_out) {
  __shared__ int64_t wave_totals[16];
  __shared__ int64_t carry;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < num_tiles; base += 1024 * kScanItems) {
    const int64_t first = base + static_cast<int64_t>(threadIdx.x) * kScanItems;
    CountT c[kScanItems];
    int64_t local = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      c[k] = first + k < num_tiles ? counts[first + k] : 0;
      local += c[k];
    }
    int64_t incl = local;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int64_t up = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += up;
    }
    if (lane == kWave - 1) wave_totals[wave] = incl;
    __syncthreads();
    int64_t wave_base = 0;
    for (int w = 0; w < wave; ++w) wave_base += wave_totals[w];
    const int64_t block_carry = carry;
    int64_t run = block_carry + wave_base + incl - local;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      if (first + k < num_tiles) offsets[first + k] = run;
      run += c[k];
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = run;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    offsets[num_tiles] = carry;
    if (total_out != nullptr) *total_out = carry;
  }
}

// ---- multi-workgroup scan for large count arrays (radix-partition histograms) --------------
// Phase 1: every workgroup sums its chunk of 4096 counts; phase 2: one workgroup scans the
// chunk sums; phase 3: every workgroup scans its chunk again on top of its base.
constexpr int kScanChunk = 4096;  // 256 threads x 16 counts

static __global__ __launch_bounds__(256) void scan_chunk_sums_kernel(const int32_t *__restrict__ counts, int64_t n,
                                                                     int64_t *__restrict__ chunk_sums) {
  const int64_t first = static_cast<int64_t>(blockIdx.x) * kScanChunk + static_cast<int64_t>(threadIdx.x) * 16;
  int64_t local = 0;
  if (first + 16 <= n) {
    const int4 *v = reinterpret_cast<const int4 *>(counts + first);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int4 q = v[k];
      local += static_cast<int64_t>(q.x) + q.y + q.z + q.w;
    }
  } else {
    for (int k = 0; k < 16; ++k) {
      if (first + k < n) local += counts[first + k];
    }
  }
  local = wave_reduce_add(local);
  __shared__ int64_t s_sum[4];
  if (lane_id() == 0) s_sum[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) chunk_sums[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

static __global__ __launch_bounds__(256) void scan_chunks_kernel(const int32_t *__restrict__ counts, int64_t n,
                                                                 const int64_t *__restrict__ chunk_bases,
                                                                 int64_t num_chunks, int64_t *__restrict__ offsets,
                                                                 int64_t *__restrict__ total_out) {
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t first = static_cast<int64_t>(blockIdx.x) * kScanChunk + static_cast<int64_t>(threadIdx.x) * 16;
  int32_t c[16];
  int64_t local = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    c[k] = first + k < n ? counts[first + k] : 0;
    local += c[k];
  }
  int64_t incl = local;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int64_t up = __shfl_up(incl, off, kWave);
    if (lane >= off) incl += up;
  }
  __shared__ int64_t s_wave[4];
  if (lane == kWave - 1) s_wave[wave] = incl;
  __syncthreads();
  int64_t run = chunk_bases[blockIdx.x] + incl - local;
  for (int w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (first + k < n) offsets[first + k] = run;
    run += c[k];
  }
  if (blockIdx.x == num_chunks - 1 && threadIdx.x == 255) {
    offsets[n] = chunk_bases[num_chunks];  // grand total
    if (total_out != nullptr) *total_out = chunk_bases[num_chunks];
  }
}

static __global__ __launch_bounds__(1024) void tile_scan_kernel(const int32_t *__restrict__ counts, int64_t num_tiles,
                                                                int64_t *__restrict__ offsets,
                                                                int64_t *__restrict__ total_out) {
  tile_scan_body<int32_t>(counts, num_tiles, offsets, total_out);
}
static __global__ __launch_bounds__(1024) void tile_scan_kernel_i64(const int64_t *__restrict__ counts, int64_t num_tiles,
                                                                    int64_t *__restrict__ offsets,
                                                                    int64_t *__restrict__ total_out) {
  tile_scan_body<int64_t>(counts, num_tiles, offsets, total_out);
}

// Scratch (in int64 words) launch_scan needs for n counts.
inline size_t scan_workspace_words(int64_t n) { return 2 * static_cast<size_t>((n + kScanChunk - 1) / kScanChunk + 2); }

// Exclusive scan of n int32 counts into n + 1 int64 offsets, stream-ordered.
inline hipError_t launch_scan(const int32_t *counts, int64_t n, int64_t *offsets, int64_t *total_out, int64_t *workspace,
                              hipStream_t stream) {
  if (n <= 4 * kScanChunk || workspace == nullptr) {
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, n, offsets, total_out);
    return hipGetLastError();
  }
  const int64_t chunks = (n + kScanChunk - 1) / kScanChunk;
  int64_t *sums = workspace, *bases = workspace + chunks + 1;
  hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3(static_cast<unsigned>(chunks)), dim3(256), 0, stream, counts, n, sums);
  hipLaunchKernelGGL(tile_scan_kernel_i64, dim3(1), dim3(1024), 0, stream, sums, chunks, bases, static_cast<int64_t *>(nullptr));
  hipLaunchKernelGGL(scan_chunks_kernel, dim3(static_cast<unsigned>(chunks)), dim3(256), 0, stream, counts, n, bases, chunks,
                     offsets, total_out);
  return hipGetLastError();
}

}  // namespace qsx

#endif  // QSX_CSRC_SCAN_HPP_
